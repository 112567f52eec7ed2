import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from neumann_amd import engine as E
rng = np.random.default_rng(0)
d = 768
e = E.VectorEngine()
A = rng.standard_normal((50000, d)).astype(np.float32)
t0 = time.perf_counter()
for i in range(20000):
    e.store_embedding(f"k{i}", A[i])
t1 = time.perf_counter()
print("stores without a mirror: %.0f /s" % (20000 / (t1 - t0)))
e.search_similar(A[0], 10)          # builds the mirror
t0 = time.perf_counter()
for i in range(20000, 30000):
    e.store_embedding(f"k{i}", A[i])
t1 = time.perf_counter()
print("appends with a mirror:   %.0f /s" % (10000 / (t1 - t0)))
t0 = time.perf_counter()
for i in range(0, 10000):
    e.store_embedding(f"k{i}", A[i + 30000])
t1 = time.perf_counter()
print("overwrites with a mirror: %.0f /s" % (10000 / (t1 - t0)))
t0 = time.perf_counter()
r = e.search_similar(A[31000], 5)
print("search after: %.3f ms" % ((time.perf_counter() - t0) * 1e3), r[0].key)
