#!/usr/bin/env python
"""bench.py's IVF leg alone (2M x 768 CLUSTERED rows, 256 lists, nprobe 8, k = 100): 40 single-query probes.  For rocprofv3 traces."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd.ivf import GpuIvfFlat  # noqa: E402

d, k = 768, 100
rn, C_, nprobe, tn = 2_000_000, 256, 8, 200_000
rng = np.random.default_rng(5)
centers = rng.standard_normal((C_, d)).astype(np.float32) * np.float32(2.0)
pool = rng.standard_normal((8192, d)).astype(np.float32)


def rows_of(a, b):
    i = np.arange(a, b, dtype=np.int64)
    r = centers[(i * 2654435761 >> 9) % C_] + pool[(i * 40503 + 17) % 8192]
    r[:, 0] += ((i % 100003) * np.float32(1e-5)).astype(np.float32)
    return r


ivf = GpuIvfFlat.build(rows_of(0, tn), C_, nprobe=nprobe, max_iterations=3, seed=42, init_method="kmeans++", capacity_rows=rn)
with ivf:
    for a in range(tn, rn, 300_000):
        ivf.add(rows_of(a, min(a + 300_000, rn)))
    Q = rows_of(12345, 12345 + 32) + np.float32(0.05)
    for i in range(8):
        ivf.search(Q[i], k)
    t0 = time.perf_counter()
    for i in range(40):
        ivf.search(Q[i % 32], k)
    print(f"{(time.perf_counter() - t0) / 40 * 1e3:.4f} ms per single-query probe, list-major rows {ivf.list_major_rows}")
