#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/ (run from the repo root).

The reference is Rust and cannot be built or imported in this container (no cargo/rustc), and its
own tests hold no bit-level vectors for this path, so the fixtures are produced by
the numpy twin of the oracle (oracle/oracle_np.py) and cross-checked here against the C oracle.
Every fixture is data only: inputs (explicit arrays, or a seed of the integer-hash generator whose
host implementation is oracle/nmn_oracle.c:orc_synth_value) and expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_np as on  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def expect(A, q, k, metric, keep=None):
    r, s = on.search(A, q, k, metric, keep=keep)
    mask = None if keep is None else oc.mask_from_bool(keep)
    r2, s2 = oc.search(A, q, k, metric, mask=mask)
    assert np.array_equal(r, r2) and np.array_equal(s.view(np.uint32), s2.view(np.uint32)), "C vs numpy oracle"
    return r, s


def small_explicit():
    rng = np.random.default_rng(20260928)
    A = rng.standard_normal((300, 40)).astype(F)
    A[17] = 0.0                      # a zero row: cosine 0.0, still returned
    A[101] = A[100]                  # exact duplicates: tie broken by row id
    A[102] = A[100]
    Q = rng.standard_normal((4, 40)).astype(F)
    Q[3] = A[100]                    # query equal to a stored (triplicated) row
    keep50 = rng.random(300) < 0.5
    keep10 = rng.random(300) < 0.1
    d = {"A": A, "Q": Q, "keep50": keep50, "keep10": keep10}
    for m in (0, 1, 2):
        for qi in range(4):
            for tag, keep in (("all", None), ("k50", keep50), ("k10", keep10)):
                r, s = expect(A, Q[qi], 10, m, keep)
                d[f"rows_m{m}_q{qi}_{tag}"] = r
                d[f"scores_m{m}_q{qi}_{tag}"] = s
    np.savez_compressed(os.path.join(OUT, "small_explicit.npz"), **d)


def example_vector_search():
    """examples/vector_search.rs:26-67 data, queries :80,:96,:112, TOP 3 (values are test DATA)."""
    docs = np.array([
        [0.8, 0.7, 0.1, 0.2, 0.1, 0.1, 0.1, 0.1], [0.9, 0.8, 0.2, 0.1, 0.1, 0.1, 0.1, 0.1],
        [0.85, 0.75, 0.15, 0.15, 0.1, 0.1, 0.1, 0.1], [0.1, 0.1, 0.8, 0.7, 0.2, 0.1, 0.1, 0.1],
        [0.1, 0.1, 0.75, 0.8, 0.25, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.2, 0.8, 0.7, 0.1, 0.1],
        [0.2, 0.1, 0.3, 0.3, 0.3, 0.3, 0.8, 0.7], [0.15, 0.1, 0.25, 0.25, 0.25, 0.25, 0.75, 0.8]], dtype=F)
    Q = np.array([[0.85, 0.75, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1], [0.1, 0.1, 0.8, 0.75, 0.2, 0.1, 0.1, 0.1],
                  [0.1, 0.1, 0.2, 0.2, 0.2, 0.2, 0.8, 0.8]], dtype=F)
    d = {"A": docs, "Q": Q}
    for m in (0, 1, 2):
        for qi in range(3):
            r, s = expect(docs, Q[qi], 3, m)
            d[f"rows_m{m}_q{qi}"] = r
            d[f"scores_m{m}_q{qi}"] = s
    np.savez_compressed(os.path.join(OUT, "example_vector_search.npz"), **d)


def synth_case(name, seed, n, dim, k, metrics, sels, n_queries=3, plant=True):
    """Corpus = generator(seed) with a few planted rows; stores only seed + planted rows + expectations."""
    A = oc.synth(seed, 0, n, dim)
    Q = oc.synth(seed ^ 0xABCDEF, 0, n_queries, dim)
    planted_idx = np.zeros(0, dtype=np.int64)
    planted = np.zeros((0, dim), dtype=F)
    if plant:
        # near-duplicates of query 0 (scores within a few ulps of each other) + exact duplicates
        rng = np.random.default_rng(seed)
        planted_idx = rng.choice(n, size=16, replace=False).astype(np.int64)
        planted = np.repeat(Q[0][None, :], 16, axis=0).copy()
        for i in range(16):
            if i >= 4:  # rows 0..3 stay exact copies of the query (and of each other)
                j = rng.integers(0, dim, size=3)
                planted[i, j] = np.nextafter(planted[i, j], F(np.inf) if i % 2 else F(-np.inf))
        A[planted_idx] = planted
    d = {"seed": np.uint64(seed), "n": np.int64(n), "dim": np.int64(dim), "k": np.int64(k), "Q": Q,
         "planted_idx": planted_idx, "planted": planted}
    rng = np.random.default_rng(seed + 1)
    for sel in sels:
        keep = None
        tag = "all"
        if sel < 1.0:
            keep = rng.random(n) < sel
            tag = f"sel{int(sel * 100)}"
            d[f"mask_{tag}"] = oc.mask_from_bool(keep)
        for m in metrics:
            for qi in range(n_queries):
                r, s = expect(A, Q[qi], k, m, keep)
                d[f"rows_m{m}_q{qi}_{tag}"] = r
                d[f"scores_m{m}_q{qi}_{tag}"] = s
    np.savez_compressed(os.path.join(OUT, name), **d)


if __name__ == "__main__":
    small_explicit()
    example_vector_search()
    synth_case("synth_10000x128_top5.npz", 0x5EED0010, 10000, 128, 5, (0, 1, 2), (1.0,))
    synth_case("synth_4096x768_top100.npz", 0x5EED0001, 4096, 768, 100, (0, 1, 2), (1.0, 0.5, 0.1))
    synth_case("synth_4096x1536_top1000.npz", 0x5EED0005, 4096, 1536, 1000, (1,), (1.0, 0.5, 0.1), n_queries=2)
    synth_case("synth_5000x100_top64.npz", 0x5EED0077, 5000, 100, 64, (0, 1, 2), (1.0, 0.5))  # dim % 8 != 0
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
