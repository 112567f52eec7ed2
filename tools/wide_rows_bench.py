#!/usr/bin/env python
"""NMN_INDEX_WIDE_ROWS A/B: rows of 300 floats stored with stride 304 (default) vs 384 (flag) — single-query and
64-query-batch throughput of the same shard.   python tools/wide_rows_bench.py [rows] [dim]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from neumann_amd import GpuFlatIndex  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
Q = rng.standard_normal((64, dim)).astype(np.float32)
for wide in (False, True):
    with GpuFlatIndex(dim, rows, device=0, wide_rows=wide) as idx:
        idx.fill_synthetic(7, rows)
        out = {}
        for nq in (1, 64):
            q = Q[:nq]
            idx.search(q, 100, 0)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 2.0:
                r = idx.search(q, 100, 0)
                n += 1
            dt = time.perf_counter() - t0
            out[nq] = (nq * n / dt, r)
        print(f"rows={rows} dim={dim} wide_rows={wide} stride={idx.row_stride}: nq=1 {out[1][0]:.0f} q/s, nq=64 {out[64][0]:.0f} q/s")
        if wide:
            same = all(np.array_equal(a, b) for a, b in zip(out[64][1][:2], keep[:2]))
            print("64-query answers identical with and without the flag:", same)
        keep = out[64][1]
