#!/usr/bin/env python3
"""Host upload rate of nmn_index_upload (pageable numpy rows -> shard: H2D copy + ingest kernel), by upload size.

    python tools/host_upload_bench.py [--dim 768] [--rows 2000000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from neumann_amd import GpuFlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--rows", type=int, default=2_000_000)
    a = ap.parse_args()
    A = np.random.default_rng(1).standard_normal((a.rows, a.dim), dtype=np.float32)
    with GpuFlatIndex(a.dim, a.rows) as idx:
        idx.upload(A[:4096])  # warm-up (allocations, first-touch)
        for n in (10_000, 100_000, 500_000, a.rows):
            n = min(n, a.rows)
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                idx.upload(A[:n], row0=0)
                best = min(best, time.perf_counter() - t)
            print(f"upload {n:>8} rows x {a.dim}: {best * 1e3:8.2f} ms  {n * a.dim * 4 / best / 1e9:6.1f} GB/s of host rows")


if __name__ == "__main__":
    main()
