#!/usr/bin/env python3
"""Time of a k > NMN_MAX_TOP_K search (exact scores of every row + ordering of the k best) on a full-size shard.

    python tools/largek_probe.py [--rows 10000000] [--dim 768] [--k 10000 100000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, nargs="*", default=[10_000, 100_000, 3_000_000])
    a = ap.parse_args()
    Q = synth_rows(4, 0, 4, a.dim)
    with GpuFlatIndex(a.dim, a.rows) as idx:
        idx.fill_synthetic(3, a.rows)
        for k in a.k:
            for metric, name in ((0, "cosine"), (1, "euclidean")):
                idx.search(Q[0], k, metric)
                t = []
                for i in range(5):
                    t0 = time.perf_counter()
                    idx.search(Q[i % 4], k, metric)
                    t.append(time.perf_counter() - t0)
                print(f"rows={a.rows} dim={a.dim} k={k} {name}: {np.median(t) * 1e3:.2f} ms per query")


if __name__ == "__main__":
    main()
