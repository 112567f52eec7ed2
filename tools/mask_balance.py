"""How much of a masked f32 sweep's time is imbalance between waves?  The same sweep (10M x 1536 Euclidean TOP-1000 over the f32 rows)
under a RANDOM bitmap of selectivity s and under a REGULAR one (every (1/s)-th row: every wave holds the same number of kept rows).
   python tools/mask_balance.py [--rows N] [--dim D]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, d = a.rows, a.dim
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(0)
        idx.fill_synthetic(7, n)
        q = torch.randn(8, d, device=dev)
        words = (n + 63) // 64
        rng = np.random.default_rng(3)
        for period in (2, 10, 50, 250):
            s = 1.0 / period
            keep_r = rng.random(n) < s
            keep_g = (np.arange(n) % period) == 0
            keep_g2 = ((np.arange(n) * 2654435761) % (2 ** 32) % period) == 0  # (pseudo-random but a different draw)
            nb = n // period
            keep_b = np.zeros(n, dtype=bool)  # one kept row per block of `period` rows, at a random place in it: level counts, random addresses
            keep_b[np.arange(nb) * period + rng.integers(0, period, nb)] = True
            for name, keep in (("random", keep_r), ("hashed", keep_g2), ("regular", keep_g), ("blocked", keep_b)):
                bits = np.zeros(words * 64, dtype=bool)
                bits[:n] = keep
                m = np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
                mt = torch.from_numpy(m.view(np.int64)).to(dev)
                st = torch.cuda.current_stream()
                idx.set_timing(2)
                for i in range(3):
                    idx.search_device(q[i % 8:i % 8 + 1], a.k, a.metric, mask_t=mt)
                torch.cuda.synchronize()
                idx.scan_history(st)
                for i in range(a.steps):
                    idx.search_device(q[i % 8:i % 8 + 1], a.k, a.metric, mask_t=mt)
                torch.cuda.synchronize()
                ms = [x for x in idx.scan_history(st) if x > 0]
                idx.set_timing(False)
                kept = int(keep.sum())
                avg = float(np.median(ms))
                gbs = kept * d * 4 / avg / 1e6
                print(f"selectivity 1/{period:<4} {name:<8} kept {kept:>8}  sweep {avg:.4f} ms  {gbs:7.0f} GB/s  frac {gbs / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()
