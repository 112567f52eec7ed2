#!/usr/bin/env python3
"""The exact-fallback cliff: a shard whose rows are ALL the same vector (every score ties, the candidate list overflows
at any threshold, the crowd path cannot separate anything) answered through the device-wide radix select.

    python tools/fallback_probe.py [--rows 10000000] [--dim 768] [--k 10]
    NMN_NO_GRID_SELECT=1 python tools/fallback_probe.py     # round 1's single-workgroup select, for the A/B
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neumann_amd import GpuFlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    v = rng.standard_normal(a.dim).astype(np.float32)
    chunk = 1_000_000
    src = torch.from_numpy(v).cuda().repeat(chunk, 1).contiguous()
    idx = GpuFlatIndex(a.dim, a.rows)
    for r0 in range(0, a.rows, chunk):
        idx.upload_device(src[: min(chunk, a.rows - r0)], r0)
    torch.cuda.synchronize()
    idx.set_timing(True)
    q = rng.standard_normal(a.dim).astype(np.float32)
    out = {"workload": f"{a.rows} identical rows x {a.dim}, TOP-{a.k}",
           "select": "single workgroup (NMN_NO_GRID_SELECT)" if os.environ.get("NMN_NO_GRID_SELECT") else "device-wide radix select"}
    for metric, name in ((0, "cosine"), (1, "euclidean"), (2, "dot")):
        for _ in range(2):
            idx.search(q, a.k, metric)
        wall, dev = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            rows, scores, counts, st = idx.search(q, a.k, metric, with_stats=True)
            wall.append((time.perf_counter() - t0) * 1e3)
            dev.append(st.total_ms)
        assert counts[0] == a.k and list(rows[0]) == list(range(a.k)), rows[0]   # ties -> ascending row ids
        assert len(set(scores[0].view(np.uint32).tolist())) == 1
        out[name] = {"wall_ms_median": round(float(np.median(wall)), 3), "device_ms_median": round(float(np.median(dev)), 3),
                     "fallback_queries": int(st.fallback_queries)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
