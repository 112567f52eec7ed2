#!/usr/bin/env python
"""One shard, N searches — the generic child for rocprofv3 / PMC / timing passes (replaces the one-off *_child.py scripts).

    python tools/search_child.py --rows 10000000 --dim 1536 --metric 1 --k 1000 --mask 0.1 --mirror 1 --api device --nq 1 --reps 8

--api device: nmn_index_search_device on one stream, one search after the other, one synchronize at the end (the asynchronous
chain, all eleven launches); --api host: nmn_index_search (host buffers, the short chain).  --sync: synchronize after every
device search (a lone caller).  Prints one JSON line: wall ms per search (median), bytes per corpus element of the sweep."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--nq", type=int, default=1)
    ap.add_argument("--mask", type=float, default=1.0)
    ap.add_argument("--runs", type=int, default=0, help="bitmap = this many runs of consecutive rows (an IVF probe's shape) instead of random bits")
    ap.add_argument("--mirror", type=int, default=1)
    ap.add_argument("--api", default="device", choices=["device", "host"])
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0x5EED0003)
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    with GpuFlatIndex(a.dim, a.rows, device=0) as idx:
        idx.set_mirror(a.mirror)
        idx.fill_synthetic(a.seed, a.rows)
        mask = mask_t = None
        kept = a.rows
        if a.mask < 1.0:
            if a.runs:
                keep = np.zeros(a.rows, dtype=bool)
                per = int(a.rows * a.mask / a.runs)
                for r in range(a.runs):
                    s0 = (2 * r + 1) * a.rows // (2 * a.runs)
                    keep[s0:s0 + per] = True
            else:
                keep = np.random.default_rng(5).random(a.rows) < a.mask
            kept = int(keep.sum())
            words = np.packbits(keep, bitorder="little")
            mask = np.pad(words, (0, (-len(words)) % 8)).view(np.uint64)
            mask_t = torch.from_numpy(mask.view(np.int64)).to(dev)
        Q = synth_rows(0x5EED0002, 0, 4 * a.nq, a.dim).reshape(4, a.nq, a.dim)
        Qd = torch.from_numpy(Q).to(dev)
        t = []
        st = None
        for i in range(a.reps + 2):
            t0 = time.perf_counter()
            if a.api == "host":
                _, _, _, st = idx.search(Q[i % 4], a.k, a.metric, mask=mask, with_stats=True)
            else:
                idx.search_device(Qd[i % 4], a.k, a.metric, mask_t=mask_t)
                if a.sync:
                    torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        if st is None:
            st = idx.last_stats(None)
        eb = int(st.bytes_scanned // max(1, st.rows_scanned * a.dim))
        print(json.dumps({"rows": a.rows, "dim": a.dim, "k": a.k, "nq": a.nq, "metric": a.metric, "mask": a.mask, "kept_rows": kept,
                          "api": a.api, "searches": a.reps + 2, "bytes_per_corpus_element": eb,
                          "wall_ms_median": float(np.median(t[2:]) * 1e3) if (a.api == "host" or a.sync) else None,
                          "candidates_rescored": int(st.candidates_rescored)}))


if __name__ == "__main__":
    main()
