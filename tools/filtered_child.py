#!/usr/bin/env python
"""bench.py's filtered-SIMILAR leg alone (nmn_index_search_pred, WHERE bucket = 3 of 10 over 10M x 768, selectivity 0.1): N calls, wall
time per call — the child for rocprofv3 traces (tools/trace_chain.py DB --first pred_batch prints the launch chain of the last call).

    python tools/filtered_child.py [--rows 10000000] [--dim 768] [--k 100] [--reps 40] [--mirror 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402
from neumann_amd import columns as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--mirror", type=int, default=1)
a = ap.parse_args()
n, d = a.rows, a.dim
with GpuFlatIndex(d, n, device=0) as idx, g.GpuColumns(n) as cols:
    idx.set_mirror(a.mirror)
    idx.fill_synthetic(0x5EED0003, n)
    col = cols.add_column()
    bucket = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(10)
    cols.write(col, 0, np.full(n, g.CELL_INT, np.uint8), bucket)
    cols.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    prog = [(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, col, 3, 0)]
    Q = synth_rows(0x5EED0002 + 7, 0, 8, d)
    for i in range(4):
        idx.search_pred(cols, prog, [], Q[i], a.k, 0)
    t = []
    for i in range(a.reps):
        t0 = time.perf_counter()
        rows, scores, counts, selected = idx.search_pred(cols, prog, [], Q[i % 8], a.k, 0)
        t.append(time.perf_counter() - t0)
    print(json.dumps({"rows": n, "dim": d, "k": a.k, "selected": int(selected), "wall_ms_median": float(np.median(t)) * 1e3,
                      "wall_ms_p10": float(np.percentile(t, 10)) * 1e3}))
