#!/usr/bin/env python
"""IVF-Flat probe throughput on one MI355X (neumann_amd/csrc/nmn_ivf.hip).

  python tools/ivf_bench.py [--rows 2000000] [--dim 768] [--clusters 256] [--nprobe 16] [--k 100]

Centroids are a random sample of the rows (the k-means is the host's job and not what is measured);
reports add() throughput (exact nearest-centroid assignment), queries/s of the probe and, for scale,
queries/s of the exhaustive Euclidean scan over the same rows.  One JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--clusters", type=int, default=256)
    ap.add_argument("--nprobe", type=int, default=16)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=50)
    args = ap.parse_args()
    from neumann_amd import _capi
    from neumann_amd.flat_index import synth_rows
    from neumann_amd.ivf import GpuIvfFlat
    import ctypes as C

    n, d = args.rows, args.dim
    chunk = 250_000
    cents = synth_rows(0x1F5, 0, args.clusters, d)  # same generator as the rows: plausible centroids
    out = {"rows": n, "dim": d, "clusters": args.clusters, "nprobe": args.nprobe, "k": args.k}
    with GpuIvfFlat(cents, capacity_rows=n, nprobe=args.nprobe) as ivf:
        t_add = 0.0
        for r0 in range(0, n, chunk):
            rows = synth_rows(0x1F6, r0, min(chunk, n - r0), d)
            t0 = time.perf_counter()
            ivf.add(rows)
            t_add += time.perf_counter() - t0
        sizes = ivf.cluster_sizes()
        out["add_rows_per_s"] = round(n / t_add)
        out["list_size_min_mean_max"] = [int(sizes.min()), float(sizes.mean()), int(sizes.max())]
        Q = synth_rows(0x1F7, 0, args.queries, d)
        ivf.search(Q[0], args.k)
        t0 = time.perf_counter()
        scanned = 0
        for q in Q:
            ids, dist, counts = ivf.search(q, args.k)
        dt = (time.perf_counter() - t0) / len(Q)
        out["probe_ms_per_query"] = round(dt * 1e3, 3)
        out["probe_queries_per_s"] = round(1 / dt, 1)
        out["rows_in_probed_lists_mean"] = float(sizes.mean()) * args.nprobe
        out["probe_GBps_algorithmic"] = round(out["rows_in_probed_lists_mean"] * d * 4 / dt / 1e9, 1)
        # exhaustive Euclidean scan over the same rows through the same C ABI
        lib = _capi.load()
        vec = lib.nmn_ivf_vectors(ivf._h)
        rows_o = np.empty(args.k, np.uint64); sc_o = np.empty(args.k, np.float32); cnt_o = np.empty(1, np.uint32)
        def flat(q):
            _capi.check(lib.nmn_index_search(vec, C.c_void_p(q.ctypes.data), 1, args.k, 1, None, C.c_void_p(rows_o.ctypes.data),
                                             C.c_void_p(sc_o.ctypes.data), C.c_void_p(cnt_o.ctypes.data), None))
        flat(Q[0])
        t0 = time.perf_counter()
        for q in Q:
            flat(q)
        dt2 = (time.perf_counter() - t0) / len(Q)
        out["flat_ms_per_query"] = round(dt2 * 1e3, 3)
        out["speedup_vs_flat"] = round(dt2 / dt, 2)
        # recall of the probe against the exhaustive ranking for the last query
        out["recall_at_k_last_query"] = float(np.intersect1d(ids[0], rows_o).size) / args.k
    print(json.dumps(out))


if __name__ == "__main__":
    main()
