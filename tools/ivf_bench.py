#!/usr/bin/env python
"""IVF-Flat probe throughput on one MI355X (neumann_amd/csrc/nmn_ivf.hip).

  python tools/ivf_bench.py [--rows 2000000] [--dim 768] [--clusters 256] [--nprobe 16] [--k 100]

Trains on the first --train-rows rows (k-means++ and Lloyd iterations exactly as the reference runs them, on the GPU),
adds the rest (exact nearest-centroid assignment), then reports queries/s of the probe and, for scale, of the
exhaustive Euclidean scan over the same rows.  One JSON object."""
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
    ap.add_argument("--train-rows", type=int, default=500_000)
    ap.add_argument("--train-iterations", type=int, default=5)
    args = ap.parse_args()
    from neumann_amd import _capi
    from neumann_amd.flat_index import synth_rows
    from neumann_amd.ivf import GpuIvfFlat
    import ctypes as C

    n, d = args.rows, args.dim
    chunk = 250_000
    out = {"rows": n, "dim": d, "clusters": args.clusters, "nprobe": args.nprobe, "k": args.k}
    # ---- training: k-means as the reference runs it, on the GPU (nmn_ivf_build), on the first --train-rows rows ----
    tn = min(n, args.train_rows)
    t0 = time.perf_counter()
    ivf = GpuIvfFlat.build(synth_rows(0x1F6, 0, tn, d), args.clusters, nprobe=args.nprobe, max_iterations=args.train_iterations,
                           seed=42, init_method="kmeans++", capacity_rows=n)
    t_train = time.perf_counter() - t0
    out["train"] = {"rows": tn, "init": "kmeans++", "max_iterations": args.train_iterations, "seconds": round(t_train, 2),
                    "sequential_f32_MACs_restated": float(tn) * args.clusters * d * (args.clusters - 1 + args.train_iterations + 1)}
    with ivf:
        t_add = 0.0
        for r0 in range(tn, n, chunk):
            rows = synth_rows(0x1F6, r0, min(chunk, n - r0), d)
            t0 = time.perf_counter()
            ivf.add(rows)
            t_add += time.perf_counter() - t0
        sizes = ivf.cluster_sizes()
        out["add_rows_per_s"] = round((n - tn) / t_add) if t_add else None
        out["list_size_min_mean_max"] = [int(sizes.min()), float(sizes.mean()), int(sizes.max())]
        Q = synth_rows(0x1F7, 0, args.queries, d)
        ivf.search(Q[0], args.k)
        t0 = time.perf_counter()
        for q in Q:
            ids, dist, counts = ivf.search(q, args.k)
        dt = (time.perf_counter() - t0) / len(Q)
        out["probe_ms_per_query"] = round(dt * 1e3, 3)
        out["probe_queries_per_s"] = round(1 / dt, 1)
        # concurrent callers (probe slots + the flat index's coalescer): T threads, each its own queries
        import threading
        conc = {}
        for T in (4, 16, 64):
            QT = synth_rows(0x1F8, 0, T, d)
            reps = 40
            start = threading.Barrier(T + 1)

            def work(t):
                start.wait()
                for _ in range(reps):
                    ivf.search(QT[t], args.k)

            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            for x in th:
                x.start()
            start.wait()
            t0 = time.perf_counter()
            for x in th:
                x.join()
            conc[str(T)] = round(T * reps / (time.perf_counter() - t0), 1)
        out["probe_queries_per_s_by_threads"] = conc
        # exhaustive Euclidean scan over the same rows through the same C ABI
        lib = _capi.load()
        vec = lib.nmn_ivf_vectors(ivf._h)
        rows_o = np.empty(args.k, np.uint64)
        sc_o = np.empty(args.k, np.float32)
        cnt_o = np.empty(1, np.uint32)

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
        out["recall_at_k_last_query"] = float(np.intersect1d(ids[0], rows_o).size) / args.k
    print(json.dumps(out))


if __name__ == "__main__":
    main()
