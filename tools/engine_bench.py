#!/usr/bin/env python
"""End-to-end throughput of the VectorEngine facade (string keys in, SearchResult lists out) on one GPU.

  python tools/engine_bench.py [--rows 1000000] [--dim 768] [--k 100] [--threads 1,4]

Measures search_similar through neumann_amd.engine (ctypes -> C++ nmn_engine -> libneumann_gpu host-buffer API),
from 1..N Python threads, plus the store rate of the incremental mirror.  NOTE: with more than one Python thread the
numbers are dominated by the GIL hand-over around result marshalling (they drop); the library's own multi-thread scaling
is measured without Python by tools/micro/engine_mt.cpp."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--queries", type=int, default=200)
    args = ap.parse_args()
    from neumann_amd import engine as E
    from neumann_amd.flat_index import synth_rows

    eng = E.VectorEngine()
    chunk = 100_000
    t0 = time.perf_counter()
    for r0 in range(0, args.rows, chunk):
        n = min(chunk, args.rows - r0)
        eng.batch_store_embeddings([f"k{r0 + i}" for i in range(n)], synth_rows(7, r0, n, args.dim))
    t_store = time.perf_counter() - t0
    Q = synth_rows(8, 0, 64, args.dim)
    t0 = time.perf_counter()
    eng.search_similar(Q[0], args.k)          # builds the mirror
    t_first = time.perf_counter() - t0
    out = {"rows": args.rows, "dim": args.dim, "k": args.k, "batch_store_rows_per_s": round(args.rows / t_store),
           "first_search_s_incl_mirror_build": round(t_first, 3), "threads": {}}
    for nt in [int(x) for x in args.threads.split(",")]:
        per = max(1, args.queries // nt)

        def work(tid):
            for i in range(per):
                eng.search_similar(Q[(tid * per + i) % 64], args.k)

        ths = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        out["threads"][str(nt)] = {"queries_per_s": round(per * nt / dt, 1), "ms_per_query": round(dt / (per * nt) * 1e3, 3)}
    # incremental writes while searching: one store + one search per iteration
    t0 = time.perf_counter()
    for i in range(200):
        eng.store_embedding(f"new{i}", Q[i % 64])
        eng.search_similar(Q[(i + 1) % 64], args.k)
    out["store_then_search_ms"] = round((time.perf_counter() - t0) / 200 * 1e3, 3)
    out["mirror_builds"] = eng.mirror_builds()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
