#!/usr/bin/env python
"""Measure the WHERE-predicate kernel (neumann_amd/csrc/nmn_columns.hip) at BASELINE config-5 scale.

  python tools/filter_bench.py [--rows 10000000] [--host-rows 1000000]

Part 1: 10M rows x 3 metadata columns resident in HBM; predicates of 1..5 leaves; wall time per
evaluation (program upload + kernel + count read-back, what the engine pays per filtered query) and
the algorithmic bytes the leaves read (1 B kind + 8 B payload per row and leaf, 1 B for Exists).
Part 2: the host evaluator the device path replaces — the C++ port of evaluate_filter
(nmn_engine.cpp, following vector_engine/src/lib.rs:3592-3670) walking every entry, timed through
`count_matching` — against the engine's device pre-filter on the same store.
Prints one JSON object."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--host-rows", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from neumann_amd import columns as g
    from neumann_amd import engine as E

    n = args.rows
    rng = np.random.default_rng(1)
    out = {"rows": n, "predicates": []}
    with g.GpuColumns(n) as gc:
        c_price, c_score, c_cat = gc.add_column(), gc.add_column(), gc.add_column()
        price = rng.integers(0, 1000, n)
        gc.write(c_price, 0, np.full(n, g.CELL_INT, np.uint8), price.astype(np.uint64))
        score = rng.random(n)
        gc.write(c_score, 0, np.full(n, g.CELL_FLOAT, np.uint8), score.view(np.uint64))
        cat = rng.integers(0, 1000, n)
        gc.write(c_cat, 0, np.full(n, g.CELL_STRING, np.uint8), cat.astype(np.uint64))
        gc.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
        lt = (g.PRED_CMP, g.CMP_LT, g.CELL_INT, c_price, 100, 0)
        gt = (g.PRED_CMP, g.CMP_GT, g.CELL_FLOAT, c_score, g.f64_bits(0.5), 0)
        ex = (g.PRED_EXISTS, 0, 0, c_cat, 0, 0)
        bitset = [int(x) for x in np.packbits(np.arange(1024) % 10 == 0, bitorder="little").view(np.uint64)]
        ss = (g.PRED_STRSET, 0, 0, c_cat, 0, 1000)
        AND, OR = (g.PRED_AND, 0, 0, 0, 0, 0), (g.PRED_OR, 0, 0, 0, 0, 0)
        cases = [
            ("price < 100", [lt], [], 9, int((price < 100).sum())),
            ("exists(category)", [ex], [], 1, n),
            ("category in <100 of 1000 strings>", [ss], bitset, 9, int((cat % 10 == 0).sum())),
            ("price < 100 and score > 0.5", [lt, gt, AND], [], 18, int(((price < 100) & (score > 0.5)).sum())),
            ("(price < 100 and score > 0.5) or category in set", [lt, gt, AND, ss, OR], bitset, 27,
             int((((price < 100) & (score > 0.5)) | (cat % 10 == 0)).sum())),
        ]
        for name, prog, consts, bytes_per_row, expect in cases:
            cnt = gc.eval(prog, consts, n)
            assert cnt == expect, (name, cnt, expect)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                gc.eval(prog, consts, n)
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            out["predicates"].append({"predicate": name, "selected": cnt, "ms_per_eval_wall": round(ms, 4),
                                      "rows_per_s": round(n / ms * 1e3), "algorithmic_bytes": bytes_per_row * n + n // 8,
                                      "GBps_wall": round((bytes_per_row * n + n // 8) / ms / 1e6, 1)})

    # ---- host evaluator vs device pre-filter on the same engine store ----
    hn, d = args.host_rows, 8
    engine = E.VectorEngine()
    vecs = rng.standard_normal((hn, d)).astype(np.float32)
    t0 = time.perf_counter()
    for i in range(hn):
        engine.store_embedding_with_metadata(f"k{i}", vecs[i], {"price": int(price[i]), "score": float(score[i]),
                                                                 "category": f"c{int(cat[i])}"})
    t_store = time.perf_counter() - t0
    FC = E.FilterCondition
    cond = FC.Lt("price", 100).and_(FC.Gt("score", 0.5))
    t0 = time.perf_counter()
    host_cnt = engine.count_matching(cond)                    # host evaluator over every entry
    t_host = time.perf_counter() - t0
    q = rng.standard_normal(d).astype(np.float32)
    cfg = E.FilteredSearchConfig.pre_filter()
    engine.search_similar_filtered(q, 10, cond, cfg)          # builds mirror + columns
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        res = engine.search_similar_filtered(q, 10, cond, cfg)
    t_dev = (time.perf_counter() - t0) / reps
    assert engine.device_filter_evals() == reps + 1 and len(res) == 10
    out["engine"] = {"rows": hn, "store_seconds": round(t_store, 2), "matching": host_cnt,
                     "host_evaluator_ms": round(t_host * 1e3, 2), "host_rows_per_s": round(hn / t_host),
                     "device_prefilter_search_ms": round(t_dev * 1e3, 3),
                     "note": "device figure is the WHOLE filtered query (predicate kernel + masked scan + top-k); "
                             "host figure is the predicate alone (1 thread, C++ port of evaluate_filter)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
