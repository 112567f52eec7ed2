#!/usr/bin/env python3
"""Exactness soak at BASELINE sizes: many queries x metrics x masks x corpus shapes, every answer certified.

The -m gpu tests compare with the oracle at sizes the oracle finishes in seconds.  This tool covers the
sizes the oracle cannot reach with the size-independent certificate bench.py uses (only the product's
exact, reference-order kernels, which the tests pin bit for bit to the oracle):
  1. every returned score == exact score of its row;
  2. the list is ordered (score desc, row asc);
  3. #rows with exact score > s_k == #returned above s_k, and the ties returned at s_k exist.
Corpora: the bench's iid synthetic rows, and a CLUSTERED corpus (tight groups of near-duplicates around
few centres, queries at the centres), which crowds the neighbourhood of the k-th score and drives the
bf16 margins, the f32 retry sweep and the exact fallback.

  python tools/soak.py [--rows 10000000] [--dim 768] [--queries 64] [--out profiles/x.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def certify(idx, q, metric, rows, scores, count, mask):
    r = rows[:count].astype(np.uint64)
    s = scores[:count]
    if count == 0:
        return True
    ex = idx.score_rows(q, r - np.uint64(idx.row_base), metric)[0]
    if not np.array_equal(ex.view(np.uint32), s.view(np.uint32)):
        return False
    if count > 1 and not np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (r[:-1] < r[1:]))):
        return False
    sk = s[-1]
    gt, eq = idx.count_exact(q, float(sk), metric, mask=mask)
    n_gt = int(np.sum(s > sk))
    n_eq = int(np.sum(s == sk))
    return gt == n_gt and 1 <= n_eq <= eq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--clusters", type=int, default=64)
    ap.add_argument("--out", default="")
    ap.add_argument("--corpora", default="iid,clustered,duplicated")
    ap.add_argument("--metrics", default="cosine,euclidean,dot")
    ap.add_argument("--mirror", type=int, default=1, help="nmn_index_set_mirror before the rows arrive: 1 = the library's default, 2 = bf16 only, "
                    "0 = no mirror: single queries on the f32 VALU sweep, batches on the f32-rows matrix-core sweep (round 5)")
    ap.add_argument("--wide-rows", action="store_true", help="NMN_INDEX_WIDE_ROWS: 300 -> 384 etc., batches on the matrix cores")
    args = ap.parse_args()

    import torch
    from neumann_amd import DistanceMetric, GpuFlatIndex
    from neumann_amd.flat_index import synth_rows

    dev = torch.device("cuda:0")
    metrics = [("cosine", DistanceMetric.Cosine), ("euclidean", DistanceMetric.Euclidean),
               ("dot", DistanceMetric.DotProduct)]
    metrics = [m for m in metrics if m[0] in args.metrics.split(",")]
    corpora = args.corpora.split(",")
    report = {"rows": args.rows, "dim": args.dim, "k": args.k, "mirror": args.mirror, "cases": []}
    bad_total = 0
    n, d, nq = args.rows, args.dim, args.queries
    words = (n + 63) // 64
    rng = np.random.default_rng(5)

    def run_cases(idx, tag, queries, masks):
        nonlocal bad_total
        for mname, metric in metrics:
            for mask_name, mask in masks:
                for mode in ("single", "batch"):
                    t0 = time.perf_counter()
                    bad = 0
                    fallback = 0
                    cand = 0
                    if mode == "single":
                        todo = queries[:8]
                        outs = []
                        for q in todo:
                            r, s, c, st = idx.search(q, args.k, metric, mask=mask, with_stats=True)
                            outs.append((r[0], s[0], int(c[0])))
                            fallback += st.fallback_queries
                            cand = max(cand, st.candidates_rescored)
                    else:
                        todo = queries
                        r, s, c, st = idx.search(todo, args.k, metric, mask=mask, with_stats=True)
                        outs = [(r[i], s[i], int(c[i])) for i in range(len(todo))]
                        fallback = st.fallback_queries
                        cand = st.candidates_rescored
                    t_search = time.perf_counter() - t0
                    for q, (r, s, c) in zip(todo, outs):
                        if not certify(idx, q, metric, r, s, c, mask):
                            bad += 1
                    bad_total += bad
                    case = {"corpus": tag, "metric": mname, "mask": mask_name, "mode": mode, "queries": len(todo),
                            "not_certified": bad, "fallback_queries": int(fallback), "max_candidates": int(cand),
                            "search_s": round(t_search, 4)}
                    report["cases"].append(case)
                    print(json.dumps(case), flush=True)

    def make_masks():
        m50 = rng.integers(0, 2**64, size=words, dtype=np.uint64)
        m01 = np.zeros(words, dtype=np.uint64)
        sel = rng.choice(n, size=n // 100, replace=False)
        np.bitwise_or.at(m01, sel // 64, np.uint64(1) << (sel % 64).astype(np.uint64))
        return [("none", None), ("0.5", m50), ("0.01", m01)]

    masks = make_masks()

    g = torch.Generator(device=dev)
    g.manual_seed(11)

    def fill_chunks(idx, make):
        chunk = 1_000_000
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            rows = make(r0, m).contiguous()
            idx.upload_device(rows, row0=r0)
            torch.cuda.synchronize()
            del rows

    # --- iid corpus (the bench's) --------------------------------------------------------------
    if "iid" in corpora:
        with GpuFlatIndex(d, n, row_base=0, device=0, wide_rows=args.wide_rows) as idx:
            idx.set_mirror(args.mirror)
            idx.fill_synthetic(20240601, n)
            q = synth_rows(777, 0, nq, d)
            # half the queries ARE corpus rows (self-match at the top), the rest fresh
            q[: nq // 2] = synth_rows(20240601, 12345, nq // 2, d)
            run_cases(idx, "iid", q, masks)

    # --- clustered corpus ----------------------------------------------------------------------
    if "clustered" in corpora:
        centres = torch.randn(args.clusters, d, device=dev, generator=g)
        noise = torch.tensor([1e-4, 1e-2, 0.3], device=dev)  # near-duplicates, a tight shell, a loose shell

        def clustered(r0, m):
            which = torch.randint(0, args.clusters, (m,), device=dev, generator=g)
            scale = noise[torch.randint(0, 3, (m,), device=dev, generator=g)]
            return centres[which] + scale[:, None] * torch.randn(m, d, device=dev, generator=g)

        with GpuFlatIndex(d, n, row_base=0, device=0, wide_rows=args.wide_rows) as idx:
            idx.set_mirror(args.mirror)
            fill_chunks(idx, clustered)
            q = (centres[torch.arange(nq, device=dev) % args.clusters]
                 + 1e-3 * torch.randn(nq, d, device=dev, generator=g)).cpu().numpy().astype(np.float32)
            run_cases(idx, "clustered", q, masks)

    # --- duplicated corpus: every vector stored ~100 times (exact ties at every rank) ---------------
    if "duplicated" in corpora:
        base_rows = 100_003
        base = torch.randn(base_rows, d, device=dev, generator=g)
        with GpuFlatIndex(d, n, row_base=0, device=0, wide_rows=args.wide_rows) as idx:
            idx.set_mirror(args.mirror)
            fill_chunks(idx, lambda r0, m: base[(torch.arange(r0, r0 + m, device=dev) * 7919) % base_rows])
            q = base[:nq].cpu().numpy().astype(np.float32)
            q[nq // 2:] += 0.05 * rng.standard_normal((nq - nq // 2, d)).astype(np.float32)
            run_cases(idx, "duplicated", q, masks)

    report["not_certified_total"] = bad_total
    print(json.dumps({"not_certified_total": bad_total}))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
