#!/usr/bin/env python
"""Generate the golden fixtures of the widened rows under tests/golden/ — run from the
repo root.  Same rules as make_golden.py: data only (explicit inputs + expected outputs), produced by the
oracles (oracle/filter_oracle.py, oracle/ivf_oracle.py, oracle/oracle_np.py + the C oracle) because the Rust
reference cannot be built or imported here.

  filters_mixed.json      300 rows of typed metadata, 80 conditions, expected selection per condition
  ivf_flat_small.npz      600 x 24 vectors, k-means (both inits) centroids, assignments, probe results
  sparse_cos64_small.npz  400 x 40 half-sparse vectors, f64 sparse-cosine TOP-20 per query
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import filter_oracle as fo  # noqa: E402
from oracle import ivf_oracle as io  # noqa: E402
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_np as on  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F = np.float32
WORDS = ["", "a", "ab", "abc", "electronics", "clothing", "food", "Food", "é", "zeta"]


def enc(v):  # JSON has no NaN/Inf/int-vs-float distinction: tag every value
    if v is None:
        return ["null"]
    if isinstance(v, bool):
        return ["bool", v]
    if isinstance(v, int):
        return ["int", str(v)]
    if isinstance(v, float):
        return ["float", v.hex()]
    return ["str", v]


def enc_cond(c):
    op = c[0]
    if op in ("and", "or"):
        return [op, enc_cond(c[1]), enc_cond(c[2])]
    if op == "in":
        return [op, c[1], [enc(v) for v in c[2]]]
    if op in ("eq", "ne", "lt", "le", "gt", "ge"):
        return [op, c[1], enc(c[2])]
    return list(c)


def filters():
    rng = np.random.default_rng(41)

    def value(field):
        if field == "price":
            return int(rng.integers(-5, 60))
        if field == "score":
            return [0.5, 0.95, -0.0, float("nan"), float("inf"), 49.5, 50.0][int(rng.integers(0, 7))]
        if field == "category":
            return WORDS[int(rng.integers(0, len(WORDS)))]
        if field == "active":
            return bool(rng.integers(0, 2))
        if field == "opt":
            return None
        return [None, True, 7, 7.0, "7", 2**53 + 1][int(rng.integers(0, 6))]

    fields = ["price", "score", "category", "active", "opt", "mixed"]
    rows = [{f: value(f) for f in fields if rng.random() < 0.7} for _ in range(300)]

    def fval():
        return [None, True, False, int(rng.integers(-5, 60)), 50.0, 49.5, float("nan"),
                WORDS[int(rng.integers(0, len(WORDS)))], float(2**53)][int(rng.integers(0, 9))]

    def cond(depth=0):
        r = rng.random()
        if depth < 3 and r < 0.35:
            return ("and" if rng.random() < 0.5 else "or", cond(depth + 1), cond(depth + 1))
        f = (fields + ["nosuchfield"])[int(rng.integers(0, len(fields) + 1))]
        r = rng.random()
        if r < 0.05:
            return ("true",)
        if r < 0.15:
            return ("exists", f)
        if r < 0.25:
            return ("contains", f, ["", "a", "oo", "é"][int(rng.integers(0, 4))])
        if r < 0.35:
            return ("startswith", f, ["", "a", "F", "é"][int(rng.integers(0, 4))])
        if r < 0.5:
            return ("in", f, [fval() for _ in range(int(rng.integers(0, 4)))])
        return (["eq", "ne", "lt", "le", "gt", "ge"][int(rng.integers(0, 6))], f, fval())

    conds = [cond() for _ in range(80)]
    doc = {"rows": [{k: enc(v) for k, v in r.items()} for r in rows],
           "cases": [{"cond": enc_cond(c), "selected": [i for i, r in enumerate(rows) if fo.evaluate(r, c)]} for c in conds]}
    with open(os.path.join(OUT, "filters_mixed.json"), "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))


def ivf():
    rng = np.random.default_rng(43)
    n, d, c = 600, 24, 10
    V = (rng.standard_normal((n, d)) + 3.0 * rng.integers(0, 3, (n, 1))).astype(F)   # three loose blobs
    V[77] = V[5]
    Q = rng.standard_normal((6, d)).astype(F) + F(3.0)
    out = {"V": V, "Q": Q}
    for init in ("random", "kmeans++"):
        tag = "rnd" if init == "random" else "pp"
        ivf_ = io.IVFFlat(c, nprobe=3, kmeans=io.KMeansConfig(8, 1e-4, 4242, init))
        ivf_.train(V)
        for v in V:
            ivf_.add(v)
        out[f"centroids_{tag}"] = ivf_.centroids
        out[f"assign_{tag}"] = np.array(ivf_.assign, np.uint32)
        for qi in range(6):
            for nprobe in (1, 3, c):
                ids, dist = ivf_.search(Q[qi], 15, nprobe)
                out[f"ids_{tag}_q{qi}_p{nprobe}"] = np.array(ids, np.uint64)
                out[f"dist_{tag}_q{qi}_p{nprobe}"] = dist
    np.savez_compressed(os.path.join(OUT, "ivf_flat_small.npz"), **out)


def sparse_cos64():
    rng = np.random.default_rng(47)
    A = (rng.standard_normal((400, 40)) * (rng.random((400, 40)) < 0.5)).astype(F)
    A[9] = 0.0
    A[30] = A[31]
    A[32] = -A[31]
    A[40] *= F(1e30)
    A[41] *= F(1e-30)
    A[42, 3] = np.nan
    Q = (rng.standard_normal((5, 40)) * (rng.random((5, 40)) < 0.7)).astype(F)
    Q[3] = A[31]
    Q[4] = (A[40] / F(1e30)).astype(F)
    out = {"A": A, "Q": Q}
    for qi in range(5):
        r, s = on.search(A, Q[qi], 20, on.SPARSE_COS64)
        r2, s2 = oc.search(A, Q[qi], 20, oc.SPARSE_COS64)
        assert np.array_equal(r, r2) and np.array_equal(s.view(np.uint32), s2.view(np.uint32)), "C vs numpy oracle"
        out[f"rows_q{qi}"] = r
        out[f"scores_q{qi}"] = s
    np.savez_compressed(os.path.join(OUT, "sparse_cos64_small.npz"), **out)


if __name__ == "__main__":
    filters()
    ivf()
    sparse_cos64()
    print("written:", sorted(f for f in os.listdir(OUT) if f.startswith(("filters_", "ivf_", "sparse_"))))
