#!/usr/bin/env python
"""Export parity fixtures for the REFERENCE-SIDE harness integration/rust/tests/parity.rs (run from the repo root).

SURVEY.md §8c: the reference's own tests pin this path with tolerances only, and there is no cargo in this image, so
bit-level parity with the real crate is pinned by the cited source text alone.  This script writes what a maintainer
WITH cargo needs to close that gap: explicit corpora and queries as raw little-endian f32, and — per case — the keys and
the score BIT PATTERNS the oracle (oracle/oracle_np.py, cross-checked against oracle/nmn_oracle.c here) expects
`VectorEngine::search_similar_with_metric` / `search_similar_filtered` to return.  Data only; nothing of the reference.

  integration/rust/tests/fixtures/<set>.f32le     n x dim corpus, row-major
  integration/rust/tests/fixtures/<set>.json      dim, n, queries (u32 bit patterns), keep-columns, cases[]
A case: metric, query index, k, optional keep column, rows[], score_bits[], and tied_at_cut[] — every row whose exact score
equals the k-th (the reference's order among equal scores is its HashSet's, lib.rs:2027-2034 + slab_router.rs:287-305)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c as oc  # noqa: E402
from oracle import oracle_np as on  # noqa: E402

OUT = os.path.join(ROOT, "integration", "rust", "tests", "fixtures")
F = np.float32
METRIC = {0: "Cosine", 1: "Euclidean", 2: "DotProduct"}


def case(A, Q, qi, k, m, keep=None, keep_name=None):
    r, s = on.search(A, Q[qi], k, m, keep=keep)
    r2, s2 = oc.search(A, Q[qi], k, m, mask=None if keep is None else oc.mask_from_bool(keep))
    assert np.array_equal(r, r2) and np.array_equal(s.view(np.uint32), s2.view(np.uint32)), "C vs numpy oracle"
    n = A.shape[0]
    allr, alls = on.search(A, Q[qi], n, m, keep=keep)            # every participating row, to find the ties at the cut
    tied = [int(x) for x, sc in zip(allr, alls) if r.size and sc.view(np.uint32) == s[-1].view(np.uint32)]
    return {"metric": METRIC[m], "query": qi, "k": k, "keep": keep_name, "rows": [int(x) for x in r],
            "score_bits": [int(x) for x in s.view(np.uint32)], "tied_at_cut": tied}


def write(name, A, Q, keeps, cases):
    A.astype("<f4").tofile(os.path.join(OUT, name + ".f32le"))
    doc = {"set": name, "n": int(A.shape[0]), "dim": int(A.shape[1]),
           "queries_bits": [[int(x) for x in q.view(np.uint32)] for q in Q],
           "keep": {k: [int(i) for i in np.flatnonzero(v)] for k, v in keeps.items()}, "cases": cases}
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(name, A.shape, len(cases), "cases")


def main():
    os.makedirs(OUT, exist_ok=True)
    # 1. the committed small_explicit set: 300 x 40 (five whole 8-lane chunks), a zero row, triplicated rows, 50 % / 10 % filters
    g = np.load(os.path.join(ROOT, "tests", "golden", "small_explicit.npz"))
    A, Q = g["A"], g["Q"]
    keeps = {"keep50": g["keep50"], "keep10": g["keep10"]}
    cases = []
    for m in (0, 1, 2):
        for qi in range(4):
            c = case(A, Q, qi, 10, m)
            assert np.array_equal(np.array(c["rows"], np.uint64), g[f"rows_m{m}_q{qi}_all"])  # the committed fixture itself
            cases.append(c)
    for qi in range(4):   # filtered search is always cosine in the reference (lib.rs:3429-3475)
        for tag in ("keep50", "keep10"):
            cases.append(case(A, Q, qi, 10, 0, keeps[tag], tag))
    write("small_explicit_300x40", A, Q, keeps, cases)
    # 2. the scalar tail of simd::dot_product (hnsw.rs:186-190): dim 100 = 12 chunks of 8 + 4 tail elements
    rng = np.random.default_rng(20260929)
    A = rng.standard_normal((200, 100)).astype(F)
    Q = rng.standard_normal((3, 100)).astype(F)
    cases = [case(A, Q, qi, 16, m) for m in (0, 1, 2) for qi in range(3)]
    write("tail_200x100", A, Q, {}, cases)
    # 3. the dimension of the BASELINE configs: 768 = 96 chunks, values spread over several binades, planted near-duplicates
    A = (rng.standard_normal((256, 768)) * np.exp2(rng.integers(-3, 4, (256, 1)))).astype(F)
    Q = rng.standard_normal((2, 768)).astype(F)
    for t in range(8):
        A[40 + t] = Q[0] * F(1.0 + 1e-6 * t) + F(1e-3) * rng.standard_normal(768).astype(F)
    cases = [case(A, Q, qi, 20, m) for m in (0, 1, 2) for qi in range(2)]
    write("wide_256x768", A, Q, {}, cases)


if __name__ == "__main__":
    main()
