"""The fixtures of the reference-side parity harness (integration/rust/tests/parity.rs, never compiled here: no cargo) must
be what the oracle says — otherwise a maintainer running the harness against the real crate would be chasing this
repository's bug.  Every case is re-derived with the C oracle; the file set must be what export_rust_fixtures.py writes."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "integration", "rust", "tests", "fixtures")
METRIC = {"Cosine": 0, "Euclidean": 1, "DotProduct": 2}


@pytest.mark.parametrize("name", ["small_explicit_300x40", "tail_200x100", "wide_256x768"])
def test_fixture_cases_are_the_oracles(name):
    doc = json.load(open(os.path.join(FIX, name + ".json")))
    n, d = doc["n"], doc["dim"]
    A = np.fromfile(os.path.join(FIX, name + ".f32le"), dtype="<f4").reshape(n, d)
    Q = np.array(doc["queries_bits"], dtype=np.uint32).view(np.float32)
    assert len(doc["cases"]) >= 6
    for c in doc["cases"]:
        keep = None
        if c["keep"]:
            keep = np.zeros(n, bool)
            keep[doc["keep"][c["keep"]]] = True
        er, es = oc.search(A, Q[c["query"]], c["k"], METRIC[c["metric"]], mask=None if keep is None else oc.mask_from_bool(keep))
        assert [int(x) for x in er] == c["rows"]
        assert [int(x) for x in es.view(np.uint32)] == c["score_bits"]
        # tied_at_cut: exactly the participating rows whose exact score equals the k-th
        allr, alls = oc.search(A, Q[c["query"]], n, METRIC[c["metric"]], mask=None if keep is None else oc.mask_from_bool(keep))
        tied = sorted(int(r) for r, s in zip(allr, alls) if s.view(np.uint32) == es[-1].view(np.uint32))
        assert sorted(c["tied_at_cut"]) == tied and set(c["rows"]) & set(tied)


def test_harness_names_every_fixture_set_and_the_calls_it_checks():
    src = open(os.path.join(ROOT, "integration", "rust", "tests", "parity.rs")).read()
    for name in ("small_explicit_300x40", "tail_200x100", "wide_256x768"):
        assert f'run_set("{name}")' in src
    for call in ("search_similar_with_metric", "search_similar_filtered", "FilteredSearchConfig::pre_filter()", "to_bits()"):
        assert call in src
