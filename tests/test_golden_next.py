"""The oracles of the widened rows against their committed golden fixtures (tests/golden/make_golden_next.py):
guards the restatements themselves against drift.  CPU only."""
import numpy as np

from oracle import filter_oracle as fo
from oracle import ivf_oracle as io
from oracle import oracle_c as oc
from oracle import oracle_np as on
from tests import _golden as G


def test_filter_oracle_matches_golden():
    rows, cases = G.load_filters()
    assert len(rows) == 300 and len(cases) == 80
    for cond, selected in cases:
        assert [i for i, r in enumerate(rows) if fo.evaluate(r, cond)] == selected, cond


def test_ivf_oracle_matches_golden():
    g = G.load("ivf_flat_small.npz")
    V, Q = g["V"], g["Q"]
    for init, tag in (("random", "rnd"), ("kmeans++", "pp")):
        ivf = io.IVFFlat(10, nprobe=3, kmeans=io.KMeansConfig(8, 1e-4, 4242, init))
        ivf.train(V)
        assert np.array_equal(ivf.centroids, g[f"centroids_{tag}"])
        for v in V:
            ivf.add(v)
        assert np.array_equal(np.array(ivf.assign, np.uint32), g[f"assign_{tag}"])
        for qi in range(6):
            for nprobe in (1, 3, 10):
                ids, dist = ivf.search(Q[qi], 15, nprobe)
                assert np.array_equal(np.array(ids, np.uint64), g[f"ids_{tag}_q{qi}_p{nprobe}"])
                assert np.array_equal(dist, g[f"dist_{tag}_q{qi}_p{nprobe}"])


def test_sparse_cos64_oracles_match_golden():
    g = G.load("sparse_cos64_small.npz")
    A, Q = g["A"], g["Q"]
    for qi in range(5):
        for mod in (oc, on):
            r, s = mod.search(A, Q[qi], 20, mod.SPARSE_COS64)
            assert np.array_equal(r, g[f"rows_q{qi}"])
            assert np.array_equal(s.view(np.uint32), g[f"scores_q{qi}"].view(np.uint32))
    assert g["scores_q3"][:2].tolist() == [1.0, 1.0] and g["rows_q3"][:2].tolist() == [30, 31]   # duplicates of the query
    assert g["rows_q4"][0] == 40 and g["scores_q4"][0] == 1.0                                      # the 1e30-scaled row
