"""The reference's own known-answer tests for the SIMILAR TOP-K path, restated against the CPU oracle
(C restatement and numpy twin).  Each test cites the reference test it restates
(vector_engine/src/lib.rs unless noted).  These pin the oracle at the level the reference pins itself:
tolerance-based KATs — bit-level lane order is pinned by tests/test_oracle_crosscheck.py
and tests/golden/.
"""
import numpy as np
import pytest

from oracle import oracle_c as oc
from oracle import oracle_np as on

COS, EUC, DOT = 0, 1, 2
F = np.float32


def normalize(v):
    """tests::normalize (lib.rs:4040-4047): sequential f32 sum of squares, sqrt, divide."""
    v = np.asarray(v, dtype=F)
    s = F(0)
    for x in v:
        s = F(s + x * x)
    mag = np.sqrt(s)
    return v if mag == 0 else (v / mag).astype(F)


def create_test_vector(dim, seed):
    """tests::create_test_vector (lib.rs:4029-4038).  Uses f32 sin, hence only top-1/self-match is pinned."""
    i = np.arange(dim, dtype=np.int64)
    x = (seed * 31 + i * 17).astype(F)
    return (np.sin(x * F(0.0001), dtype=F) * ((seed + i).astype(F) * F(0.001))).astype(F)


def both(A, q, k, metric):
    A = np.asarray(A, dtype=F)
    r1, s1 = oc.search(A, q, k, metric)
    r2, s2 = on.search(A, q, k, metric)
    assert np.array_equal(r1, r2) and np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
    return r1, s1


def test_search_similar_basic():  # lib.rs:4119-4135
    A = [[1, 0, 0], [0, 1, 0], [1, 1, 0]]
    rows, scores = both(A, [1, 0, 0], 3, COS)
    assert len(rows) == 3 and rows[0] == 0 and abs(scores[0] - 1.0) < 1e-6


def test_cosine_identical_orthogonal_opposite():  # lib.rs:4182-4206
    assert abs(oc.compute_similarity([1, 2, 3], [1, 2, 3]) - 1.0) < 1e-6
    assert abs(oc.compute_similarity([1, 0], [0, 1])) < 1e-6
    assert abs(oc.compute_similarity([1, 0], [-1, 0]) + 1.0) < 1e-6


def test_cosine_normalized_45deg():  # lib.rs:4208-4216
    a, b = normalize([1, 0]), normalize([1, 1])
    assert abs(oc.compute_similarity(a, b) - np.sqrt(F(2)) / 2) < 1e-6


def test_cosine_zero_vectors_are_zero_not_nan():  # lib.rs:4226-4239
    assert oc.compute_similarity([0, 0], [1, 0]) == 0.0
    s = oc.compute_similarity([0, 0], [0, 0])
    assert s == 0.0 and not np.isnan(s)
    # a zero ROW still appears in the results with score 0.0 (cosine_similarity, lib.rs:2261-2263)
    rows, scores = both([[0, 0], [1, 0]], [1, 0], 5, COS)
    assert list(rows) == [1, 0] and scores[1] == 0.0


def test_store_10000_vectors_search():  # lib.rs:4255-4276
    dim = 128
    A = np.stack([create_test_vector(dim, i) for i in range(10000)])
    rows, scores = both(A, create_test_vector(dim, 5000), 5, COS)
    assert len(rows) == 5 and rows[0] == 5000 and abs(scores[0] - 1.0) < 1e-5


@pytest.mark.parametrize("dim,qseed,k", [(768, 50, 3), (1536, 75, 5)])  # lib.rs:4278-4312
def test_high_dimensional(dim, qseed, k):
    A = np.stack([create_test_vector(dim, i) for i in range(100)])
    rows, _ = both(A, create_test_vector(dim, qseed), k, COS)
    assert len(rows) == k and rows[0] == qseed


def test_similarity_scores_mathematically_correct():  # lib.rs:4314-4352
    A = np.stack([normalize(v) for v in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [-1, 0, 0])])
    rows, scores = both(A, normalize([1, 0, 0]), 5, COS)
    got = dict(zip(rows.tolist(), scores.tolist()))
    assert abs(got[0] - 1.0) < 1e-6 and abs(got[1]) < 1e-6 and abs(got[2]) < 1e-6
    assert abs(got[3] - np.sqrt(2.0) / 2) < 1e-6 and abs(got[4] + 1.0) < 1e-6


def test_zero_query_returns_empty_for_cosine_and_dot():  # lib.rs:4458-4466, 4940-4949
    A = np.array([[1, 0]], dtype=F)
    for m in (COS, DOT):
        rows, _ = oc.search(A, [0, 0], 5, m)
        assert rows.size == 0
        assert on.search(A, [0, 0], 5, m)[0].size == 0


def test_no_embeddings():  # lib.rs:4468-4473
    rows, _ = oc.search(np.zeros((0, 2), dtype=F), [1, 0], 5, COS)
    assert rows.size == 0


def test_invalid_arguments():  # lib.rs:4926-4938
    with pytest.raises(ValueError, match="InvalidTopK"):
        oc.search(np.ones((1, 1), dtype=F), [1.0], 0, COS)
    with pytest.raises(ValueError, match="EmptyVector"):
        oc.search(np.zeros((0, 0), dtype=F), np.zeros(0, dtype=F), 5, COS)


def test_search_with_metric_cosine():  # lib.rs:4876-4890
    rows, scores = both([[1, 0], [0.707, 0.707], [0, 1]], [1, 0], 3, COS)
    assert len(rows) == 3 and rows[0] == 0 and abs(scores[0] - 1.0) < 0.01


def test_search_with_metric_dot_product():  # lib.rs:4892-4906
    rows, scores = both([[1, 0], [2, 0], [0.5, 0]], [1, 0], 3, DOT)
    assert rows[0] == 1 and abs(scores[0] - 2.0) < 0.01


def test_search_with_metric_euclidean():  # lib.rs:4908-4924
    rows, scores = both([[1, 0], [2, 0], [10, 0]], [1, 0], 3, EUC)
    assert list(rows[:2]) == [0, 1] and abs(scores[0] - 1.0) < 0.01 and abs(scores[1] - 0.5) < 0.01


def test_zero_query_euclidean_is_scored():  # lib.rs:4951-4970
    rows, scores = both([[0, 0], [1, 0], [10, 0]], [0, 0], 3, EUC)
    assert list(rows) == [0, 1, 2] and abs(scores[0] - 1.0) < 0.01 and abs(scores[1] - 0.5) < 0.01


def test_euclidean_distance_kats():  # lib.rs:4972-4995
    assert abs(oc.euclidean_seq([1, 2, 3], [1, 2, 3])) < 1e-6
    assert abs(oc.euclidean_seq([0, 0], [1, 0]) - 1.0) < 1e-6
    assert abs(oc.euclidean_seq([0, 0], [3, 4]) - 5.0) < 1e-6


def test_parallel_path_returns_k():  # lib.rs:6583-6603 (parallel_threshold: 5)
    A = np.array([[i, 0, 0] for i in range(10)], dtype=F)
    rows, _ = oc.search(A, [5, 0, 0], 3, EUC, nthreads=4, partial=True)
    assert len(rows) == 3
    assert np.array_equal(rows, oc.search(A, [5, 0, 0], 3, EUC)[0])


def test_n_less_than_k_returns_n():  # lib.rs:4137-4150 (search_similar_top_k family)
    A = np.eye(4, dtype=F)
    rows, _ = both(A, [1, 1, 0, 0], 10, COS)
    assert len(rows) == 4


def test_merge_top_k():  # query_router/src/distributed.rs:645-680
    rows = np.array([[[0, 1]], [[2, 99]]], dtype=np.uint64)          # shard0: a,b ; shard1: c
    scores = np.array([[[0.9, 0.8]], [[0.95, 0.0]]], dtype=F)
    counts = np.array([[2], [1]], dtype=np.uint32)
    r, s, c = oc.merge_topk(rows, scores, counts, 2)
    assert c[0] == 2 and list(r[0]) == [2, 0] and np.allclose(s[0], [0.95, 0.9])


def test_example_vector_search_rs():  # examples/vector_search.rs:26-67, queries :80,:96,:112 (TOP 3)
    docs = np.array([
        [0.8, 0.7, 0.1, 0.2, 0.1, 0.1, 0.1, 0.1], [0.9, 0.8, 0.2, 0.1, 0.1, 0.1, 0.1, 0.1],
        [0.85, 0.75, 0.15, 0.15, 0.1, 0.1, 0.1, 0.1], [0.1, 0.1, 0.8, 0.7, 0.2, 0.1, 0.1, 0.1],
        [0.1, 0.1, 0.75, 0.8, 0.25, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.2, 0.8, 0.7, 0.1, 0.1],
        [0.2, 0.1, 0.3, 0.3, 0.3, 0.3, 0.8, 0.7], [0.15, 0.1, 0.25, 0.25, 0.25, 0.25, 0.75, 0.8]], dtype=F)
    ml, _ = both(docs, [0.85, 0.75, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1], 3, COS)
    db, _ = both(docs, [0.1, 0.1, 0.8, 0.75, 0.2, 0.1, 0.1, 0.1], 3, COS)
    sy, _ = both(docs, [0.1, 0.1, 0.2, 0.2, 0.2, 0.2, 0.8, 0.8], 3, COS)
    assert set(ml.tolist()) == {0, 1, 2}       # the three ML documents
    assert set(db.tolist()[:2]) == {3, 4}      # both database documents first
    assert set(sy.tolist()[:2]) == {6, 7}      # both systems documents first


def test_ties_rank_by_ascending_row():
    # the reference's tie order is its HashSet scan order (slab_router.rs:287-305); ours is row-ascending
    A = np.array([[0, 1], [1, 0], [0, 2], [2, 0], [0, 3]], dtype=F)
    rows, scores = both(A, [1, 0], 5, COS)
    assert list(rows) == [1, 3, 0, 2, 4] and list(scores) == [1, 1, 0, 0, 0]


def test_mask_is_prefilter_semantics():  # search_with_pre_filter, lib.rs:3514-3557
    rng = np.random.default_rng(5)
    A = rng.standard_normal((300, 24)).astype(F)
    q = rng.standard_normal(24).astype(F)
    keep = rng.random(300) < 0.3
    rows, scores = oc.search(A, q, 10, COS, mask=oc.mask_from_bool(keep))
    r2, s2 = on.search(A, q, 10, COS, keep=keep)
    assert np.array_equal(rows, r2) and np.array_equal(scores, s2)
    sub_rows, sub_scores = oc.search(A[keep], q, 10, COS)
    assert np.array_equal(np.flatnonzero(keep)[sub_rows.astype(np.int64)], rows.astype(np.int64))
    assert np.array_equal(sub_scores, scores)


# ---- tensor_blob artifact similarity: SparseVector::cosine_similarity in f64 (sparse_vector.rs:1320-1375) ----
def test_sparse_cos64_kats():
    f = oc.sparse_cos64
    assert abs(f([1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) - 1.0) < 1e-6          # cosine_similarity_identical
    assert abs(f([1.0, 0.0], [0.0, 1.0])) < 1e-6                          # cosine_similarity_orthogonal
    assert f([1.0, 2.0, 3.0], [0.0, 0.0, 0.0]) == 0.0                     # zero_vector_returns_zero
    assert f([0.0, 0.0, 0.0], [0.0, 0.0, 0.0]) == 0.0                     # both_zero_returns_zero
    r = f([1.0, 2.0, 3.0], [1.0, 2.0, 3.0])                               # clamps_to_valid_range
    assert -1.0 <= r <= 1.0 and abs(r - 1.0) < 1e-6
    assert abs(f([1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]) + 1.0) < 1e-6         # opposite_returns_negative_one
    # sanitisation (sparse_vector.rs:593-598): NaN/Inf quotients become 0.0; f64 keeps huge f32 values finite
    assert f([np.nan, 1.0], [0.0, 1.0]) == 0.0 and f([np.inf, 1.0], [1.0, 1.0]) == 0.0
    assert f([3e38, 3e38], [3e38, 3e38]) == 1.0
    # a NaN opposite a zero is never multiplied (the position is not stored on the other side) ...
    assert f([1.0, 2.0], [np.nan, 0.0]) == 0.0      # ... but it still poisons its own magnitude
    # C restatement == numpy twin on random sparse data, bit for bit
    from oracle import oracle_np as onp
    rng = np.random.default_rng(5)
    A = rng.standard_normal((200, 37)).astype(np.float32) * (rng.random((200, 37)) < 0.4)
    q = (rng.standard_normal(37) * (rng.random(37) < 0.6)).astype(np.float32)
    a = np.array([f(q, row) for row in A], np.float32)
    assert np.array_equal(a, onp.sparse_cos64_rows(A, q))
    assert np.array_equal(a, oc.scores_all(A, q, oc.SPARSE_COS64))


# ---- router-level cases (integration_tests/tests/distance_metrics.rs:37-140: `SIMILAR ... LIMIT 3 [COSINE|EUCLIDEAN|DOT_PRODUCT]`):
# the data and the assertions of those tests at the engine boundary the router calls (query_router/src/lib.rs:5429-5439)
ROUTER_CASES = [  # (name, rows, query, metric, expected first key[, expected second key])
    ("default_metric", {"vec:1": [1.0, 0.0, 0.0, 0.0], "vec:2": [0.9, 0.1, 0.0, 0.0], "vec:3": [0.0, 1.0, 0.0, 0.0]}, "vec:1", COS, "vec:1", "vec:2"),
    ("cosine", {"cos:1": [1.0, 0.0, 0.0, 0.0], "cos:2": [0.707, 0.707, 0.0, 0.0], "cos:3": [0.0, 1.0, 0.0, 0.0]}, "cos:1", COS, "cos:1", None),
    ("euclidean", {"euc:1": [0.5, 0.5, 0.5, 0.5], "euc:2": [0.6, 0.5, 0.5, 0.5], "euc:3": [1.0, 1.0, 1.0, 1.0]}, "euc:1", EUC, "euc:1", "euc:2"),
    ("dot_product", {"dot:1": [1.0, 0.0, 0.0, 0.0], "dot:2": [0.5, 0.5, 0.0, 0.0], "dot:3": [0.0, 0.0, 1.0, 0.0]}, "dot:1", DOT, "dot:1", "dot:2"),
    ("vector_with_cosine", {"target:1": [1.0, 0.0, 0.0, 0.0], "target:2": [0.8, 0.2, 0.0, 0.0], "target:3": [0.0, 0.0, 1.0, 0.0]},
     [1.0, 0.0, 0.0, 0.0], COS, "target:1", "target:2"),
    # query_router/src/lib.rs:9436-9524 (parsed_similar_{cosine,euclidean,euclidean_zero_query,dot_product}_metric)
    ("router_cosine", {"cos_a": [1.0, 0.0], "cos_b": [0.0, 1.0], "cos_c": [0.707, 0.707]}, [1.0, 0.0], COS, "cos_a", "cos_c"),
    ("router_euclidean", {"euc_a": [1.0, 0.0], "euc_b": [2.0, 0.0], "euc_c": [10.0, 0.0]}, [1.0, 0.0], EUC, "euc_a", "euc_b"),
    ("router_euclidean_zero_query", {"zero_origin": [0.0, 0.0], "zero_unit": [1.0, 0.0], "zero_far": [10.0, 0.0]}, [0.0, 0.0], EUC, "zero_origin", "zero_unit"),
    ("router_dot_product", {"dot_a": [1.0, 0.0], "dot_b": [2.0, 0.0], "dot_c": [0.5, 0.0]}, [1.0, 0.0], DOT, "dot_b", "dot_a"),
]


@pytest.mark.parametrize("case", ROUTER_CASES, ids=[c[0] for c in ROUTER_CASES])
def test_router_level_similar_cases(case):  # integration_tests/tests/distance_metrics.rs:37-140
    _, rows, query, metric, first, second = case
    keys = list(rows)
    A = [rows[k] for k in keys]
    q = rows[query] if isinstance(query, str) else query
    r, s = both(A, q, 3, metric)
    assert len(r) == 3 and keys[r[0]] == first
    if second is not None:
        assert keys[r[1]] == second
    assert all(s[i] >= s[i + 1] for i in range(2))


def test_empty_store_then_single_embedding():  # integration_tests/tests/edge_cases.rs:80-96
    q = (np.arange(32, dtype=F) * F(0.03125)).astype(F)
    r, s = both(np.zeros((0, 32), F), q, 10, COS)
    assert len(r) == 0
    r, s = both([q], q, 10, COS)
    assert list(r) == [0] and abs(s[0] - 1.0) < 1e-6


def test_zero_vector_is_stored_and_searched_gracefully():  # integration_tests/tests/edge_cases.rs:98-127
    z = np.zeros(32, F)
    r, s = both([z], z, 10, EUC)         # (Euclidean scores a zero query: distance 0 -> 1.0)
    assert list(r) == [0] and s[0] == 1.0
    assert oc.compute_similarity(z, z) == 0.0   # cosine of two zero vectors: 0.0, not NaN (lib.rs:2257-2266)


def _sparse_rows():  # integration_tests/tests/sparse_vectors.rs:205-232 (values: f32 sin; only the result count is pinned there)
    A = np.zeros((10, 64), F)
    for i in range(10):
        for j in range(10):
            A[i, (i * 5 + j) % 64] = np.sin(F((i + j)) * F(0.1), dtype=F)
    q = np.zeros(64, F)
    for j in range(10):
        q[j % 64] = np.sin(F(j) * F(0.1), dtype=F)
    return A, q


def test_sparse_vector_in_similarity_search():  # sparse_vectors.rs:205-232
    A, q = _sparse_rows()
    r, s = both(A, q, 5, COS)
    assert len(r) == 5 and r[0] == 0  # (row 0 holds the query's own values)


def test_vector_engine_sparse_search():  # sparse_vectors.rs:412-438
    A = np.zeros((3, 100), F)
    A[0, 0] = 1.0
    A[1, 0] = A[1, 1] = 0.707
    A[2, 1] = 1.0
    q = np.zeros(100, F)
    q[0] = 1.0
    r, s = both(A, q, 3, COS)
    assert list(r) == [0, 1, 2] and s[0] == 1.0 and s[2] == 0.0
