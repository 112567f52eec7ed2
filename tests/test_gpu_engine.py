"""The reference's own VectorEngine tests for the SIMILAR path, restated against the host-side mirror
(neumann_amd.engine.VectorEngine -> C++ nmn_engine -> libneumann_gpu C ABI -> HIP kernels).
Each test cites the reference test it follows (vector_engine/src/lib.rs)."""
import threading

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture
def E():
    from neumann_amd import engine
    return engine


def normalize(v):  # tests::normalize (lib.rs:4040-4047)
    v = np.asarray(v, dtype=F)
    s = F(0)
    for x in v:
        s = F(s + x * x)
    mag = np.sqrt(s)
    return v if mag == 0 else (v / mag).astype(F)


def create_test_vector(dim, seed):  # tests::create_test_vector (lib.rs:4029-4038)
    i = np.arange(dim, dtype=np.int64)
    x = (seed * 31 + i * 17).astype(F)
    return (np.sin(x * F(0.0001), dtype=F) * ((seed + i).astype(F) * F(0.001))).astype(F)


# ---- basic CRUD + search (lib.rs:4049-4180) ---------------------------------------------------------
def test_store_and_retrieve_embedding(E):  # lib.rs:4049-4059
    engine = E.VectorEngine()
    engine.store_embedding("test", [1.0, 2.0, 3.0])
    assert np.array_equal(engine.get_embedding("test"), np.array([1, 2, 3], F))


def test_store_overwrites_existing(E):  # lib.rs:4061-4070
    engine = E.VectorEngine()
    engine.store_embedding("key", [1.0, 2.0])
    engine.store_embedding("key", [3.0, 4.0])
    assert np.array_equal(engine.get_embedding("key"), np.array([3, 4], F))
    assert engine.count() == 1


def test_delete_embedding_and_errors(E):  # lib.rs:4072-4100
    engine = E.VectorEngine()
    engine.store_embedding("key", [1.0, 2.0])
    engine.delete_embedding("key")
    assert not engine.exists("key")
    with pytest.raises(E.VectorError) as e:
        engine.delete_embedding("missing")
    assert e.value.kind == "NotFound" and str(e.value) == "Embedding not found: missing"
    with pytest.raises(E.VectorError) as e:
        engine.get_embedding("missing")
    assert e.value.kind == "NotFound"
    with pytest.raises(E.VectorError) as e:
        engine.store_embedding("empty", [])
    assert e.value.kind == "EmptyVector" and str(e.value) == "Empty vector provided"


def test_search_similar_basic(E):  # lib.rs:4119-4135
    engine = E.VectorEngine()
    engine.store_embedding("a", [1.0, 0.0, 0.0])
    engine.store_embedding("b", [0.0, 1.0, 0.0])
    engine.store_embedding("c", [1.0, 1.0, 0.0])
    results = engine.search_similar([1.0, 0.0, 0.0], 3)
    assert len(results) == 3
    assert results[0].key == "a" and abs(results[0].score - 1.0) < 1e-6


def test_search_similar_top_k_and_errors(E):  # lib.rs:4137-4180
    engine = E.VectorEngine()
    for i in range(10):
        engine.store_embedding(f"v{i}", [float(i), 1.0])
    assert len(engine.search_similar([1.0, 1.0], 3)) == 3
    with pytest.raises(E.VectorError) as e:
        engine.search_similar([1.0, 1.0], 0)
    assert e.value.kind == "InvalidTopK" and "Invalid top_k" in str(e.value)
    with pytest.raises(E.VectorError) as e:
        engine.search_similar([], 5)
    assert e.value.kind == "EmptyVector"


def test_compute_similarity_kats(E):  # lib.rs:4182-4239
    engine = E.VectorEngine()
    assert abs(engine.compute_similarity([1, 2, 3], [1, 2, 3]) - 1.0) < 1e-6
    assert abs(engine.compute_similarity([1, 0], [0, 1])) < 1e-6
    assert abs(engine.compute_similarity([1, 0], [-1, 0]) + 1.0) < 1e-6
    a, b = normalize([1, 0]), normalize([1, 1])
    assert abs(engine.compute_similarity(a, b) - np.sqrt(F(2)) / 2) < 1e-6
    with pytest.raises(E.VectorError) as e:
        engine.compute_similarity([1, 2], [1, 2, 3])
    assert e.value.kind == "DimensionMismatch" and str(e.value) == "Dimension mismatch: expected 2, got 3"
    assert engine.compute_similarity([0, 0], [1, 0]) == 0.0
    s = engine.compute_similarity([0, 0], [0, 0])
    assert s == 0.0 and not np.isnan(s)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(333).astype(F), rng.standard_normal(333).astype(F)
    assert engine.compute_similarity(x, y) == float(oc.compute_similarity(x, y))  # bit-exact vs the oracle


def test_search_skips_dimension_mismatch(E):  # lib.rs:4241-4253
    engine = E.VectorEngine()
    engine.store_embedding("2d", [1.0, 0.0])
    engine.store_embedding("3d", [1.0, 0.0, 0.0])
    results = engine.search_similar([1.0, 0.0], 10)
    assert len(results) == 1 and results[0].key == "2d"


def test_store_10000_vectors_search(E):  # lib.rs:4255-4276
    engine = E.VectorEngine()
    dim = 128
    engine.batch_store_embeddings([f"v{i}" for i in range(10000)],
                                  np.stack([create_test_vector(dim, i) for i in range(10000)]))
    assert engine.count() == 10000
    results = engine.search_similar(create_test_vector(dim, 5000), 5)
    assert len(results) == 5
    assert results[0].key == "v5000" and abs(results[0].score - 1.0) < 1e-5


@pytest.mark.parametrize("dim,qseed,k", [(768, 50, 3), (1536, 75, 5), (4096, 10, 3)])  # lib.rs:4278-4312, 6021
def test_high_dimensional(E, dim, qseed, k):
    engine = E.VectorEngine()
    for i in range(100):
        engine.store_embedding(f"v{i}", create_test_vector(dim, i))
    results = engine.search_similar(create_test_vector(dim, qseed), k)
    assert len(results) == k and results[0].key == f"v{qseed}"


def test_similarity_scores_mathematically_correct(E):  # lib.rs:4314-4352
    engine = E.VectorEngine()
    engine.store_embedding("unit_x", normalize([1, 0, 0]))
    engine.store_embedding("unit_y", normalize([0, 1, 0]))
    engine.store_embedding("unit_z", normalize([0, 0, 1]))
    engine.store_embedding("diag_xy", normalize([1, 1, 0]))
    engine.store_embedding("neg_x", normalize([-1, 0, 0]))
    results = engine.search_similar(normalize([1, 0, 0]), 5)
    assert len(results) == 5
    for r in results:
        if r.key == "unit_x":
            assert abs(r.score - 1.0) < 1e-6
        elif r.key in ("unit_y", "unit_z"):
            assert abs(r.score) < 1e-6
        elif r.key == "diag_xy":
            assert abs(r.score - np.sqrt(2.0) / 2) < 1e-6
        elif r.key == "neg_x":
            assert abs(r.score + 1.0) < 1e-6
        else:
            raise AssertionError(r.key)


def test_zero_query_and_empty_engine(E):  # lib.rs:4458-4473
    engine = E.VectorEngine()
    assert engine.search_similar([1.0, 0.0], 5) == []
    engine.store_embedding("a", [1.0, 0.0])
    assert engine.search_similar([0.0, 0.0], 5) == []


def test_search_with_metric(E):  # lib.rs:4876-4970
    M = E.DistanceMetric
    engine = E.VectorEngine()
    engine.store_embedding("a", [1.0, 0.0])
    engine.store_embedding("b", [0.707, 0.707])
    engine.store_embedding("c", [0.0, 1.0])
    r = engine.search_similar_with_metric([1.0, 0.0], 3, M.Cosine)
    assert len(r) == 3 and r[0].key == "a" and abs(r[0].score - 1.0) < 0.01

    engine = E.VectorEngine()
    engine.store_embedding("a", [1.0, 0.0])
    engine.store_embedding("b", [2.0, 0.0])
    engine.store_embedding("c", [0.5, 0.0])
    r = engine.search_similar_with_metric([1.0, 0.0], 3, M.DotProduct)
    assert r[0].key == "b" and abs(r[0].score - 2.0) < 0.01

    engine = E.VectorEngine()
    engine.store_embedding("a", [1.0, 0.0])
    engine.store_embedding("b", [2.0, 0.0])
    engine.store_embedding("c", [10.0, 0.0])
    r = engine.search_similar_with_metric([1.0, 0.0], 3, M.Euclidean)
    assert [x.key for x in r[:2]] == ["a", "b"] and abs(r[0].score - 1.0) < 0.01 and abs(r[1].score - 0.5) < 0.01

    with pytest.raises(E.VectorError) as e:
        engine.search_similar_with_metric([], 5, M.Cosine)
    assert e.value.kind == "EmptyVector"
    with pytest.raises(E.VectorError) as e:
        engine.search_similar_with_metric([1.0], 0, M.Cosine)
    assert e.value.kind == "InvalidTopK"
    assert engine.search_similar_with_metric([0.0, 0.0], 5, M.Cosine) == []

    engine = E.VectorEngine()  # zero query is valid for Euclidean (lib.rs:4951-4970)
    engine.store_embedding("origin", [0.0, 0.0])
    engine.store_embedding("unit", [1.0, 0.0])
    engine.store_embedding("far", [10.0, 0.0])
    r = engine.search_similar_with_metric([0.0, 0.0], 3, M.Euclidean)
    assert [x.key for x in r] == ["origin", "unit", "far"]
    assert abs(r[0].score - 1.0) < 0.01 and abs(r[1].score - 0.5) < 0.01


def test_engine_matches_oracle_on_random_corpus(E):
    rng = np.random.default_rng(21)
    n, d, k = 6000, 96, 40
    A = rng.standard_normal((n, d)).astype(F)
    A[rng.integers(0, n, 600)] *= 0.0  # zero rows (stored sparse by the reference; scored 0.0)
    sp = rng.integers(0, n, 800)
    A[sp, : d // 2 + 8] = 0.0          # >= 50 % zeros -> TensorValue::Sparse in the reference: exact either way
    q = rng.standard_normal(d).astype(F)
    engine = E.VectorEngine()
    engine.batch_store_embeddings([f"k{i}" for i in range(n)], A)
    for metric in E.DistanceMetric:
        res = engine.search_similar_with_metric(q, k, metric)
        er, es = oc.search(A, q, k, int(metric))
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)


# ---- max_dimension / timeout (lib.rs:1960-1967, 2005-2024) --------------------------------------------
def test_max_dimension_and_timeout(E):
    engine = E.VectorEngine(E.VectorEngineConfig(max_dimension=4))
    with pytest.raises(E.VectorError) as e:
        engine.store_embedding("big", [1.0] * 5)
    assert e.value.kind == "DimensionMismatch" and str(e.value) == "Dimension mismatch: expected 4, got 5"
    with pytest.raises(E.VectorError) as e:
        engine.search_similar([1.0] * 5, 3)
    assert e.value.kind == "DimensionMismatch"

    engine = E.VectorEngine(E.VectorEngineConfig(search_timeout=0.0))  # expires immediately
    engine.store_embedding("a", [1.0, 0.0])
    with pytest.raises(E.VectorError) as e:
        engine.search_similar([1.0, 0.0], 1)
    assert e.value.kind == "SearchTimeout" and str(e.value) == "search timeout: search_similar exceeded 0ms"
    with pytest.raises(E.VectorError) as e:
        engine.search_similar_with_metric([1.0, 0.0], 1, E.DistanceMetric.Euclidean)
    assert "search_similar_with_metric" in str(e.value)
    engine = E.VectorEngine(E.VectorEngineConfig(search_timeout=30.0))
    engine.store_embedding("a", [1.0, 0.0])
    assert engine.search_similar([1.0, 0.0], 1)[0].key == "a"
    with pytest.raises(E.VectorError) as e:
        E.VectorEngine(E.VectorEngineConfig(sparse_threshold=1.5))  # lib.rs:5174
    assert e.value.kind == "ConfigurationError"


# ---- mirror cache protocol (the hnsw_cache lifecycle, lib.rs:9686-9944) ------------------------------
def test_mirror_built_lazily_and_patched_on_writes(E):
    """The reference invalidates its cache on every store/delete (cache_invalidated_on_store / _on_delete);
    the flat GPU mirror is patched in place instead — what must hold is what those tests
    protect: a search after a write sees exactly the store's current contents."""
    engine = E.VectorEngine()
    for i in range(50):
        engine.store_embedding(f"v{i}", [float(i + 1), 1.0, 0.5])
    assert not engine.mirror_cached() and engine.mirror_builds() == 0
    engine.search_similar([1.0, 1.0, 1.0], 3)
    assert engine.mirror_cached() and engine.mirror_builds() == 1
    engine.search_similar([2.0, 1.0, 1.0], 3)
    engine.search_similar_with_metric([2.0, 1.0, 1.0], 3, E.DistanceMetric.Euclidean)
    assert engine.mirror_builds() == 1                       # reused across searches and metrics
    engine.store_embedding("new", [100.0, 1.0, 0.5])         # appended into spare capacity
    assert engine.search_similar([100.0, 1.0, 0.5], 1)[0].key == "new"
    engine.store_embedding("v3", [-5.0, 2.0, 9.0])           # overwritten in place
    r = engine.search_similar([-5.0, 2.0, 9.0], 1)[0]
    assert r.key == "v3" and abs(r.score - 1.0) < 1e-6
    engine.delete_embedding("new")                           # tombstoned
    assert all(x.key != "new" for x in engine.search_similar([100.0, 1.0, 0.5], 51))
    assert len(engine.search_similar([100.0, 1.0, 0.5], 100)) == 50
    assert engine.mirror_builds() == 1                       # none of the above rebuilt the mirror
    engine.store_in_collection("other", "x", [1.0, 2.0, 3.0])  # another collection has its own mirror
    assert engine.mirror_cached() and not engine.mirror_cached("other")


def test_incremental_mirror_matches_oracle_under_churn(E):
    """Random interleaving of inserts, overwrites (same and different dimension), deletes and searches;
    after every batch the engine must equal the oracle run over its current contents."""
    rng = np.random.default_rng(99)
    d = 48
    engine = E.VectorEngine()
    truth = {}
    for i in range(1500):
        v = rng.standard_normal(d).astype(F)
        engine.store_embedding(f"k{i}", v)
        truth[f"k{i}"] = v
    nxt = 1500
    for batch in range(12):
        for _ in range(120):
            op = rng.integers(0, 4)
            if op == 0 or not truth:
                v = rng.standard_normal(d).astype(F)
                engine.store_embedding(f"k{nxt}", v)
                truth[f"k{nxt}"] = v
                nxt += 1
            elif op == 1:
                key = list(truth)[rng.integers(0, len(truth))]
                v = rng.standard_normal(d).astype(F)
                engine.store_embedding(key, v)
                truth[key] = v
            elif op == 2:
                key = list(truth)[rng.integers(0, len(truth))]
                engine.delete_embedding(key)
                del truth[key]
            else:  # overwrite with another dimension: leaves this dimension's mirror
                key = list(truth)[rng.integers(0, len(truth))]
                engine.store_embedding(key, rng.standard_normal(d + 8).astype(F))
                del truth[key]
        keys = sorted(truth)                      # oracle over the current d-dimensional contents
        A = np.stack([truth[k] for k in keys])
        q = rng.standard_normal(d).astype(F)
        for metric in E.DistanceMetric:
            res = engine.search_similar_with_metric(q, 25, metric)
            s = oc.scores_all(A, q, int(metric))
            got = {r.key: np.float32(r.score) for r in res}
            assert len(res) == 25 and set(got) <= set(keys)
            exp_scores = np.sort(s)[::-1][:25]
            assert np.array_equal(np.array([r.score for r in res], F), exp_scores)
            for r in res:                          # each returned score is that key's exact score
                assert np.float32(r.score) == s[keys.index(r.key)]
    assert engine.mirror_builds() <= 4            # churn is absorbed by patching, not by rebuilding every time
    # mass delete: more than a quarter dead -> one rebuild, still exact
    for key in list(truth)[: len(truth) // 2]:
        engine.delete_embedding(key)
        del truth[key]
    keys = sorted(truth)
    A = np.stack([truth[k] for k in keys])
    q = rng.standard_normal(d).astype(F)
    res = engine.search_similar(q, 10)
    assert np.array_equal(np.array([r.score for r in res], F), np.sort(oc.scores_all(A, q, 0))[::-1][:10])


# ---- collections (lib.rs:7723-7960) ---------------------------------------------------------------------
def test_collections(E):
    engine = E.VectorEngine()
    engine.create_collection("test", E.VectorCollectionConfig())
    assert engine.collection_exists("test")
    with pytest.raises(E.VectorError) as e:
        engine.create_collection("test", E.VectorCollectionConfig())
    assert e.value.kind == "CollectionExists" and str(e.value) == "Collection already exists: test"
    with pytest.raises(E.VectorError) as e:
        engine.delete_collection("nope")
    assert e.value.kind == "CollectionNotFound"
    engine.store_in_collection("test", "k", [1.0, 2.0])
    assert engine.collection_count("test") == 1
    engine.delete_collection("test")
    assert not engine.collection_exists("test") and engine.collection_count("test") == 0

    engine.store_in_collection("auto", "k", [1.0, 2.0])      # store_in_collection_without_prior_create
    assert np.array_equal(engine.get_from_collection("auto", "k"), np.array([1, 2], F))
    with pytest.raises(E.VectorError) as e:
        engine.get_from_collection("auto", "missing")
    assert str(e.value) == "Embedding not found: auto:missing"

    engine.create_collection("fixed", E.VectorCollectionConfig().with_dimension(3))
    engine.store_in_collection("fixed", "ok", [1.0, 2.0, 3.0])
    with pytest.raises(E.VectorError) as e:
        engine.store_in_collection("fixed", "bad", [1.0, 2.0])
    assert str(e.value) == "Dimension mismatch: expected 3, got 2"
    with pytest.raises(E.VectorError) as e:
        engine.search_in_collection("fixed", [1.0, 2.0], 5)   # search_in_collection_dimension_constraint
    assert str(e.value) == "Dimension mismatch: expected 3, got 2"


def test_search_in_collection(E):  # lib.rs:7900-7930
    engine = E.VectorEngine()
    engine.store_in_collection("products", "p1", [1.0, 0.0, 0.0])
    engine.store_in_collection("products", "p2", [0.0, 1.0, 0.0])
    engine.store_in_collection("products", "p3", [0.0, 0.0, 1.0])
    engine.store_embedding("p1", [0.0, 0.0, 1.0])             # default collection is a separate key space
    r = engine.search_in_collection("products", [1.0, 0.0, 0.0], 2)
    assert len(r) == 2 and r[0].key == "p1" and abs(r[0].score - 1.0) < 1e-6
    assert engine.search_in_collection("empty", [1.0, 2.0], 5) == []
    # per-collection metric (lib.rs:1614-1616)
    engine.create_collection("l2", E.VectorCollectionConfig().with_metric(E.DistanceMetric.Euclidean))
    engine.store_in_collection("l2", "near", [1.0, 0.0])
    engine.store_in_collection("l2", "far", [10.0, 0.0])
    r = engine.search_in_collection("l2", [0.0, 0.0], 2)      # zero query is fine for a non-cosine collection
    assert [x.key for x in r] == ["near", "far"] and abs(r[0].score - 0.5) < 1e-6
    engine.create_collection("dot", E.VectorCollectionConfig().with_metric(E.DistanceMetric.DotProduct))
    engine.store_in_collection("dot", "small", [1.0, 0.0])
    engine.store_in_collection("dot", "big", [3.0, 0.0])
    assert engine.search_in_collection("dot", [1.0, 0.0], 1)[0].key == "big"


# ---- filtered search (lib.rs:6968-7722) ----------------------------------------------------------------
def setup_filtered_search_engine(E):  # lib.rs:6968-7001
    engine = E.VectorEngine()
    for i, (cat, price) in enumerate(zip(["electronics", "clothing", "food"], [100, 50, 25])):
        engine.store_embedding_with_metadata(f"item{i}", [float(i + 1), 1.0, 1.0],
                                             {"category": cat, "price": price, "active": i % 2 == 0})
    return engine


def test_search_filtered_operators(E):  # lib.rs:7004-7135, 7277-7335
    FC = E.FilterCondition
    engine = setup_filtered_search_engine(E)
    q = [1.0, 1.0, 1.0]
    r = engine.search_similar_filtered(q, 10, FC.Eq("category", "electronics"))
    assert len(r) == 1 and r[0].key == "item0"
    r = engine.search_similar_filtered([1.0, 0.0, 0.0], 10, FC.Eq("price", 50))
    assert len(r) == 1 and r[0].key == "item1"
    assert len(engine.search_similar_filtered(q, 10, FC.Gt("price", 30))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Lt("price", 60))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Le("price", 50))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Ge("price", 50))) == 2
    r = engine.search_similar_filtered(q, 10, FC.Gt("price", 30).and_(FC.Lt("price", 80)))
    assert len(r) == 1 and r[0].key == "item1"
    assert len(engine.search_similar_filtered(q, 10, FC.Eq("category", "electronics").or_(FC.Eq("category", "food")))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.TRUE)) == 3
    assert len(engine.search_similar_filtered(q, 10, FC.In("category", ["electronics", "food"]))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Ne("category", "electronics"))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Eq("active", True))) == 2
    assert len(engine.search_similar_filtered(q, 10, FC.Eq("active", False))) == 1
    assert engine.search_similar_filtered(q, 10, FC.Eq("category", "nonexistent")) == []
    assert engine.search_similar_filtered(q, 10, FC.Eq("missing_field", 1)) == []
    assert len(engine.search_similar_filtered(q, 10, FC.Exists("price"))) == 3
    assert len(engine.search_similar_filtered(q, 10, FC.Contains("category", "oth"))) == 1
    assert len(engine.search_similar_filtered(q, 10, FC.StartsWith("category", "f"))) == 1
    assert engine.search_similar_filtered(q, 10, FC.Contains("price", "5")) == []      # non-string field
    assert len(engine.search_similar_filtered(q, 10, FC.Ge("price", 50.0))) == 2         # int vs float filter
    assert engine.search_similar_filtered(q, 10, FC.Eq("price", "50")) == []             # incompatible types
    assert engine.count_matching(FC.Gt("price", 30)) == 2


def test_search_filtered_strategies_and_errors(E):  # lib.rs:7337-7400, 7701-7722
    FC, FS = E.FilterCondition, E.FilteredSearchConfig
    engine = setup_filtered_search_engine(E)
    q = [1.0, 1.0, 1.0]
    f = FC.Eq("category", "electronics")
    assert len(engine.search_similar_filtered(q, 10, f, FS.pre_filter())) == 1
    assert len(engine.search_similar_filtered(q, 10, f, FS.post_filter())) == 1
    with pytest.raises(E.VectorError) as e:
        engine.search_similar_filtered([], 10, f)
    assert e.value.kind == "EmptyVector"
    with pytest.raises(E.VectorError) as e:
        engine.search_similar_filtered(q, 0, f)
    assert e.value.kind == "InvalidTopK"
    assert engine.search_similar_filtered([0.0, 0.0, 0.0], 10, f, FS.pre_filter()) == []
    # respects top_k
    engine = E.VectorEngine()
    for i in range(20):
        engine.store_embedding_with_metadata(f"i{i}", [float(i + 1), 1.0], {"g": "a"})
    assert len(engine.search_similar_filtered([1.0, 1.0], 5, FC.Eq("g", "a"))) == 5


def test_prefilter_is_exact_against_oracle(E):
    """search_with_pre_filter semantics (lib.rs:3514-3557) on a corpus where the filter is selective."""
    rng = np.random.default_rng(33)
    n, d, k = 5000, 64, 25
    A = rng.standard_normal((n, d)).astype(F)
    bucket = rng.integers(0, 10, n)
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_embedding_with_metadata(f"k{i}", A[i], {"bucket": int(bucket[i]), "name": f"n{i % 7}"})
    q = rng.standard_normal(d).astype(F)
    FC = E.FilterCondition
    for cond, keep in ((FC.Eq("bucket", 3), bucket == 3),
                       (FC.Lt("bucket", 2).or_(FC.Eq("name", "n0")), (bucket < 2) | (np.arange(n) % 7 == 0))):
        res = engine.search_similar_filtered(q, k, cond, E.FilteredSearchConfig.pre_filter())
        er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(keep))
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)
    # post-filter: oversample x3 then filter (approximate by design, lib.rs:3560-3579)
    res = engine.search_similar_filtered(q, k, FC.Eq("bucket", 3), E.FilteredSearchConfig.post_filter())
    er, es = oc.search(A, q, 3 * k, 0)
    exp = [f"k{i}" for i in er if bucket[int(i)] == 3][:k]
    assert [r.key for r in res] == exp


def test_search_filtered_in_collection(E):  # lib.rs:7935-8000
    engine = E.VectorEngine()
    engine.store_in_collection_with_metadata("test", "item1", [1.0, 0.0], {"category": "A"})
    engine.store_in_collection_with_metadata("test", "item2", [0.9, 0.1], {"category": "B"})
    engine.store_in_collection_with_metadata("test", "item3", [0.8, 0.2], {"category": "A"})
    FC = E.FilterCondition
    for cfg in (None, E.FilteredSearchConfig.pre_filter(), E.FilteredSearchConfig.post_filter()):
        r = engine.search_filtered_in_collection("test", [1.0, 0.0], 10, FC.Eq("category", "A"), cfg)
        assert [x.key for x in r] == ["item1", "item3"]
    assert engine.search_filtered_in_collection("test", [0.0, 0.0], 10, FC.TRUE) == []


def test_top_k_beyond_4096_returns_the_full_ranking(E):
    """The reference sorts every score and truncates (lib.rs:2026-2034): any top_k is legal."""
    rng = np.random.default_rng(8)
    n, d = 6000, 16
    A = rng.standard_normal((n, d)).astype(F)
    engine = E.VectorEngine()
    engine.batch_store_embeddings([f"k{i}" for i in range(n)], A)
    q = rng.standard_normal(d).astype(F)
    for top_k in (5000, n, 10 * n):
        res = engine.search_similar(q, top_k)
        er, es = oc.search(A, q, top_k, 0)
        assert len(res) == min(top_k, n)
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)
    res = engine.search_similar_with_metric(q, n, E.DistanceMetric.Euclidean)
    er, es = oc.search(A, q, n, 1)
    assert [r.key for r in res] == [f"k{i}" for i in er]
    # post-filter oversampling asks the scan for 3 * top_k (lib.rs:3567-3568)
    eng2 = E.VectorEngine()
    for i in range(n):
        eng2.store_embedding_with_metadata(f"k{i}", A[i], {"even": i % 2 == 0})
    res = eng2.search_similar_filtered(q, 1500, E.FilterCondition.Eq("even", True), E.FilteredSearchConfig.post_filter())
    er, _ = oc.search(A, q, 4500, 0)
    assert [r.key for r in res] == [f"k{i}" for i in er if i % 2 == 0][:1500]


# ---- metadata CRUD + pagination (lib.rs:5367-5396, 6693-6855) -----------------------------------------------
def test_metadata_crud(E):
    engine = E.VectorEngine()
    engine.store_embedding_with_metadata("item", [1.0, 2.0], {"color": "red"})
    engine.update_metadata("item", {"size": "large", "color": "blue"})            # update_metadata_basic
    md = engine.get_metadata("item")
    assert md == {"color": "blue", "size": "large"}
    assert np.array_equal(engine.get_embedding("item"), np.array([1.0, 2.0], F))  # the vector is untouched
    for call in (lambda: engine.get_metadata("nope"), lambda: engine.update_metadata("nope", {"a": 1}),
                 lambda: engine.remove_metadata_field("nope", "a"), lambda: engine.get_metadata_field("nope", "a")):
        with pytest.raises(E.VectorError) as e:
            call()
        assert e.value.kind == "NotFound"
    engine.remove_metadata_field("item", "color")                                  # remove_metadata_field_basic
    assert not engine.has_metadata_field("item", "color") and engine.has_metadata_field("item", "size")
    engine.remove_metadata_field("item", "never_there")                            # not an error
    assert not engine.has_metadata_field("nope", "size")
    assert engine.get_metadata_field("item", "size") == "large"
    assert engine.get_metadata_field("item", "missing") is None                    # Ok(None)
    engine.update_metadata("item", {"n": 3, "f": 0.5, "b": True, "z": None})
    assert engine.get_metadata("item") == {"size": "large", "n": 3, "f": 0.5, "b": True, "z": None}
    engine.store_embedding("plain", [1.0, 0.0])
    assert engine.get_metadata("plain") == {}


def test_metadata_updates_reach_the_device_columns(E):
    """update_metadata / remove_metadata_field patch single cells of the HBM columns; the next filtered search sees them."""
    FC, PRE = E.FilterCondition, E.FilteredSearchConfig.pre_filter()
    rng = np.random.default_rng(4)
    n, d = 300, 8
    A = rng.standard_normal((n, d)).astype(F)
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_embedding_with_metadata(f"k{i}", A[i], {"g": i % 3})
    q = rng.standard_normal(d).astype(F)
    assert len(engine.search_similar_filtered(q, n, FC.Eq("g", 7), PRE)) == 0     # builds mirror + columns
    engine.update_metadata("k5", {"g": 7, "tag": "x"})                             # changed cell + a brand-new column
    engine.update_metadata("k9", {"g": 7.0})                                       # the cell changes type: Float 7.0 == Int 7
    engine.remove_metadata_field("k12", "g")
    assert sorted(r.key for r in engine.search_similar_filtered(q, n, FC.Eq("g", 7), PRE)) == ["k5", "k9"]
    assert [r.key for r in engine.search_similar_filtered(q, n, FC.Exists("tag"), PRE)] == ["k5"]
    got = {r.key for r in engine.search_similar_filtered(q, n, FC.Exists("g"), PRE)}
    assert got == {f"k{i}" for i in range(n)} - {"k12"}
    assert engine.column_builds() == 1 and engine.mirror_builds() == 1
    assert engine.list_keys_matching(FC.Eq("g", 7)) == ["k5", "k9"] and engine.count_matching(FC.Eq("g", 7)) == 2
    assert engine.estimate_filter_selectivity(FC.Exists("g")) == pytest.approx(0.99)   # first 100 keys, k12 among them
    assert engine.batch_delete_embeddings(["k5", "nope", "k6"]) == 2               # lib.rs:2924-2940
    assert [r.key for r in engine.search_similar_filtered(q, n, FC.Eq("g", 7), PRE)] == ["k9"]
    assert engine.dimension() == d and E.VectorEngine().dimension() is None


def test_paginated_searches(E):
    engine = E.VectorEngine()
    for i in range(10):
        engine.store_embedding(f"v{i}", [float(i), 1.0])
    P = E.Pagination
    res = engine.search_similar_paginated([5.0, 1.0], 10, P(0, 3).with_total())   # search_similar_paginated_basic
    full = engine.search_similar([5.0, 1.0], 10)
    assert [r.key for r in res.items] == [r.key for r in full[:3]]
    assert res.total_count == 3 and not res.has_more                               # min(skip + limit, top_k) results were searched
    res = engine.search_similar_paginated([5.0, 1.0], 10, P(2, 3).with_total())
    assert [r.key for r in res.items] == [r.key for r in full[2:5]] and res.total_count == 5 and not res.has_more
    res = engine.search_similar_paginated([5.0, 1.0], 4, P(1, None))               # no limit: skip + top_k, capped at top_k
    assert [r.key for r in res.items] == [r.key for r in full[1:4]] and res.total_count is None and not res.has_more
    res = engine.search_similar_paginated([5.0, 1.0], 10, P(20, 5).with_total())   # skip past the end
    assert res.items == [] and res.total_count == 10 and not res.has_more
    for i in range(5):
        engine.set_entity_embedding(f"user:{i}", [float(i), 1.0])
    res = engine.search_entities_paginated([2.0, 1.0], 5, P(0, 2).with_total())    # search_entities_paginated_basic
    assert len(res.items) == 2 and res.items[0].key == "user:2"
    engine.store_in_collection("c", "a", [1.0])
    assert engine.exists_in_collection("c", "a") and not engine.exists_in_collection("c", "b")
    assert engine.list_collection_keys("c") == ["a"] and engine.list_collection_keys("nope") == []


# ---- unified entity mode (lib.rs:4692-4867) -----------------------------------------------------------------
def test_entity_embedding_crud(E):  # lib.rs:4692-4770, 4815-4840
    engine = E.VectorEngine()
    assert not engine.entity_has_embedding("user:1")
    assert engine.count_entities_with_embeddings() == 0
    engine.set_entity_embedding("user:1", [1.0, 2.0, 3.0])
    assert np.array_equal(engine.get_entity_embedding("user:1"), np.array([1.0, 2.0, 3.0], F))
    assert engine.entity_has_embedding("user:1")
    engine.set_entity_embedding("user:2", [3.0, 4.0])
    assert sorted(engine.scan_entities_with_embeddings()) == ["user:1", "user:2"]
    assert engine.count_entities_with_embeddings() == 2
    engine.remove_entity_embedding("user:1")
    assert not engine.entity_has_embedding("user:1")
    for call in (lambda: engine.remove_entity_embedding("user:999"), lambda: engine.get_entity_embedding("user:999")):
        with pytest.raises(E.VectorError) as e:
            call()
        assert e.value.kind == "NotFound"
    with pytest.raises(E.VectorError) as e:
        engine.set_entity_embedding("user:1", [])
    assert e.value.kind == "EmptyVector"
    # entity keys and embedding keys are separate key spaces (`user:1` vs `emb:user:1`)
    engine.store_embedding("user:2", [9.0, 9.0])
    assert np.array_equal(engine.get_entity_embedding("user:2"), np.array([3.0, 4.0], F))
    assert engine.count() == 1 and engine.count_entities_with_embeddings() == 1


def test_search_entities(E):  # lib.rs:4772-4790, 4842-4864, 6372-6390
    engine = E.VectorEngine()
    engine.set_entity_embedding("user:1", [1.0, 0.0, 0.0])
    engine.set_entity_embedding("user:2", [0.0, 1.0, 0.0])
    engine.set_entity_embedding("user:3", [1.0, 1.0, 0.0])
    engine.store_embedding("doc", [1.0, 0.0, 0.0])          # not an entity: never returned
    engine.set_entity_embedding("other_dim", [1.0, 0.0])    # dimension mismatch: skipped
    r = engine.search_entities([1.0, 0.0, 0.0], 3)
    assert len(r) == 3 and r[0].key == "user:1" and abs(r[0].score - 1.0) < 1e-6
    assert [x.key for x in r] == ["user:1", "user:3", "user:2"]
    with pytest.raises(E.VectorError) as e:
        engine.search_entities([], 5)
    assert e.value.kind == "EmptyVector"
    with pytest.raises(E.VectorError) as e:
        engine.search_entities([1.0], 0)
    assert e.value.kind == "InvalidTopK"
    assert engine.search_entities([0.0, 0.0, 0.0], 5) == []
    limited = E.VectorEngine(E.VectorEngineConfig(max_dimension=2))
    with pytest.raises(E.VectorError) as e:
        limited.search_entities([1.0, 0.0, 0.0], 1)
    assert e.value.kind == "DimensionMismatch"
    with pytest.raises(E.VectorError) as e:
        limited.set_entity_embedding("u", [1.0, 0.0, 0.0])
    assert e.value.kind == "DimensionMismatch"


def test_search_entities_matches_oracle(E):
    rng = np.random.default_rng(61)
    n, d, k = 4000, 96, 30
    A = rng.standard_normal((n, d)).astype(F)
    engine = E.VectorEngine()
    for i in range(n):
        engine.set_entity_embedding(f"k{i}", A[i])
    live = np.ones(n, bool)
    for i in rng.choice(n, 300, replace=False):
        engine.remove_entity_embedding(f"k{int(i)}")
        live[int(i)] = False
    for i in rng.choice(np.flatnonzero(live), 200, replace=False):
        A[int(i)] = rng.standard_normal(d).astype(F)
        engine.set_entity_embedding(f"k{int(i)}", A[int(i)])
    for t in range(3):
        q = rng.standard_normal(d).astype(F)
        res = engine.search_entities(q, k)
        er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(live))
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)


def test_search_entities_is_bounded_by_max_keys_per_scan(E):
    """`let keys = self.store.scan("").into_iter().take(max_scan)` (lib.rs:3179-3180): only the first max_keys_per_scan keys of
    the scan are scored.  Which keys those are is the scan order's business (a HashSet in the reference, slot order here);
    what is checkable is that the answer is exactly the oracle's top-k over SOME max_scan keys — here the bounded
    scan_entities_with_embeddings() — and that an unbounded engine sees everything."""
    rng = np.random.default_rng(67)
    n, d, k, bound = 900, 64, 20, 300
    A = rng.standard_normal((n, d)).astype(F)
    bounded = E.VectorEngine(E.VectorEngineConfig(max_keys_per_scan=bound))
    free = E.VectorEngine()
    for i in range(n):
        bounded.set_entity_embedding(f"k{i}", A[i])
        free.set_entity_embedding(f"k{i}", A[i])
    for i in (5, 17, 250):   # removed keys free their slots: the scan order no longer equals insertion order
        bounded.remove_entity_embedding(f"k{i}")
        free.remove_entity_embedding(f"k{i}")
    scanned = bounded.scan_entities_with_embeddings()
    assert len(scanned) == bound
    part = np.zeros(n, bool)
    part[[int(s[1:]) for s in scanned]] = True
    live = np.ones(n, bool)
    live[[5, 17, 250]] = False
    for t in range(3):
        q = rng.standard_normal(d).astype(F)
        res = bounded.search_entities(q, k)
        er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(part))
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)
        res = free.search_entities(q, k)
        er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(live))
        assert [r.key for r in res] == [f"k{i}" for i in er]
    # top_k larger than the bounded scan: every scanned key comes back, nothing else
    res = bounded.search_entities(A[0], n)
    assert len(res) == bound and {r.key for r in res} == set(scanned)


# ---- concurrency contract (lib.rs:5615-5711) -------------------------------------------------------------
def test_concurrent_search_and_store(E):
    engine = E.VectorEngine()
    for i in range(200):
        engine.store_embedding(f"v{i}", create_test_vector(16, i))
    errors = []

    def searcher(t):
        try:
            for j in range(5):
                r = engine.search_similar(create_test_vector(16, (t * 7 + j) % 200), 5)
                assert len(r) == 5
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def writer(t):
        try:
            for j in range(5):
                engine.store_embedding(f"w{t}_{j}", create_test_vector(16, 1000 + t * 10 + j))
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=searcher, args=(t,)) for t in range(20)]
    threads += [threading.Thread(target=writer, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert engine.count() == 220


def test_native_threads_searching_while_writers_write():
    """C++ threads on the engine ABI (no GIL): 6 searchers overlap on the GPU through the shard's host slots while 2
    writers overwrite, append and delete — no error, no crash, every search returns a full result list."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "micro", "engine_mt")
    lib = os.path.join(root, "neumann_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tools", "micro", "engine_mt.cpp"), "-L", lib, "-lneumann_gpu", "-lpthread",
                           "-Wl,-rpath," + lib])
    import torch  # the harness must resolve libamdhip64.so.7 the way the Python processes do (torch's bundled runtime)
    env = dict(os.environ, ENGINE_MT_WRITERS="2",
               LD_LIBRARY_PATH=os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, "30000", "64", "50", "300", "1", "6"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("rows=")]
    assert len(lines) == 2 and not any("ERRORS" in l for l in lines), out.stdout
    assert all("writers=2" in l for l in lines)


def test_interleaved_stores_deletes_and_searches_on_the_8_bit_mirror(E):
    """The incremental mirror protocol (lib.rs:1840-1868, 1915-1925) on a collection large enough for the pipeline path and
    the 8-bit mirror (80 000 x 256: beyond the single-launch search, row stride a multiple of 128): overwrites re-quantize
    rows in place, appends extend the mirror at the next search, deletes tombstone — interleaved with searches of every
    metric, each compared with the oracle over the current contents; then the same from threads (lib.rs:5615-5669's
    contract: searches and stores from many threads, no error, full lists), and a last comparison with the oracle."""
    rng = np.random.default_rng(4242)
    n0, d = 80_000, 256
    A0 = rng.standard_normal((n0, d)).astype(np.float32)
    eng = E.VectorEngine()
    eng.batch_store_embeddings([f"k{i}" for i in range(n0)], A0)
    model = {f"k{i}": A0[i] for i in range(n0)}
    next_key = n0

    def check(metric=None):
        keys = sorted(model, key=lambda s: int(s[1:]))
        A = np.stack([model[k_] for k_ in keys])
        q = rng.standard_normal(d).astype(np.float32)
        k = int(rng.choice([1, 10, 100]))
        metric = metric if metric is not None else (E.DistanceMetric.Cosine, E.DistanceMetric.Euclidean, E.DistanceMetric.DotProduct)[int(rng.integers(0, 3))]
        res = eng.search_similar_with_metric(q, k, metric)
        er, es = oc.search(A, q, k, int(metric), nthreads=8, partial=True, native=True)
        assert [np.float32(r.score) for r in res] == list(es)
        assert [r.key for r in res] == [keys[int(i)] for i in er]   # (random rows: no ties)

    check(E.DistanceMetric.Cosine)
    hb = eng.mirror_hbm_bytes(d)
    assert hb is not None and hb[1] == hb[0] // 4 + 12 * (hb[2] // 8), ("the mirror of this collection is the 8-bit one (1 B per element)", hb)
    for step in range(120):
        op = rng.random()
        if op < 0.35:                              # append
            key = f"k{next_key}"
            next_key += 1
            model[key] = rng.standard_normal(d).astype(np.float32)
            eng.store_embedding(key, model[key])
        elif op < 0.6:                             # overwrite
            key = f"k{int(rng.integers(0, n0))}"
            if key in model:
                model[key] = rng.standard_normal(d).astype(np.float32)
                eng.store_embedding(key, model[key])
        elif op < 0.75:                            # delete
            key = f"k{int(rng.integers(0, n0))}"
            if key in model:
                eng.delete_embedding(key)
                del model[key]
        else:
            check()
    check()
    # threads: 6 searchers, 2 writers of NEW keys (what they wrote is known afterwards)
    errors = []
    written = [{} for _ in range(2)]

    def searcher(t):
        try:
            r = np.random.default_rng(100 + t)
            for j in range(40):
                res = eng.search_similar(r.standard_normal(d).astype(np.float32), 10)
                assert len(res) == 10
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def writer(t):
        try:
            r = np.random.default_rng(200 + t)
            for j in range(60):
                key = f"k{1_000_000 * (t + 1) + j}"
                v = r.standard_normal(d).astype(np.float32)
                eng.store_embedding(key, v)
                written[t][key] = v
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=searcher, args=(t,)) for t in range(6)] + [threading.Thread(target=writer, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for w in written:
        model.update(w)
    assert eng.count() == len(model)
    for metric in (E.DistanceMetric.Cosine, E.DistanceMetric.Euclidean, E.DistanceMetric.DotProduct):
        check(metric)
    hb = eng.mirror_hbm_bytes(d)
    # the searcher threads' calls were merged into batches of 2, 3, 4 ... queries (the coalescer): on 256-element rows those run
    # as pairs over the 8-bit mirror — nobody built a bf16 mirror beside it
    assert hb is not None and hb[1] == hb[0] // 4 + 12 * (hb[2] // 8), ("still ONE mirror, the 8-bit one, after the soak", hb)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("NMN_FUZZ_SEEDS", "6"))))
def test_random_interleaving_of_stores_deletes_and_searches(E, seed):
    """Stores are only recorded next to a live mirror and uploaded in batches before the next search: any order of
    new keys, overwrites, deletes, re-inserts and searches (three metrics, random k) must answer like the oracle over
    the current contents."""
    rng = np.random.default_rng(400 + seed)
    d = int(rng.choice([8, 48, 100, 128]))
    model = {}                                     # key -> vector
    tags = {}                                      # key -> metadata value "b" (every store writes one)
    eng = E.VectorEngine()
    next_key = 0

    def check():
        keys = sorted(model, key=lambda s: int(s[1:]))
        if not keys:
            return
        A = np.stack([model[k_] for k_ in keys])
        q = rng.standard_normal(d).astype(np.float32)
        k = int(rng.choice([1, 3, 10, 50]))
        metric = (E.DistanceMetric.Cosine, E.DistanceMetric.Euclidean, E.DistanceMetric.DotProduct)[int(rng.integers(0, 3))]
        if rng.random() < 0.4:                     # pre-filtered (always cosine, lib.rs:3514-3557)
            b = int(rng.integers(0, 4))
            keep = np.array([tags[k_] == b for k_ in keys], bool)
            res = eng.search_similar_filtered(q, k, E.FilterCondition.Eq("b", b), E.FilteredSearchConfig.pre_filter())
            if not keep.any():
                assert res == []
                return
            er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(keep))
        else:
            res = eng.search_similar_with_metric(q, k, metric)
            er, es = oc.search(A, q, k, int(metric))
        got = {r.key: np.float32(r.score) for r in res}
        # ties: the reference's order among equal scores is unspecified (HashSet scan order); compare as score lists and
        # as sets above the last score
        assert [np.float32(r.score) for r in res] == list(es)
        last = es[-1] if len(es) else None
        for i, s_ in zip(er, es):
            if s_ != last:
                assert keys[int(i)] in got and got[keys[int(i)]] == s_

    for step in range(260):
        op = rng.random()
        if op < 0.45 or not model:                 # new key
            key = f"k{next_key}"
            next_key += 1
            v = rng.standard_normal(d).astype(np.float32)
            tags[key] = int(rng.integers(0, 4))
            eng.store_embedding_with_metadata(key, v, {"b": tags[key]})
            model[key] = v
        elif op < 0.65:                            # overwrite
            key = list(model)[int(rng.integers(0, len(model)))]
            v = rng.standard_normal(d).astype(np.float32)
            tags[key] = int(rng.integers(0, 4))
            eng.store_embedding_with_metadata(key, v, {"b": tags[key]})
            model[key] = v
        elif op < 0.8:                             # delete
            key = list(model)[int(rng.integers(0, len(model)))]
            eng.delete_embedding(key)
            del model[key]
            del tags[key]
        else:
            check()
    check()
    assert eng.count() == len(model)


# ---- router-level cases at the engine boundary (integration_tests/tests/distance_metrics.rs:37-140, edge_cases.rs:80-127) -------------
@pytest.mark.parametrize("name,rows,query,metric,first,second", [
    ("default_metric", {"vec:1": [1.0, 0.0, 0.0, 0.0], "vec:2": [0.9, 0.1, 0.0, 0.0], "vec:3": [0.0, 1.0, 0.0, 0.0]}, "vec:1", None, "vec:1", "vec:2"),
    ("cosine", {"cos:1": [1.0, 0.0, 0.0, 0.0], "cos:2": [0.707, 0.707, 0.0, 0.0], "cos:3": [0.0, 1.0, 0.0, 0.0]}, "cos:1", "Cosine", "cos:1", None),
    ("euclidean", {"euc:1": [0.5, 0.5, 0.5, 0.5], "euc:2": [0.6, 0.5, 0.5, 0.5], "euc:3": [1.0, 1.0, 1.0, 1.0]}, "euc:1", "Euclidean", "euc:1", "euc:2"),
    ("dot_product", {"dot:1": [1.0, 0.0, 0.0, 0.0], "dot:2": [0.5, 0.5, 0.0, 0.0], "dot:3": [0.0, 0.0, 1.0, 0.0]}, "dot:1", "DotProduct", "dot:1", "dot:2"),
    ("vector_with_cosine", {"target:1": [1.0, 0.0, 0.0, 0.0], "target:2": [0.8, 0.2, 0.0, 0.0], "target:3": [0.0, 0.0, 1.0, 0.0]},
     [1.0, 0.0, 0.0, 0.0], "Cosine", "target:1", "target:2"),
    # query_router/src/lib.rs:9436-9524
    ("router_cosine", {"cos_a": [1.0, 0.0], "cos_b": [0.0, 1.0], "cos_c": [0.707, 0.707]}, [1.0, 0.0], "Cosine", "cos_a", "cos_c"),
    ("router_euclidean", {"euc_a": [1.0, 0.0], "euc_b": [2.0, 0.0], "euc_c": [10.0, 0.0]}, [1.0, 0.0], "Euclidean", "euc_a", "euc_b"),
    ("router_euclidean_zero_query", {"zero_origin": [0.0, 0.0], "zero_unit": [1.0, 0.0], "zero_far": [10.0, 0.0]}, [0.0, 0.0], "Euclidean",
     "zero_origin", "zero_unit"),
    ("router_dot_product", {"dot_a": [1.0, 0.0], "dot_b": [2.0, 0.0], "dot_c": [0.5, 0.0]}, [1.0, 0.0], "DotProduct", "dot_b", "dot_a"),
])
def test_router_level_similar_cases(E, name, rows, query, metric, first, second):  # distance_metrics.rs:37-140
    engine = E.VectorEngine()
    for k, v in rows.items():
        engine.store_embedding(k, v)
    q = rows[query] if isinstance(query, str) else query
    res = engine.search_similar(q, 3) if metric is None else engine.search_similar_with_metric(q, 3, getattr(E.DistanceMetric, metric))
    assert len(res) == 3 and res[0].key == first
    if second is not None:
        assert res[1].key == second
    assert res[0].score >= res[1].score >= res[2].score
    # ... and the scores are the oracle's, bit for bit
    keys = list(rows)
    A = np.array([rows[k] for k in keys], F)
    er, es = oc.search(A, np.asarray(q, F), 3, {None: 0, "Cosine": 0, "Euclidean": 1, "DotProduct": 2}[metric])
    assert [r.key for r in res] == [keys[i] for i in er]
    assert np.array_equal(np.array([r.score for r in res], F).view(np.uint32), es.view(np.uint32))


def test_empty_store_then_single_embedding(E):  # edge_cases.rs:80-96
    engine = E.VectorEngine()
    q = (np.arange(32, dtype=F) * F(0.03125)).astype(F)
    assert engine.search_similar(q, 10) == []
    engine.store_embedding("single", q)
    res = engine.search_similar(q, 10)
    assert len(res) == 1 and res[0].key == "single"


def test_zero_vector_handling(E):  # edge_cases.rs:98-127
    engine = E.VectorEngine()
    z = np.zeros(32, F)
    engine.store_embedding("zero", z)
    assert np.array_equal(engine.get_embedding("zero"), z)
    assert engine.search_similar(z, 10) == []  # (a zero query: Ok([]), lib.rs:1966-1969)
    res = engine.search_similar_with_metric(z, 10, E.DistanceMetric.Euclidean)
    assert len(res) == 1 and res[0].key == "zero" and res[0].score == 1.0


def test_sparse_vectors_in_similarity_search(E):  # integration_tests/tests/sparse_vectors.rs:205-232, 412-438
    engine = E.VectorEngine()
    A = np.zeros((10, 64), F)
    for i in range(10):
        for j in range(10):
            A[i, (i * 5 + j) % 64] = np.sin(F((i + j)) * F(0.1), dtype=F)
        engine.store_embedding(f"sparse:{i}", A[i])
    q = np.zeros(64, F)
    for j in range(10):
        q[j % 64] = np.sin(F(j) * F(0.1), dtype=F)
    res = engine.search_similar(q, 5)
    assert len(res) == 5
    er, es = oc.search(A, q, 5, 0)
    assert [r.key for r in res] == [f"sparse:{i}" for i in er]
    assert np.array_equal(np.array([r.score for r in res], F).view(np.uint32), es.view(np.uint32))
    engine2 = E.VectorEngine()
    v = np.zeros((3, 100), F)
    v[0, 0] = 1.0
    v[1, 0] = v[1, 1] = 0.707
    v[2, 1] = 1.0
    for i in range(3):
        engine2.store_embedding(f"v{i + 1}", v[i])
    q2 = np.zeros(100, F)
    q2[0] = 1.0
    res = engine2.search_similar(q2, 3)
    assert [r.key for r in res] == ["v1", "v2", "v3"]


def test_search_filtered_typed_comparisons(E):  # lib.rs:7491-7700
    FC = E.FilterCondition

    def engine_with(items):
        e = E.VectorEngine()
        for key, vec, meta in items:
            if meta is None:
                e.store_embedding(key, vec)
            else:
                e.store_embedding_with_metadata(key, vec, meta)
        return e

    # search_filtered_float_comparison (7491-7519)
    e = engine_with([("high", [1.0, 0.0], {"score": 0.95}), ("low", [0.0, 1.0], {"score": 0.5})])
    r = e.search_similar_filtered([1.0, 0.0], 10, FC.Gt("score", 0.8))
    assert len(r) == 1 and r[0].key == "high"
    # search_filtered_mixed_int_float_comparison (7522-7541): a float field against an int filter value
    e = engine_with([("item", [1.0, 0.0], {"value": 50.5})])
    assert len(e.search_similar_filtered([1.0, 0.0], 10, FC.Gt("value", 50))) == 1
    # search_filtered_int_vs_float_filter (7544-7572): an int field against a float filter value, and the boundary 100 > 100.0
    e = engine_with([("item", [1.0, 0.0], {"count": 100})])
    assert len(e.search_similar_filtered([1.0, 0.0], 10, FC.Gt("count", 50.5))) == 1
    assert e.search_similar_filtered([1.0, 0.0], 10, FC.Gt("count", 100.0)) == []
    # search_filtered_null_comparison (7575-7601): an explicit null matches Eq(Null), a missing field does not
    e = engine_with([("with_null", [1.0, 0.0], {"optional": None}), ("without_field", [0.0, 1.0], None)])
    r = e.search_similar_filtered([1.0, 0.0], 10, FC.Eq("optional", None))
    assert len(r) == 1 and r[0].key == "with_null"
    # search_filtered_string_comparison (7604-7644): lexicographic order
    e = engine_with([("item1", [1.0, 0.0], {"name": "apple"}), ("item2", [0.0, 1.0], {"name": "banana"})])
    assert len(e.search_similar_filtered([1.0, 1.0], 10, FC.Gt("name", "app"))) == 2
    r = e.search_similar_filtered([1.0, 1.0], 10, FC.Le("name", "apple"))
    assert len(r) == 1 and r[0].key == "item1"
    # search_filtered_bool_false (7647-7676)
    e = engine_with([("active_item", [1.0, 0.0], {"active": True}), ("inactive_item", [0.0, 1.0], {"active": False})])
    r = e.search_similar_filtered([1.0, 1.0], 10, FC.Eq("active", False))
    assert len(r) == 1 and r[0].key == "inactive_item"
    # search_filtered_incompatible_types (7679-7698)
    e = engine_with([("item", [1.0, 0.0], {"value": "text"})])
    assert e.search_similar_filtered([1.0, 0.0], 10, FC.Eq("value", 42)) == []
    # search_filtered_contains / starts_with on their own stores (7155-7258)
    e = engine_with([("item1", [1.0, 0.0], {"description": "blue shirt"}), ("item2", [0.0, 1.0], {"description": "red pants"})])
    r = e.search_similar_filtered([1.0, 0.0], 10, FC.Contains("description", "shirt"))
    assert len(r) == 1 and r[0].key == "item1"
    e = engine_with([("item1", [1.0, 0.0], {"sku": "ABC123"}), ("item2", [0.0, 1.0], {"sku": "XYZ789"})])
    r = e.search_similar_filtered([1.0, 0.0], 10, FC.StartsWith("sku", "ABC"))
    assert len(r) == 1 and r[0].key == "item1"
    e = engine_with([("item", [1.0, 0.0], {"count": 123})])
    assert e.search_similar_filtered([1.0, 0.0], 10, FC.StartsWith("count", "1")) == []
    assert e.search_similar_filtered([1.0, 0.0], 10, FC.Contains("count", "2")) == []
    # search_filtered_missing_field (7261-7274)
    e = engine_with([("item", [1.0, 0.0], None)])
    assert e.search_similar_filtered([1.0, 0.0], 10, FC.Eq("missing", 42)) == []


def test_search_probe_times_native_calls_and_returns_the_same_answers(E):
    """nmn_engine_search_probe (bench.py's published_shapes leg: the reference's own bench shapes, vector_engine_bench.rs:40-77): N native
    back-to-back search_similar calls, one duration per call; the engine's answer for the same query is the oracle's."""
    rng = np.random.default_rng(7)
    n, d = 1000, 128
    A = rng.uniform(-1.0, 1.0, (n, d)).astype(F)
    Q = rng.uniform(-1.0, 1.0, (4, d)).astype(F)
    eng = E.VectorEngine()
    eng.batch_store_embeddings([f"v{i}" for i in range(n)], A)
    us = eng.search_probe(Q, 10, 50)
    assert us.shape == (50,) and np.all(us > 0) and np.all(np.isfinite(us))
    res = eng.search_similar(Q[1], 10)
    er, es = oc.search(A, Q[1], 10, 0)
    assert [r.key for r in res] == [f"v{int(i)}" for i in er]
    assert np.array_equal(np.array([r.score for r in res], dtype=F), es)
    eng.close()
