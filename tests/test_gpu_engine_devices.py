"""nmn_engine_config.devices[]: an engine whose collections are spread over several GPUs (one nmn_sharded index per
mirror) must answer exactly like the single-GPU engine and like the oracle.  The GPU box has one device, so the
shards are logical (the ordinal repeats): same code path — per-shard streams, gather, device merge — with peer copies
standing in for the RCCL all-gather (tests/test_gpu_sharded_handle.py covers the RCCL gather with one rank)."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture
def E():
    from neumann_amd import engine
    return engine


def engines(E, n_shards=3):
    one = E.VectorEngine()
    many = E.VectorEngine(E.VectorEngineConfig(devices=(0,) * n_shards))
    return one, many


def same(a, b):
    assert [r.key for r in a] == [r.key for r in b]
    assert np.array_equal(np.array([r.score for r in a], F), np.array([r.score for r in b], F))


def test_sharded_engine_equals_single_device_engine_and_oracle(E):
    rng = np.random.default_rng(2024)
    n, d = 7000, 96
    A = rng.standard_normal((n, d)).astype(F)
    one, many = engines(E)
    for eng in (one, many):
        eng.batch_store_embeddings([f"k{i}" for i in range(n)], A)
    for t in range(4):
        q = rng.standard_normal(d).astype(F)
        for metric in E.DistanceMetric:
            a = one.search_similar_with_metric(q, 40, metric)
            b = many.search_similar_with_metric(q, 40, metric)
            same(a, b)
            er, es = oc.search(A, q, 40, int(metric))
            assert [r.key for r in b] == [f"k{i}" for i in er]
            assert np.array_equal(np.array([r.score for r in b], F), es)
    assert many.mirror_builds() == 1


def test_sharded_engine_under_churn_deletes_and_appends(E):
    """appends land in the spare rows of the ranges in order, deletes travel as the live bitmap sliced per shard"""
    rng = np.random.default_rng(7)
    d = 40
    one, many = engines(E, 4)
    truth = {}
    for i in range(3000):
        v = rng.standard_normal(d).astype(F)
        truth[f"k{i}"] = v
        for eng in (one, many):
            eng.store_embedding(f"k{i}", v)
    nxt = 3000
    for batch in range(6):
        if batch == 0:
            same(one.search_similar(truth["k5"], 10), many.search_similar(truth["k5"], 10))  # builds the mirrors
        for _ in range(150):
            op = rng.integers(0, 3)
            if op == 0:
                v = rng.standard_normal(d).astype(F)
                truth[f"k{nxt}"] = v
                for eng in (one, many):
                    eng.store_embedding(f"k{nxt}", v)
                nxt += 1
            elif op == 1:
                key = list(truth)[rng.integers(0, len(truth))]
                v = rng.standard_normal(d).astype(F)
                truth[key] = v
                for eng in (one, many):
                    eng.store_embedding(key, v)
            else:
                key = list(truth)[rng.integers(0, len(truth))]
                del truth[key]
                for eng in (one, many):
                    eng.delete_embedding(key)
        keys = sorted(truth)
        M = np.stack([truth[k] for k in keys])
        q = rng.standard_normal(d).astype(F)
        for metric in E.DistanceMetric:
            b = many.search_similar_with_metric(q, 30, metric)
            same(one.search_similar_with_metric(q, 30, metric), b)
            s = oc.scores_all(M, q, int(metric))
            assert np.array_equal(np.array([r.score for r in b], F), np.sort(s)[::-1][:30])
            for r in b:
                assert np.float32(r.score) == s[keys.index(r.key)]
    assert many.mirror_builds() <= 3


def test_sharded_engine_filtered_search_and_large_k(E):
    rng = np.random.default_rng(33)
    n, d, k = 6000, 64, 25
    A = rng.standard_normal((n, d)).astype(F)
    bucket = rng.integers(0, 10, n)
    one, many = engines(E)
    for eng in (one, many):
        for i in range(n):
            eng.store_embedding_with_metadata(f"k{i}", A[i], {"bucket": int(bucket[i]), "name": f"n{i % 7}"})
    q = rng.standard_normal(d).astype(F)
    FC = E.FilterCondition
    for cond, keep in ((FC.Eq("bucket", 3), bucket == 3),
                       (FC.Lt("bucket", 2).or_(FC.Eq("name", "n0")), (bucket < 2) | (np.arange(n) % 7 == 0))):
        res = many.search_similar_filtered(q, k, cond, E.FilteredSearchConfig.pre_filter())
        same(one.search_similar_filtered(q, k, cond, E.FilteredSearchConfig.pre_filter()), res)
        er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(keep))
        assert [r.key for r in res] == [f"k{i}" for i in er]
        assert np.all(np.array([r.score for r in res], F) == es)
    assert many.device_filter_evals() >= 2   # the predicate ran on the GPU (devices[0]), not on the host
    # nothing selected -> empty (lib.rs:3532-3534)
    assert many.search_similar_filtered(q, k, FC.Eq("bucket", 99), E.FilteredSearchConfig.pre_filter()) == []
    # beyond the candidate pipeline: every shard's large-k path, merged
    big = many.search_similar(q, 5000)
    er, es = oc.search(A, q, 5000, 0)
    assert [r.key for r in big] == [f"k{i}" for i in er]
    assert np.array_equal(np.array([r.score for r in big], F), es)


def test_sharded_engine_snapshot_round_trip(E, tmp_path):
    """save_index_binary from a sharded engine, load into a sharded engine: the magnitude check reads every shard"""
    rng = np.random.default_rng(5)
    n, d = 2500, 32
    A = rng.standard_normal((n, d)).astype(F)
    _, many = engines(E)
    many.create_collection("docs", E.VectorCollectionConfig())
    for i in range(n):
        many.store_in_collection("docs", f"k{i}", A[i])
    q = rng.standard_normal(d).astype(F)
    before = many.search_in_collection("docs", q, 15)
    path = str(tmp_path / "docs.nmnidx")
    many.save_index_binary("docs", path)
    _, fresh = engines(E, 2)
    assert fresh.load_index_binary(path) == "docs"
    same(before, fresh.search_in_collection("docs", q, 15))


def test_engine_mirror_is_spread_evenly_over_the_devices(E):
    """get_mirror gives a mirror 50 % spare capacity; the rows HELD must still be dealt evenly (contiguous ranges
    of the capacity put 75 % / 25 % on two GPUs and left the fourth of four empty), and stay so under appends."""
    rng = np.random.default_rng(11)
    n, d = 9000, 48
    A = rng.standard_normal((n, d)).astype(F)
    for G in (2, 4):
        eng = E.VectorEngine(E.VectorEngineConfig(devices=(0,) * G))
        eng.batch_store_embeddings([f"k{i}" for i in range(n)], A)
        q = rng.standard_normal(d).astype(F)
        res = eng.search_similar(q, 25)   # builds the mirror
        er, es = oc.search(A, q, 25, 0)
        assert [r.key for r in res] == [f"k{i}" for i in er]
        rows = eng.mirror_shard_rows(d)
        assert len(rows) == G and sum(rows) == n and max(rows) - min(rows) <= 64, rows
        extra = rng.standard_normal((1000, d)).astype(F)
        for i in range(1000):
            eng.store_embedding(f"x{i}", extra[i])
        res = eng.search_similar(q, 25)
        er, es = oc.search(np.concatenate([A, extra]), q, 25, 0)
        assert [r.key for r in res] == [f"k{i}" if i < n else f"x{i - n}" for i in er]
        rows = eng.mirror_shard_rows(d)
        assert sum(rows) == n + 1000 and max(rows) - min(rows) <= 64, rows
        assert eng.mirror_builds() == 1
