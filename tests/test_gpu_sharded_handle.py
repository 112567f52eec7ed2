"""nmn_sharded — ONE process driving several row-range shards through the C ABI (include/neumann_gpu.h): what the Rust
host, a single process sharing an Arc<VectorEngine> (query_router/src/lib.rs:710), binds to use every GPU of a node.
Reference semantics of the gather step: ResultMerger::merge_top_k, query_router/src/distributed.rs:413-433.

On the 1-GPU box: one shard through RCCL (ncclCommInitAll with one rank, ncclAllGather, device merge) and S LOGICAL shards
on device 0 (peer-copy gather) against the committed 8-shard golden fixture and the oracle; with >= 2 GPUs the same
tests also run one shard per device over RCCL.
"""
import numpy as np
import pytest

from oracle import oracle_c as oc
from tests import _golden

pytestmark = pytest.mark.gpu

NO_ROW = np.uint64(0xFFFFFFFFFFFFFFFF)


def _n_gpus():
    import ctypes as C
    from neumann_amd import _capi
    n = C.c_int32(0)
    _capi.load().nmn_device_count(C.byref(n))
    return n.value


def _check(sh, A, Q, k, metric, mask=None, row_base=0):
    rows, scores, counts = sh.search(Q, k, metric, mask=mask)
    for qi in range(Q.shape[0]):
        er, es = oc.search(A, Q[qi], k, metric, mask=mask, row_base=row_base)
        c = er.size
        assert counts[qi] == c, (qi, counts[qi], c)
        assert np.array_equal(rows[qi, :c], er), (metric, qi)
        assert np.all(scores[qi, :c] == es), (metric, qi)
        assert np.all(rows[qi, c:] == NO_ROW) and np.all(np.isneginf(scores[qi, c:]))


def test_one_shard_through_rccl_matches_oracle():
    """n_shards = 1 with NMN_GATHER_RCCL: the communicator, the grouped all-gather and the device merge are the real ones."""
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import GATHER_RCCL
    n, d, k = 30000, 128, 50
    A = oc.synth(31, 500, n, d)
    Q = oc.synth(32, 0, 3, d)
    with GpuShardedIndex(d, n, 1, devices=[0], row_base=500, gather=GATHER_RCCL) as sh:
        assert sh.gather_mode == GATHER_RCCL
        sh.fill_synthetic(31, n)
        assert sh.rows == n
        for metric in (0, 1, 2):
            _check(sh, A, Q, k, metric, row_base=500)
        keep = np.random.default_rng(5).random(n) < 0.3
        _check(sh, A, Q, k, 0, mask=oc.mask_from_bool(keep), row_base=500)


def per_gap(n, n_shards):
    """first row of the second shard: an upload there while the first shard is empty leaves a gap"""
    return -(-n // n_shards)


@pytest.mark.parametrize("crew", [False, True])
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_logical_shards_on_one_device_match_oracle(n_shards, crew, monkeypatch):
    """S shards on device 0 (peer-copy gather): uploads that straddle shard boundaries, shard ranges that are not multiples
    of 64 (the global bitmap is re-sliced bit by bit), duplicates across shards (ties by global row id).  crew: the handle
    drives every shard from a host thread of its own (what it does when the shards sit on different GPUs; forced here)."""
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import NeumannGpuError
    from neumann_amd._capi import GATHER_PEER
    monkeypatch.setenv("NMN_SHARDED_CREW", "1" if crew else "0")
    n, d, k = 10007, 96, 64
    rng = np.random.default_rng(77 + n_shards)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[5000:5040] = A[17]           # the same vector in several shards: equal scores must come back in row order
    A[n - 1] = A[17]
    Q = np.stack([A[17] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)])
    with GpuShardedIndex(d, n, n_shards, devices=[0] * n_shards, gather=GATHER_PEER) as sh:
        assert sh.gather_mode == GATHER_PEER
        with pytest.raises(NeumannGpuError):   # a gap in the global numbering is refused before any shard is touched
            sh.upload(A[:10], row0=per_gap(n, n_shards))
        assert sh.rows == 0
        for a, b in ((0, 3000), (3000, 3001), (3001, n)):   # pieces that cross shard boundaries
            sh.upload(A[a:b], row0=a)
        per = -(-n // n_shards)
        assert [sh.shard_rows(g) for g in range(n_shards)] == [min(per, n - g * per) for g in range(n_shards)]
        for metric in (0, 1, 2):
            _check(sh, A, Q, k, metric)
        for sel in (0.5, 0.02, 0.0):
            keep = rng.random(n) < sel
            _check(sh, A, Q, k, 0, mask=oc.mask_from_bool(keep))
        _check(sh, A, Q, 5000, 1)  # k beyond NMN_MAX_TOP_K: every shard takes its large-k path, the merge has no size limit


def test_eight_logical_shards_reproduce_the_golden_fixture():
    """The committed 8-shard fixture (tests/golden/make_golden.py): the sharded handle must return the UNSHARDED lists."""
    from neumann_amd import GpuShardedIndex
    g = _golden.load("synth_4096x768_top100.npz")
    A = _golden.rebuild_corpus(g, oc.synth)
    n, d = A.shape
    k = int(g["k"])
    with GpuShardedIndex(d, n, 8, devices=[0] * 8) as sh:
        sh.upload(A)
        for metric, qi, tag, mask, exp_rows, exp_scores in _golden.synth_cases(g):
            rows, scores, counts = sh.search(g["Q"][qi], k, metric, mask=mask)
            c = exp_rows.size
            assert counts[0] == c
            assert np.array_equal(rows[0, :c], exp_rows) and np.array_equal(scores[0, :c].view(np.uint32), exp_scores.view(np.uint32))


def test_batched_queries_and_stats():
    from neumann_amd import GpuShardedIndex
    n, d, k, nq = 200_000, 768, 100, 64
    A = oc.synth(41, 0, n, d, nthreads=8)
    Q = oc.synth(42, 0, nq, d)
    with GpuShardedIndex(d, n, 4, devices=[0] * 4) as sh:
        sh.fill_synthetic(41, n)
        sh.set_timing(True)
        rows, scores, counts, st = sh.search(Q, k, 0, with_stats=True)
        assert st.rows_scanned == n and st.bytes_scanned == n * d and st.scan_ms > 0   # (every shard's batch sweeps its 8-bit mirror)
        assert sh.last_gather_ms() >= 0
        for qi in (0, 31, 63):
            er, es = oc.search(A, Q[qi], k, 0, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es)
        sh.set_mirror(False)  # the f32-corpus sweep on every shard: same answers
        r2, s2, c2, st2 = sh.search(Q[:2], k, 0, with_stats=True)
        assert st2.bytes_scanned == n * d * 4
        assert np.array_equal(r2, rows[:2]) and np.array_equal(s2.view(np.uint32), scores[:2].view(np.uint32))
        # ... and the whole batch: the matrix-core sweep over every shard's f32 rows (round 5) — all 64 lists unchanged
        r3, s3, c3, st3 = sh.search(Q, k, 0, with_stats=True)
        assert st3.bytes_scanned == n * d * 4 and st3.fallback_queries == 0
        assert np.array_equal(r3, rows) and np.array_equal(s3.view(np.uint32), scores.view(np.uint32)) and np.array_equal(c3, counts)


def test_argument_errors():
    from neumann_amd import GpuShardedIndex, NeumannGpuError, _capi
    with pytest.raises(NeumannGpuError) as e:
        GpuShardedIndex(8, 100, 2, devices=[0, 0], gather=_capi.GATHER_RCCL)  # one communicator rank per GPU
    assert e.value.status == _capi.ERR_INVALID_ARGUMENT
    with pytest.raises(NeumannGpuError) as e:
        GpuShardedIndex(8, 100, 0)
    assert e.value.status == _capi.ERR_INVALID_ARGUMENT
    with pytest.raises(NeumannGpuError) as e:
        GpuShardedIndex(8, 100, 2, devices=[0, 99])
    assert e.value.status == _capi.ERR_NO_DEVICE
    with GpuShardedIndex(8, 100, 2, devices=[0, 0]) as sh:
        with pytest.raises(NeumannGpuError) as e:
            sh.upload(np.zeros((101, 8), np.float32), row0=0)
        assert e.value.status == _capi.ERR_CAPACITY
        with pytest.raises(NeumannGpuError) as e:
            sh.search(np.ones(8, np.float32), 0)
        assert e.value.status == _capi.ERR_INVALID_TOP_K
        rows, scores, counts = sh.search(np.ones(8, np.float32), 3)  # empty shards: nothing found, padded
        assert counts[0] == 0 and np.all(rows == NO_ROW)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs in this process (the 1-GPU box covers RCCL with one rank)")
def test_one_shard_per_device_over_rccl():
    """Config 4's shape at small scale: one shard per GPU of the node, RCCL all-gather over xGMI, merge on device 0."""
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import GATHER_RCCL
    G = _n_gpus()
    n, d, k = 100_000 * G, 768, 100
    A = oc.synth(51, 0, n, d, nthreads=16)
    Q = oc.synth(52, 0, 4, d)
    with GpuShardedIndex(d, n, G, devices=list(range(G))) as sh:
        assert sh.gather_mode == GATHER_RCCL
        sh.fill_synthetic(51, n)
        for metric in (0, 1):
            rows, scores, counts = sh.search(Q, k, metric)
            for qi in range(4):
                er, es = oc.search(A, Q[qi], k, metric, nthreads=16, partial=True, native=True)
                assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es)


def test_concurrent_callers_of_the_handle_share_sweeps():
    """Many threads searching one nmn_sharded handle: calls that arrive while a search runs leave together as one query
    batch (nmn_sharded_coalesce_stats), writers run alone in between, and every caller gets bit for bit what a lone call
    returns — different k per caller, a masked caller and a k = 5000 caller among them (never merged), an upload in the middle."""
    import threading
    from neumann_amd import GpuShardedIndex
    n, d = 300_000, 64
    A = oc.synth(91, 0, n, d, nthreads=8)
    Q = oc.synth(92, 0, 48, d)
    keep = np.random.default_rng(4).random(n) < 0.4
    mask = oc.mask_from_bool(keep)
    with GpuShardedIndex(d, n + 1000, 3, devices=[0, 0, 0]) as s:
        s.upload(A)
        ks = [5 + (t % 4) * 20 for t in range(48)]
        lone = [s.search(Q[t], ks[t], t % 3) for t in range(48)]
        lone_masked = s.search(Q[0], 10, 0, mask=mask)
        lone_large = s.search(Q[1], 5000, 0)   # k > NMN_MAX_TOP_K: the large-k path, never merged with other callers
        errors, lock = [], threading.Lock()
        start = threading.Barrier(26)

        def caller(t):
            try:
                start.wait()
                for rep in range(6):
                    qi = (t * 2 + rep) % 48
                    r, sc, c = s.search(Q[qi], ks[qi], qi % 3)
                    er, es, ec = lone[qi]
                    assert np.array_equal(r, er) and np.array_equal(sc.view(np.uint32), es.view(np.uint32)) and np.array_equal(c, ec), (t, rep)
            except Exception as e:  # noqa: BLE001
                with lock:
                    errors.append(repr(e))

        def masked_caller():
            try:
                start.wait()
                for rep in range(4):
                    r, sc, c = s.search(Q[0], 10, 0, mask=mask)
                    assert np.array_equal(r, lone_masked[0]) and np.array_equal(sc.view(np.uint32), lone_masked[1].view(np.uint32))
            except Exception as e:  # noqa: BLE001
                with lock:
                    errors.append(repr(e))

        def large_k_caller():
            try:
                start.wait()
                for rep in range(3):
                    r, sc, c = s.search(Q[1], 5000, 0)
                    assert np.array_equal(r, lone_large[0]) and np.array_equal(sc.view(np.uint32), lone_large[1].view(np.uint32))
                    assert np.array_equal(c, lone_large[2])
            except Exception as e:  # noqa: BLE001
                with lock:
                    errors.append(repr(e))

        for attempt in range(5):   # whether calls meet is timing: hammer again if none did
            threads = [threading.Thread(target=caller, args=(t,)) for t in range(24)] + [threading.Thread(target=masked_caller),
                                                                                         threading.Thread(target=large_k_caller)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            assert not errors, errors[:3]
            if s.coalesce_stats()[0] >= 1:
                break
            start.reset()
        batches, calls = s.coalesce_stats()
        assert batches >= 1 and calls >= 2 * batches, (batches, calls)
        # a writer between searches: appended rows are found afterwards, by every later caller
        extra = oc.synth(93, 0, 500, d)
        s.upload(extra, row0=n)
        A2 = np.concatenate([A, extra])
        r, sc, c = s.search(extra[7], 3, 0)
        er, es = oc.search(A2, extra[7], 3, 0, nthreads=8, partial=True, native=True)
        assert np.array_equal(r[0], er) and np.all(sc[0] == es)


# ---- cyclic layout + create-time self-test (round 3) --------------------------------------------------------------------
@pytest.mark.parametrize("crew", [False, True])
@pytest.mark.parametrize("n_shards", [2, 3, 5])
def test_cyclic_layout_matches_oracle_and_is_balanced(n_shards, crew, monkeypatch):
    """NMN_SHARDED_LAYOUT_CYCLIC: 64-row blocks dealt round-robin.  Global row ids in and out, uploads in pieces that start
    and end inside blocks, a capacity far above the rows held (what an engine mirror looks like): every shard holds its
    share of the rows HELD, and every answer is the unsharded one (duplicates across shards tie by GLOBAL row id)."""
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import NeumannGpuError
    monkeypatch.setenv("NMN_SHARDED_CREW", "1" if crew else "0")
    n, d, k, cap, base = 10007, 96, 64, 25000, 1000
    rng = np.random.default_rng(177 + n_shards)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[5000:5040] = A[17]
    A[n - 1] = A[17]
    Q = np.stack([A[17] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)])
    with GpuShardedIndex(d, cap, n_shards, devices=[0] * n_shards, row_base=base, cyclic=True) as sh:
        assert sh.layout == 1
        with pytest.raises(NeumannGpuError):
            sh.upload(A[:10], row0=64)     # a gap in the global numbering
        assert sh.rows == 0
        for a, b in ((0, 3000), (3000, 3001), (3001, 3070), (3070, n)):
            sh.upload(A[a:b], row0=a)
        assert sh.rows == n
        blocks = -(-n // 64)
        held = [sh.shard_rows(g) for g in range(n_shards)]
        expect = [sum(min(64, n - b * 64) for b in range(g, blocks, n_shards)) for g in range(n_shards)]
        assert held == expect and max(held) - min(held) <= 64
        for metric in (0, 1, 2):
            _check(sh, A, Q, k, metric, row_base=base)
        for sel in (0.5, 0.02, 0.0):
            keep = rng.random(n) < sel
            _check(sh, A, Q, k, 0, mask=oc.mask_from_bool(keep), row_base=base)
        _check(sh, A, Q, 5000, 1, row_base=base)            # large-k path per shard, ids remapped all the same
        A[4000:4100] = rng.standard_normal((100, d)).astype(np.float32)
        sh.upload(A[4000:4100], row0=4000)                  # overwrite in place, across blocks
        _check(sh, A, Q, k, 0, row_base=base)
        B = rng.standard_normal((300, d)).astype(np.float32)
        sh.upload(B, row0=n)                                # append
        _check(sh, np.concatenate([A, B]), Q, k, 2, row_base=base)


def test_cyclic_fill_synthetic_equals_the_unsharded_corpus():
    from neumann_amd import GpuShardedIndex
    n, d, k = 20000, 128, 30
    A = oc.synth(71, 300, n, d)
    Q = oc.synth(72, 0, 2, d)
    with GpuShardedIndex(d, n + 5000, 4, devices=[0] * 4, row_base=300, cyclic=True) as sh:
        sh.fill_synthetic(71, n)
        for metric in (0, 1, 2):
            _check(sh, A, Q, k, metric, row_base=300)


def test_create_runs_the_collective_selftest():
    """>= 2 shards: create ends with every shard's rank travelling through the gather a search uses.  One device: peer
    copies (rccl_ranks == 0); with >= 2 GPUs the RCCL all-gather over distinct devices (rccl_ranks == shards)."""
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import GATHER_PEER, GATHER_RCCL
    with GpuShardedIndex(64, 1000, 4, devices=[0] * 4) as sh:
        assert sh.gather_mode == GATHER_PEER and sh.rccl_ranks == 0
    g = _n_gpus()
    if g >= 2:
        with GpuShardedIndex(64, 1000, g, devices=list(range(g))) as sh:
            assert sh.gather_mode == GATHER_RCCL and sh.rccl_ranks == g
