"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

METRICS = [0, 1, 2]


def _check(idx, A, q, k, metric, mask=None, row_base=0):
    rows, scores, counts = idx.search(q, k, metric, mask=mask)
    er, es = oc.search(A, q, k, metric, mask=mask, row_base=row_base)
    assert counts[0] == er.size, (counts[0], er.size)
    c = er.size
    assert np.array_equal(rows[0, :c], er), (rows[0, :c][:10], er[:10])
    # scores are produced by the exact-rescore kernel: compare as floats (==) — -0.0 vs +0.0 allowed
    assert np.all(scores[0, :c] == es), np.abs(scores[0, :c] - es).max()
    assert np.all(rows[0, c:] == np.uint64(0xFFFFFFFFFFFFFFFF))
    assert np.all(np.isneginf(scores[0, c:]))


@pytest.mark.parametrize("single_launch", [True, False])  # small shards: the one-kernel search, and the general pipeline
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,d,k", [(1000, 128, 5), (4096, 768, 100), (777, 36, 10), (100, 7, 3), (50, 2, 60),
                                   (20000, 64, 10), (3000, 1536, 100)])
def test_search_matches_oracle(metric, n, d, k, single_launch):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(1234 + n + d)
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n, single_launch=single_launch) as idx:
        idx.upload(A)
        assert idx.rows == n
        _check(idx, A, q, k, metric)


@pytest.mark.parametrize("metric", METRICS)
def test_search_with_mask(metric):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(7)
    n, d, k = 5000, 128, 50
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for sel in (0.5, 0.1, 0.001, 0.0):
            keep = rng.random(n) < sel
            mask = oc.mask_from_bool(keep)
            _check(idx, A, q, k, metric, mask=mask)


def test_norms_bit_exact():
    import ctypes as C
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(3)
    n, d = 513, 77
    A = (rng.standard_normal((n, d)) * 10).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        q = rng.standard_normal(d).astype(np.float32)
        rows = np.arange(n, dtype=np.uint64)
        for metric in METRICS:
            got = idx.score_rows(q, rows, metric)[0]
            exp = oc.scores_all(A, q, metric)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), metric


def test_synth_matches_host():
    from neumann_amd import GpuFlatIndex, synth_rows
    n, d = 300, 40
    host = synth_rows(99, 1000, n, d)
    assert np.array_equal(host, oc.synth(99, 1000, n, d))
    with GpuFlatIndex(d, n, row_base=1000) as idx:
        idx.fill_synthetic(99, n)
        q = host[5]
        rows, scores, counts = idx.search(q, 3, 0)
        assert rows[0, 0] == 1005
        er, es = oc.search(host, q, 3, 0, row_base=1000)
        assert np.array_equal(rows[0], er) and np.all(scores[0] == es)


@pytest.mark.parametrize("n,d", [(5000, 768), (4097, 128), (6000, 96), (4096, 32), (300, 1536), (70, 64)])
def test_one_pass_ingest_magnitudes_and_mirror(n, d):
    """Rows in whole 32-float stages take the one-pass ingest kernel (nmn_ingest.hip): magnitudes must equal
    simd::magnitude bit for bit (cosine scores of EVERY row, which divide by them, equal the oracle's bits), and the bf16
    mirror it writes in the same pass must serve exact searches — after a bulk upload (mirror built by the upload), an
    append, an overwrite in place and a single-row patch."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n * 31 + d)
    A = (rng.standard_normal((n, d)) * rng.choice([1e-3, 1.0, 50.0], size=(n, 1))).astype(np.float32)
    A[3] = 0.0
    extra = rng.standard_normal((137, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n + 200) as idx:
        idx.upload(A)
        idx.upload(extra)                      # append: extends norms (and the mirror when the bulk upload built it)
        A = np.concatenate([A, extra])
        patch = rng.standard_normal((11, d)).astype(np.float32) * np.float32(3.0)
        idx.upload(patch, row0=10)             # overwrite in place
        A[10:21] = patch
        idx.set_row(n // 2, q * np.float32(2.0))   # single-row patch: the query's direction becomes the best match
        A[n // 2] = q * np.float32(2.0)
        rows = np.arange(A.shape[0], dtype=np.uint64)
        for metric in (0, 2):
            got = idx.score_rows(q, rows, metric)[0]
            exp = oc.scores_all(A, q, metric)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), metric
        for metric in METRICS:
            _check(idx, A, q, 50, metric)
        r, s, c = idx.search(q, 1, 0)
        assert r[0, 0] == n // 2


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d", [(1, 8), (63, 3), (64, 128), (65, 128), (255, 40), (1000, 128), (10_000, 128), (10_007, 768),
                                 (65_536, 64), (40_000, 384)])
def test_single_launch_search_of_small_shards(metric, n, d):
    """Shards of <= 65 536 rows answer a lone query in ONE launch (tiny_search_kernel: exact scores of every row, per-workgroup
    sort, the last workgroup merges; query in the kernel arguments, results written to pinned host memory).  Against the
    oracle for k below / at / above the rows of a workgroup and of the shard, with ties (blocks of identical rows) and a
    DEVICE bitmap; stats say which sweep ran (exact scores from the f32 rows: 4 bytes per element)."""
    import torch
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n * 7 + d + metric)
    A = rng.standard_normal((n, d)).astype(np.float32)
    if n >= 64:
        A[n // 3:n // 3 + 20] = A[n // 3]       # 20 identical rows: equal scores come back in row order
        A[5] = 0.0                                # a zero row
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n, row_base=10**10) as idx:
        idx.upload(A)
        for k in sorted({1, 5, 100, min(n, 300), 1000, 1024}):
            rows, scores, counts, st = idx.search(q, k, metric, with_stats=True)
            er, es = oc.search(A, q, k, metric, row_base=10**10)
            c = er.size
            assert counts[0] == c and np.array_equal(rows[0, :c], er) and np.all(scores[0, :c] == es), (k, c)
            assert np.all(rows[0, c:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isneginf(scores[0, c:]))
            assert st.bytes_scanned == n * d * 4 and st.fallback_queries == 0
        # the same row twice as the query: cosine 1.0 first, ties by row
        r, s_, c_ = idx.search(A[n // 3], 3, metric)
        er, es = oc.search(A, A[n // 3], 3, metric, row_base=10**10)
        assert np.array_equal(r[0, :er.size], er) and np.all(s_[0, :er.size] == es)
        # device bitmap (what the predicate kernel / an IVF probe hands over)
        for sel in (0.5, 0.02, 0.0):
            keep = rng.random(n) < sel
            m = oc.mask_from_bool(keep)
            mt = torch.from_numpy(m.view(np.int64)).cuda()
            rows, scores, counts = idx.search_dmask(q, 10, metric, mt.data_ptr())
            er, es = oc.search(A, q, 10, metric, mask=m, row_base=10**10)
            assert counts[0] == er.size and np.array_equal(rows[0, :er.size], er) and np.all(scores[0, :er.size] == es), sel
