"""GPU edge cases of the SIMILAR TOP-K path through the C ABI: ties, duplicates, the exact-fallback
path, n<k, empty shards, k=NMN_MAX_TOP_K, multi-query batches, tile-threshold mode, error codes."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def check(idx, A, Q, k, metric, mask=None, row_base=0):
    Q = np.atleast_2d(Q)
    out = idx.search(Q, k, metric, mask=mask, with_stats=True)
    rows, scores, counts, stats = out
    for qi in range(Q.shape[0]):
        er, es = oc.search(A, Q[qi], k, metric, mask=mask, row_base=row_base)
        c = er.size
        assert counts[qi] == c, (qi, counts[qi], c)
        assert np.array_equal(rows[qi, :c], er), (qi, rows[qi, :c][:8], er[:8])
        assert np.all(scores[qi, :c] == es)
        assert np.all(rows[qi, c:] == U64_MAX) and np.all(np.isneginf(scores[qi, c:]))
    return stats


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_all_rows_identical_ties_by_row(metric):
    from neumann_amd import GpuFlatIndex
    n, d = 5000, 24
    A = np.tile(np.linspace(-1, 1, d, dtype=np.float32), (n, 1))
    q = np.linspace(1, 2, d, dtype=np.float32)
    with GpuFlatIndex(d, n, single_launch=False) as idx:   # (the pipeline) default cand_cap 4096 < 5000 tied rows -> exact fallback
        idx.upload(A)
        st = check(idx, A, q, 10, metric)
        assert st.fallback_queries == 1
        rows, _, _ = idx.search(q, 10, metric)
        assert list(rows[0]) == list(range(10))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_forced_fallback_small_cand_cap(metric):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(11)
    n, d, k = 3000, 40, 7
    base = rng.standard_normal((30, d)).astype(np.float32)
    A = base[rng.integers(0, 30, n)]          # 30 distinct vectors, ~100 exact copies each
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n, cand_cap=8, single_launch=False) as idx:
        idx.upload(A)
        st = check(idx, A, q, k, metric)
        assert st.fallback_queries == 1
        keep = rng.random(n) < 0.4
        check(idx, A, q, k, metric, mask=oc.mask_from_bool(keep))


def test_mixed_fallback_and_normal_queries_in_one_batch():
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(12)
    n, d, k = 6000, 16, 5
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[1000:5500] = A[1000]                    # 4500 duplicates: a query near them overflows 4096
    Q = np.stack([A[1000] + 0.0, rng.standard_normal(d).astype(np.float32), -A[1000]])
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        st = check(idx, A, Q, k, 0)
        assert 1 <= st.fallback_queries <= 2


@pytest.mark.parametrize("n,k", [(3, 10), (1, 1), (64, 64), (65, 100), (100, 4096)])
def test_n_less_than_or_equal_k(n, k):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n * 31 + k)
    A = rng.standard_normal((n, 12)).astype(np.float32)
    q = rng.standard_normal(12).astype(np.float32)
    with GpuFlatIndex(12, max(n, 1)) as idx:
        idx.upload(A)
        for m in (0, 1, 2):
            check(idx, A, q, k, m)


def test_empty_index_and_empty_mask():
    from neumann_amd import GpuFlatIndex
    with GpuFlatIndex(8, 100) as idx:
        rows, scores, counts = idx.search(np.ones(8, np.float32), 5, 0)
        assert counts[0] == 0 and np.all(rows == U64_MAX) and np.all(np.isneginf(scores))
        A = np.random.default_rng(0).standard_normal((100, 8)).astype(np.float32)
        idx.upload(A)
        check(idx, A, np.ones(8, np.float32), 5, 0, mask=np.zeros(2, dtype=np.uint64))


def test_k_max_and_too_large():
    from neumann_amd import GpuFlatIndex, NeumannGpuError, _capi
    rng = np.random.default_rng(77)
    n, d = 20000, 20
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        check(idx, A, q, 4096, 0)
        check(idx, A, q, 1000, 1)
        check(idx, A, q, 4097, 0)   # beyond the candidate pipeline: the large-k path (full sort), same answer
        with pytest.raises(NeumannGpuError) as e:
            idx.search(q, 0, 0)
        assert e.value.status == _capi.ERR_INVALID_TOP_K and "Invalid top_k" in str(e.value)


@pytest.mark.parametrize("nq", [2, 3, 4, 5, 9])
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_multi_query_batches(nq, metric):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(nq)
    n, d, k = 7000, 72, 20
    A = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        check(idx, A, Q, k, metric)
        check(idx, A, Q, k, metric, mask=oc.mask_from_bool(rng.random(n) < 0.3))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_tile_threshold_mode_large_n(metric):
    from neumann_amd import GpuFlatIndex
    n, d, k = 200_000, 32, 10                # n_tiles = 3125 >= 2k and n > 16384 -> tile maxima mode
    A = oc.synth(5, 0, n, d)
    Q = oc.synth(6, 0, 2, d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(5, n)
        check(idx, A, Q, k, metric)
        rng = np.random.default_rng(1)
        check(idx, A, Q, k, metric, mask=oc.mask_from_bool(rng.random(n) < 0.1))
        check(idx, A, Q, k, metric, mask=oc.mask_from_bool(rng.random(n) < 0.0002))  # fewer valid rows than tiles


def test_dot_product_with_outlier_norms():
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(3)
    n, d = 30000, 48
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[::1000] *= 1000.0                       # huge-norm rows inflate the dot margin (max |v|)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        check(idx, A, q, 50, 2)
        check(idx, A, q, 50, 0)


def test_near_ties_around_rank_k():
    """Rows whose scores differ by single ulps straddle rank k: only the exact rescore orders them."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(8)
    n, d, k = 4000, 768, 10
    A = (rng.standard_normal((n, d)) * 0.01).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    for i in range(40):                       # 40 near-copies of q, perturbed in the last ulps
        v = q.copy()
        j = rng.integers(0, d, 5)
        v[j] = np.nextafter(v[j], np.float32(np.inf if i % 2 else -np.inf))
        A[rng.integers(0, n)] = v
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for m in (0, 1, 2):
            check(idx, A, q, k, m)


def test_upload_errors_and_append():
    from neumann_amd import GpuFlatIndex, NeumannGpuError, _capi
    rng = np.random.default_rng(0)
    A = rng.standard_normal((100, 10)).astype(np.float32)
    with GpuFlatIndex(10, 100) as idx:
        idx.upload(A[:60])
        idx.upload(A[60:])                    # append
        assert idx.rows == 100
        check(idx, A, A[3], 5, 0)
        with pytest.raises(NeumannGpuError) as e:
            idx.upload(A[:1])                 # capacity exceeded
        assert e.value.status == _capi.ERR_CAPACITY
        with pytest.raises(NeumannGpuError) as e:
            idx.upload(np.zeros((1, 11), np.float32), row0=0)
        assert e.value.status == _capi.ERR_DIMENSION_MISMATCH
        B = A.copy()
        B[10:20] = rng.standard_normal((10, 10)).astype(np.float32)
        idx.upload(B[10:20], row0=10)         # overwrite in place: norms recomputed
        check(idx, B, B[15], 5, 0)


def test_device_api_equals_host_api():
    import torch
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(2)
    n, d, k = 9000, 128, 25
    A = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((3, d)).astype(np.float32)
    keep = rng.random(n) < 0.5
    mask = oc.mask_from_bool(keep)
    with GpuFlatIndex(d, n, row_base=1_000_000_000_000) as idx:
        idx.upload_device(torch.from_numpy(A).cuda())
        torch.cuda.synchronize()
        for mk, mt in ((None, None), (mask, torch.from_numpy(mask.view(np.int64)).cuda())):
            hr, hs, hc = idx.search(Q, k, 0, mask=mk)
            dr, ds, dc = idx.search_device(torch.from_numpy(Q).cuda(), k, 0, mask_t=mt)
            torch.cuda.synchronize()
            assert np.array_equal(dr.cpu().numpy().view(np.uint64), hr)
            assert np.array_equal(ds.cpu().numpy(), hs) and np.array_equal(dc.cpu().numpy().view(np.uint32), hc)
            er, es = oc.search(A, Q[0], k, 0, mask=mk, row_base=1_000_000_000_000)
            assert np.array_equal(hr[0], er)


def test_count_exact_certificate():
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(4)
    n, d = 12345, 64
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for m in (0, 1, 2):
            s = oc.scores_all(A, q, m)
            ref = np.sort(s)[::-1][99]
            gt, eq = idx.count_exact(q, ref, m)
            assert gt == int(np.sum(s > ref)) and eq == int(np.sum(s == ref))


# ---- large-k path (k > NMN_MAX_TOP_K): exact scan of every row + full device sort (nmn_sortk.hip) -----------
@pytest.mark.parametrize("n,d", [(5000, 24), (4096, 8), (20011, 40), (70000, 16)])
def test_large_k_matches_oracle(n, d):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n + d)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[10] = A[3]                    # exact duplicates: ties resolve by ascending row id
    A[n // 2] = A[3]
    A[7] = 0.0                      # zero vector: cosine 0.0, still returned
    q = rng.standard_normal(d).astype(np.float32)
    keep = rng.random(n) < 0.6
    mask = oc.mask_from_bool(keep)
    with GpuFlatIndex(d, n + 50, row_base=10**9) as idx:   # spare capacity, global ids
        idx.upload(A)
        for metric in (0, 1, 2):
            for k in (4097, n - 1, n, n + 1000):
                rows, scores, counts = idx.search(q, k, metric)
                er, es = oc.search(A, q, k, metric, row_base=10**9)
                assert counts[0] == min(k, n) == er.size
                assert np.array_equal(rows[0, :er.size], er) and np.all(scores[0, :er.size] == es)
                assert np.all(rows[0, er.size:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isneginf(scores[0, er.size:]))
            rows, scores, counts = idx.search(q, n, metric, mask=mask)
            er, es = oc.search(A, q, n, metric, mask=mask, row_base=10**9)
            assert counts[0] == int(keep.sum()) == er.size
            assert np.array_equal(rows[0, :er.size], er) and np.all(scores[0, :er.size] == es)


@pytest.mark.parametrize("n,d", [(70_000, 1536), (66_000, 4096), (131_072 + 5, 96), (70_000, 40)])
def test_exact_scores_of_every_row_by_the_lane_per_row_kernel(n, d):
    """Shards of >= 2^16 rows whose rows are whole 32-float stages compute the exact score of every row (k > 4096, the exact
    fallback, the certificate) with exact_rows_kernel: a lane per row, the reference's chains in registers (one sequential sum
    for Euclidean, eight strided accumulators for dot / cosine).  Bit-equal to the oracle for every metric, with a bitmap, with
    duplicates, a zero row, a partial last tile; rows of 40 floats (not whole stages) keep the eight-lanes-per-row form."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n + d)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[10] = A[3]
    A[n - 1] = A[3]
    A[7] = 0.0
    Q = rng.standard_normal((2, d)).astype(np.float32)
    keep = rng.random(n) < 0.3
    mask = oc.mask_from_bool(keep)
    k = 5000
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            rows, scores, counts = idx.search(Q, k, metric)
            for qi in range(2):
                er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                assert counts[qi] == k and np.array_equal(rows[qi], er) and np.all(scores[qi] == es), (metric, qi)
            rows, scores, counts = idx.search(Q[0], k, metric, mask=mask)
            er, es = oc.search(A, Q[0], k, metric, mask=mask, nthreads=8, partial=True, native=True)
            assert counts[0] == er.size and np.array_equal(rows[0, :er.size], er) and np.all(scores[0, :er.size] == es), metric
            # the certificate's counts come from the same kernel: rows scoring above / at the k-th
            gt, eq = idx.count_exact(Q[0], float(es[-1]), metric, mask=mask)
            s_all = oc.scores_all(A, Q[0], metric)
            assert gt == int(np.sum(s_all[keep] > es[-1])) and eq == int(np.sum(s_all[keep] == es[-1])), metric


def test_large_k_selected_before_it_is_sorted():
    """k > NMN_MAX_TOP_K on a shard of >= 2^18 rows: while k is a small part of the shard the device-wide radix select picks
    the k best composites and only those are sorted (nmn_sortk.hip); a k too large for that keeps the full sort.  Same lists
    bit for bit: duplicates straddling the cut (ties by row id), a filter that leaves fewer than k rows, NaN / inf scores."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(515)
    n, d = 300_000, 32
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[1000:9000] = A[17]            # 8000 equal scores: the k-th sits inside the run for k = 5000 when q is near A[17]
    A[5, 0] = np.inf
    A[6, 1] = np.nan
    Q = np.stack([A[17] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)])
    few = np.zeros(n, bool)
    few[rng.choice(n, 4500, replace=False)] = True      # fewer rows than k = 5000
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for k in (5000, 70_000, 200_000):               # select + sort of 8192 / 131072 keys; full sort of 2^19
            for metric in (0, 1, 2):
                rows, scores, counts = idx.search(Q, k, metric)
                for qi in range(2):
                    er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                    assert counts[qi] == er.size == k
                    assert np.array_equal(rows[qi], er), (k, metric, qi)
                    assert np.array_equal(scores[qi].view(np.uint32), es.view(np.uint32)), (k, metric, qi)
        for sel, keep in ((0.3, rng.random(n) < 0.3), (0.015, few)):
            mask = oc.mask_from_bool(keep)
            rows, scores, counts = idx.search(Q[0], 5000, 0, mask=mask)
            er, es = oc.search(A, Q[0], 5000, 0, mask=mask, nthreads=8, partial=True, native=True)
            assert counts[0] == er.size == min(5000, int(keep.sum()))
            assert np.array_equal(rows[0, :er.size], er) and np.array_equal(scores[0, :er.size].view(np.uint32), es.view(np.uint32)), sel
            assert np.all(rows[0, er.size:] == np.uint64(2**64 - 1))


def test_large_k_multi_query_and_special_values():
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(404)
    n, d, k = 9000, 12, 6000
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[5, 0] = np.inf                # inf / nan scores rank as in the candidate pipeline: NaN below -inf
    A[6, 1] = np.nan
    A[8] = -A[9]
    Q = rng.standard_normal((3, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            rows, scores, counts = idx.search(Q, k, metric)
            small_rows, small_scores, _ = idx.search(Q, 4096, metric)      # candidate pipeline on the same data
            for qi in range(3):
                assert counts[qi] == k
                assert np.array_equal(rows[qi, :4096], small_rows[qi])
                assert np.array_equal(scores[qi, :4096], small_scores[qi], equal_nan=True)
                er, es = oc.search(A, Q[qi], k, metric)
                ok = ~np.isnan(es)
                assert np.array_equal(rows[qi][ok], er[ok]) and np.all(scores[qi][ok] == es[ok])


def test_large_k_on_empty_and_tiny_shards():
    from neumann_amd import GpuFlatIndex
    with GpuFlatIndex(4, 100) as idx:
        rows, scores, counts = idx.search(np.ones(4, np.float32), 5000, 0)
        assert counts[0] == 0 and np.all(rows == np.uint64(0xFFFFFFFFFFFFFFFF))
        idx.upload(np.eye(4, dtype=np.float32))
        rows, scores, counts = idx.search(np.array([1, 0.5, 0, 0], np.float32), 5000, 0)
        assert counts[0] == 4 and rows[0, :4].tolist() == [0, 1, 2, 3]


def test_bf16_pass_overflow_is_retried_in_f32():
    """6000 rows packed within 2e-4 of each other in cosine: the bf16-mirror pass (margin ~3e-3) admits more than the
    4096-row candidate capacity, the f32 retry sweep (margin ~1e-5) does not — the answer is the oracle's and no query
    ends in the exact-scan fallback."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(99)
    n, d, k = 300_000, 64, 50
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    crowd = rng.choice(n, 6000, replace=False)
    A[crowd] = (q[None, :] + 0.05 * rng.standard_normal((6000, d))).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 2, 1):
            rows, scores, counts, stats = idx.search(q, k, metric, with_stats=True)
            er, es = oc.search(A, q, k, metric)
            assert counts[0] == k and np.array_equal(rows[0], er) and np.all(scores[0] == es)
            assert stats.fallback_queries == 0, metric
            assert stats.bytes_scanned == n * d * 2          # the (first) sweep read the bf16 mirror


def test_mirror_switches_itself_off_when_its_margin_is_useless():
    """One row of enormous norm makes the mirror's Euclidean margin (2 max|e_r|) swallow every row: each query then pays
    a bf16 pass plus the f32 retry.  After 256 searches the shard notices (select_kernel counts retries) and sweeps the
    f32 corpus directly; answers are the oracle's throughout."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(123)
    n, d, k = 270_000, 32, 20
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[1234] *= np.float32(1e6)
    Q = rng.standard_normal((8, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        seen = set()
        for i in range(300):
            rows, scores, counts, stats = idx.search(Q[i % 8], k, 1, with_stats=True)
            seen.add(stats.bytes_scanned // (n * d))
            if i in (0, 100, 299):
                er, es = oc.search(A, Q[i % 8], k, 1)
                assert counts[0] == k and np.array_equal(rows[0], er) and np.all(scores[0] == es)
                assert stats.fallback_queries == 0          # the retry, not the exact scan of everything
        assert seen == {2, 4} and stats.bytes_scanned == n * d * 4


def test_crowd_of_duplicates_on_a_large_shard_is_resolved_from_the_crowd_list():
    """20 000 exact copies of one row (far more than the candidate capacity) on a shard large enough for the crowd
    path: the rows within the margin go to the crowd list, are re-scored exactly there, and the ties at the cut are
    broken by row id — no exact scan of the whole shard (fallback_queries == 0), answers as the oracle's."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(17)
    n, d, k = 300_000, 128, 64
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    dup = np.sort(rng.choice(n, 20_000, replace=False))
    A[dup] = (q * np.float32(1.5)).astype(np.float32)          # identical rows: 20 000 ties at the top
    near = rng.choice(np.setdiff1d(np.arange(n), dup), 9000, replace=False)
    A[near] = (q[None, :] * np.float32(1.5) + 1e-3 * rng.standard_normal((9000, d))).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            rows, scores, counts, stats = idx.search(q, k, metric, with_stats=True)
            er, es = oc.search(A, q, k, metric)
            assert counts[0] == k and np.array_equal(rows[0], er) and np.all(scores[0] == es), metric
            assert stats.fallback_queries == 0, metric
            assert stats.candidates_rescored >= 20_000, (metric, stats.candidates_rescored)
        Q = np.stack([q, q * np.float32(0.5), rng.standard_normal(d).astype(np.float32), q + np.float32(1e-3)])
        rows, scores, counts = idx.search(Q, k, 0)
        for i in range(4):
            er, es = oc.search(A, Q[i], k, 0)
            assert np.array_equal(rows[i], er) and np.all(scores[i] == es), i


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_large_shard_exact_fallback_uses_the_device_wide_select(metric):
    """Shards of >= 2^18 rows select the exact fallback's top-k with the whole device (fallback_select_kernel: per-workgroup
    LDS histograms -> global histogram -> grid barrier, six radix digits at most): all rows identical (every row within any
    margin: crowd path and f32 retry both give up), then a few hundred distinct values with thousands of exact copies each,
    with a filter, with k up to NMN_MAX_TOP_K, and two flagged queries next to a normal one in one batch."""
    from neumann_amd import GpuFlatIndex
    n, d = 300_000, 32
    rng = np.random.default_rng(31 + metric)
    A = np.tile(np.linspace(-1, 1, d, dtype=np.float32), (n, 1))
    q = np.linspace(1, 2, d, dtype=np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        st = check(idx, A, q, 10, metric)
        assert st.fallback_queries == 1
        rows, _, _ = idx.search(q, 4096, metric)
        assert list(rows[0]) == list(range(4096))      # ties by ascending row id, k = NMN_MAX_TOP_K
        keep = rng.random(n) < 0.01
        check(idx, A, q, 50, metric, mask=oc.mask_from_bool(keep))
    base = rng.standard_normal((200, d)).astype(np.float32)
    A = base[rng.integers(0, 200, n)]                  # ~1500 exact copies of each of 200 vectors
    Q = np.stack([base[3] + np.float32(0.01), base[7] * np.float32(2.0), rng.standard_normal(d).astype(np.float32)])
    with GpuFlatIndex(d, n, cand_cap=64) as idx:
        idx.upload(A)
        rows, scores, counts, st = idx.search(Q, 300, metric, with_stats=True)   # (crowds of ~1500 copies: the crowd list takes them)
        for qi in range(3):
            er, es = oc.search(A, Q[qi], 300, metric, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es), qi


def test_device_wide_select_gives_up_its_barrier_instead_of_hanging():
    """fallback_select_kernel's grid barrier is a bounded wait (nothing but an idle device guarantees that the 64
    workgroups of several concurrent launches are all resident).  With NMN_FB_TIMEOUT_TICKS=1 every barrier wait that is not
    satisfied at its first look runs out of patience: the launch raises its abort flag, the flagged query keeps
    overflow == 1 and final_kernel selects it with one workgroup — the same answer, and the next search (counters zeroed
    by select_kernel) is not disturbed by the aborted one.  Runs in a child process: the knob is read once per process."""
    import subprocess
    import sys
    code = r'''
import numpy as np
from oracle import oracle_c as oc
from neumann_amd import GpuFlatIndex
n, d = 300_000, 32
A = np.tile(np.linspace(-1, 1, d, dtype=np.float32), (n, 1))
q = np.linspace(1, 2, d, dtype=np.float32)
with GpuFlatIndex(d, n) as idx:
    idx.upload(A)
    for rep in range(3):
        for metric in (0, 1, 2):
            rows, scores, counts, st = idx.search(q, 10, metric, with_stats=True)
            er, es = oc.search(A, q, 10, metric, nthreads=8, partial=True, native=True)
            assert st.fallback_queries == 1
            assert np.array_equal(rows[0], er) and np.all(scores[0] == es), (rep, metric)
    rows, _, _ = idx.search(q, 6000, 0)    # large-k path: cooperative launch (or the full sort), same list
    assert list(rows[0]) == list(range(6000))
print("ok")
'''
    env = dict(os.environ, NMN_FB_TIMEOUT_TICKS="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


_WS_CHILD = r"""
import sys
import numpy as np
from oracle import oracle_c as oc
from neumann_amd import GpuFlatIndex
n, d, k, nq = 300_000, 256, 50, 64
A = oc.synth(0x77, 0, n, d, nthreads=8)
Q = oc.synth(0x78, 0, nq, d)
keep = np.random.default_rng(3).random(n) < 0.4
with GpuFlatIndex(d, n) as idx:
    idx.fill_synthetic(0x77, n)
    for metric, mask in ((0, None), (1, oc.mask_from_bool(keep)), (2, None)):
        rows, scores, counts = idx.search(Q, k, metric, mask=mask)
        for qi in range(nq):
            er, es = oc.search(A, Q[qi], k, metric, mask=mask, nthreads=8, partial=True, native=True)
            assert counts[qi] == er.size and np.array_equal(rows[qi, :er.size], er), (metric, qi)
            assert np.array_equal(scores[qi, :er.size].view(np.uint32), es.view(np.uint32)), (metric, qi)
    r1, s1, c1 = idx.search(Q[:3], k, 0)      # a smaller batch afterwards: the learned pass size still serves
    r0, s0, c0 = idx.search(Q, k, 0)
    assert np.array_equal(r1, r0[:3]) and np.array_equal(s1.view(np.uint32), s0[:3].view(np.uint32))
print("WS-OK")
"""


def test_query_passes_shrink_when_the_workspace_does_not_fit():
    """A device too full for the score matrix of a 64-query pass (nq x rows x 4 B: 2.6 GB at 10M rows — config 4 on one GPU)
    must serve the batch as more passes of fewer queries, not fail with NMN_ERR_OUT_OF_MEMORY (ws_alloc, nmn_api.hip).  The test
    hook NMN_WS_TEST_MAX_NQ=8 makes every pass of more than 8 queries "not fit": 64 -> 16 -> 4 queries per pass; all 64 answers of
    three metrics (one under a bitmap) must be the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMN_WS_TEST_MAX_NQ="8")
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _WS_CHILD], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WS-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_hbm_bytes_is_a_plain_getter_and_retry_declined_is_the_explicit_form():
    """nmn_index_hbm_bytes has no side effects any more (a monitor polling it used to clear the shard's out-of-memory
    verdicts); nmn_index_retry_declined is the entry point for a host that has just freed device memory.  Both are callable at any
    time, between searches whose answers do not change."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(41)
    n, d, k = 40_000, 256, 10
    A = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((8, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        r0, s0, c0 = idx.search(Q, k, 0)
        a = idx.hbm_bytes()
        for _ in range(5):
            assert idx.hbm_bytes() == a
        idx.retry_declined()
        r1, s1, c1 = idx.search(Q, k, 0)
        assert np.array_equal(r0, r1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(c0, c1)
        for i in range(8):
            er, es = oc.search(A, Q[i], k, 0)
            assert np.array_equal(r1[i], er) and np.all(s1[i] == es)
