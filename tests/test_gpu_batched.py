"""Batched queries (config 3): the MFMA sweep (>= 3-5 queries, dim % 128 == 0; cosine / dot directly, Euclidean as
|q|^2 + |v|^2 - 2 q.v) and the VALU multi-sweep path must both return exactly the oracle's rows and scores for
every query."""
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def check_batch(idx, A, Q, k, metric, mask=None):
    rows, scores, counts, st = idx.search(Q, k, metric, mask=mask, with_stats=True)
    for qi in range(Q.shape[0]):
        er, es = oc.search(A, Q[qi], k, metric, mask=mask)
        c = er.size
        assert counts[qi] == c, (qi, counts[qi], c)
        assert np.array_equal(rows[qi, :c], er), (qi, rows[qi, :8], er[:8])
        assert np.all(scores[qi, :c] == es), qi
        assert np.all(rows[qi, c:] == U64_MAX)
    return st


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq,k", [(20000, 768, 64, 100), (20000, 768, 5, 10), (9000, 128, 16, 20),
                                      (30000, 256, 70, 50), (4000, 384, 33, 7), (70000, 512, 64, 100),
                                      (100, 640, 8, 200),
                                      # more than 64 queries on rows of <= 768: 128 stationary queries per workgroup
                                      (20000, 768, 128, 50), (9000, 256, 100, 20), (5000, 640, 65, 10),
                                      (15000, 128, 200, 10), (7000, 384, 129, 30), (6000, 1024, 100, 20), (5000, 1280, 128, 10),
                                      # rows longer than 768 floats: the B-fragments of the 64 stationary queries fill 128-192 VGPRs
                                      (20000, 1024, 64, 50), (12000, 1536, 33, 100), (9000, 1280, 7, 10),
                                      (30000, 1536, 70, 20),
                                      # 2048 / 3072 / 4096: 32 stationary queries, the k-steps of a stage split over wave pairs
                                      (6000, 2048, 40, 10), (5000, 3072, 33, 20), (3000, 3072, 70, 5), (2500, 4096, 40, 10),
                                      # just short of a multiple of 128: the stride is padded up to it (zeros), same sweep
                                      (6000, 1000, 40, 10), (4000, 960, 64, 20), (2500, 3000, 33, 5), (5000, 720, 100, 10),
                                      # ... or of the next row length the sweep is built for (1152 -> 1280, 1408 -> 1536)
                                      (4000, 1152, 40, 10), (3000, 1408, 70, 5)])
def test_mfma_batch_matches_oracle(metric, n, d, nq, k):
    from neumann_amd import GpuFlatIndex
    A = oc.synth(1000 + n + d, 0, n, d)
    Q = oc.synth(2000 + nq, 0, nq, d)
    Q[nq // 2] = A[n // 3]                       # one query equal to a stored row
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(1000 + n + d, n)
        st = check_batch(idx, A, Q, k, metric)
        assert st.fallback_queries == 0
        rng = np.random.default_rng(n)
        check_batch(idx, A, Q, k, metric, mask=oc.mask_from_bool(rng.random(n) < 0.3))


@pytest.mark.parametrize("n,d,nq,k", [(20000, 768, 64, 100), (20011, 768, 5, 10), (9000, 128, 16, 20), (30000, 256, 70, 50),
                                      (4000, 384, 33, 7), (7000, 512, 64, 100), (100, 640, 8, 200),
                                      (20000, 768, 128, 50), (9000, 256, 100, 20), (5000, 640, 65, 10), (7000, 384, 129, 30),
                                      (6000, 1024, 100, 20), (20000, 1024, 64, 50), (12000, 1536, 33, 100), (9000, 1280, 7, 10),
                                      (6000, 2048, 40, 10), (5000, 3072, 33, 20), (2500, 4096, 40, 10),
                                      (6000, 1000, 40, 10), (3000, 1408, 70, 5)])
def test_mfma_batch_over_the_f32_rows_matches_oracle(n, d, nq, k):
    """nmn_index_set_mirror(0): batches take the matrix-core sweep over the ROW-MAJOR F32 CORPUS (nmn_scan_mfma_f32.hip: rows
    rounded to bf16 in registers, a-priori rounding bound in the margin) — config 3 as SURVEY §8(d) prices it, and what a shard
    whose mirror did not fit runs.  Every query's rows and scores are the oracle's, all metrics, with and without a bitmap."""
    from neumann_amd import GpuFlatIndex
    A = oc.synth(1000 + n + d, 0, n, d)
    Q = oc.synth(2000 + nq, 0, nq, d)
    Q[nq // 2] = A[n // 3]                       # one query equal to a stored row
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.set_mirror(0)                        # (before the rows arrive: no mirror is ever built)
        idx.fill_synthetic(1000 + n + d, n)
        for metric in (0, 1, 2):
            st = check_batch(idx, A, Q, k, metric)
            assert st.fallback_queries == 0
            assert st.bytes_scanned == st.rows_scanned * d * 4, "the f32 rows were not what the sweep read"
        rng = np.random.default_rng(n)
        check_batch(idx, A, Q, k, 0, mask=oc.mask_from_bool(rng.random(n) < 0.3))
        assert idx.hbm_bytes()[1] == 0, "a mirror was built under set_mirror(0)"


def test_f32_rows_matrix_core_sweep_on_a_large_shard_with_planted_neighbours():
    """The f32-rows matrix-core sweep with its sampling pass and the two-launch bound refinement (2.2M x 128, 96 queries), near
    copies of some queries planted across the shard, Euclidean near-zero distances included; and the same batch answered by the
    VALU sweeps of four (NMN_NO_F32_MFMA is process-wide, so the comparison is with the oracle)."""
    from neumann_amd import GpuFlatIndex
    n, d, nq, k = 2_200_000, 128, 96, 25
    A = oc.synth(777, 0, n, d, nthreads=8)
    Q = oc.synth(778, 0, nq, d)
    Q[5] = A[2_000_001]
    Q[6] = A[17]
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(0)
        idx.fill_synthetic(777, n)
        for metric in (0, 1, 2):
            rows, scores, counts, st = idx.search(Q, k, metric, with_stats=True)
            assert st.fallback_queries == 0 and st.bytes_scanned == st.rows_scanned * d * 4
            for qi in list(range(0, nq, 9)) + [5, 6]:
                er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                assert counts[qi] == k and np.array_equal(rows[qi], er) and np.all(scores[qi] == es), (metric, qi)
        keep = np.random.default_rng(3).random(n) < 0.3
        mask = oc.mask_from_bool(keep)
        rows, scores, counts = idx.search(Q, k, 0, mask=mask)
        for qi in (0, 5, 6, 50, 95):
            er, es = oc.search(A, Q[qi], k, 0, mask=mask, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es), qi


@pytest.mark.parametrize("n,d,nq,k", [(9000, 300, 64, 20), (7000, 200, 33, 10), (12000, 100, 128, 10), (5000, 96, 40, 5),
                                      (3000, 896, 40, 10), (2000, 2560, 33, 5)])
def test_wide_rows_flag_pads_to_the_matrix_core_stride(n, d, nq, k):
    # NMN_INDEX_WIDE_ROWS: 300 -> 384, 200 -> 256, 100 -> 128, 96 -> 128, 896 -> 1024, 2560 -> 3072; same answers,
    # batches on the matrix cores
    from neumann_amd import GpuFlatIndex
    A = oc.synth(3000 + n + d, 0, n, d)
    Q = oc.synth(4000 + nq, 0, nq, d)
    Q[1] = A[n // 2]
    with GpuFlatIndex(d, n) as plain, GpuFlatIndex(d, n, wide_rows=True) as wide:
        built_for = [128 * c for c in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32)]
        assert plain.row_stride == (d + 7) // 8 * 8 and wide.row_stride == min(x for x in built_for if x >= d)
        wide.upload(A[: n // 2])
        wide.upload(A[n // 2:], row0=n // 2)     # appended rows keep the padding zero
        for metric in (0, 1, 2):
            check_batch(wide, A, Q, k, metric)
            check_batch(wide, A, Q[:1], k, metric)
        rng = np.random.default_rng(d)
        check_batch(wide, A, Q, k, 0, mask=oc.mask_from_bool(rng.random(n) < 0.4))


def test_mfma_batch_with_planted_near_ties_and_duplicates():
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(5)
    n, d, nq, k = 12000, 768, 16, 20
    A = (rng.standard_normal((n, d)) * 0.05).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    for qi in range(4):                          # near-copies of 4 queries, last-ulp perturbations
        for j in range(30):
            v = Q[qi].copy()
            pos = rng.integers(0, d, 4)
            v[pos] = np.nextafter(v[pos], np.float32(np.inf if j % 2 else -np.inf))
            A[rng.integers(0, n)] = v
    A[5000:5010] = A[4999]                       # exact duplicates
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            check_batch(idx, A, Q, k, metric)


def test_euclidean_batches_on_rows_the_matrix_core_sweep_cannot_take():
    from neumann_amd import GpuFlatIndex
    n, d, nq, k = 15000, 200, 11, 30             # 200 is not a multiple of 128: four queries per VALU sweep
    A = oc.synth(77, 0, n, d)
    Q = oc.synth(78, 0, nq, d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(77, n)
        check_batch(idx, A, Q, k, 1)


def test_euclidean_matrix_core_sweep_near_the_query():
    """The cancellation regime of |q|^2 + |v|^2 - 2 q.v: rows a hair away from the queries (distances ~1e-3 against
    norms ~28), exact copies (distance 0), a zero query and rows of very different norms in one shard."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(8)
    n, d, nq, k = 30000, 768, 12, 40
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[:3000] *= 0.01
    A[3000:6000] *= 30.0
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[0] = 0.0
    for qi in range(1, 6):
        for j in range(25):
            A[rng.integers(6000, n)] = Q[qi] + (1e-3 * (j + 1)) * rng.standard_normal(d).astype(np.float32) / np.float32(np.sqrt(d))
        A[rng.integers(6000, n)] = Q[qi]
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        check_batch(idx, A, Q, k, 1)
        check_batch(idx, A, Q, k, 1, mask=oc.mask_from_bool(rng.random(n) < 0.5))


def test_batch_matches_single_query_calls():
    """A batch is exactly the concatenation of single-query searches (rows, scores, counts)."""
    from neumann_amd import GpuFlatIndex
    n, d, nq, k = 50000, 768, 64, 100
    Q = oc.synth(91, 0, nq, d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(90, n)
        br, bs, bc = idx.search(Q, k, 0)
        for qi in range(0, nq, 7):
            r, s, c = idx.search(Q[qi], k, 0)
            assert np.array_equal(br[qi], r[0]) and np.array_equal(bs[qi], s[0]) and bc[qi] == c[0]


def test_mfma_sampling_pass_large_shard():
    """Shards with >= 32768 tiles run the sampling pre-pass and suppress the score writes of hopeless tiles;
    results must not change (oracle on the full 2.2M x 128 corpus), with and without a mask."""
    from neumann_amd import GpuFlatIndex
    n, d, nq, k = 2_200_000, 128, 12, 25
    A = oc.synth(4242, 0, n, d, nthreads=8)
    Q = oc.synth(4243, 0, nq, d)
    Q[3] = A[1_234_567]
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(4242, n)
        for metric in (0, 1, 2):
            rows, scores, counts, st = idx.search(Q, k, metric, with_stats=True)
            assert st.fallback_queries == 0
            for qi in range(nq):
                er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                assert counts[qi] == k and np.array_equal(rows[qi], er) and np.all(scores[qi] == es), (metric, qi)
        keep = np.random.default_rng(1).random(n) < 0.2
        mask = oc.mask_from_bool(keep)
        rows, scores, counts = idx.search(Q[:6], k, 0, mask=mask)
        for qi in range(6):
            er, es = oc.search(A, Q[qi], k, 0, mask=mask, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es)


def test_more_than_64_queries_on_a_large_shard_two_launch_sweep():
    """More than 64 queries on a shard with a sampling pass: the main sweep runs as two launches over workgroup ranges and
    the score-store bound is tightened from the first round's own tile maxima in between (nmn_api.hip).  Whatever the bound
    suppresses must never be a candidate: oracle on the full 2.2M x 128 corpus, all three metrics, with and without a mask."""
    from neumann_amd import GpuFlatIndex
    n, d, nq, k = 2_200_000, 128, 96, 25
    A = oc.synth(777, 0, n, d, nthreads=8)
    Q = oc.synth(778, 0, nq, d)
    Q[5] = A[2_000_001]          # a query whose best match sits in the LAST quarter of the shard (the second launch)
    Q[6] = A[17]                 # ... and one in the first
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(777, n)
        for metric in (0, 1, 2):
            rows, scores, counts, st = idx.search(Q, k, metric, with_stats=True)
            assert st.fallback_queries == 0
            for qi in list(range(0, nq, 9)) + [5, 6]:
                er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                assert counts[qi] == k and np.array_equal(rows[qi], er) and np.all(scores[qi] == es), (metric, qi)
        assert rows[5][0] == 2_000_001 or scores[5][0] >= scores[5][1]
        keep = np.random.default_rng(3).random(n) < 0.3
        mask = oc.mask_from_bool(keep)
        rows, scores, counts = idx.search(Q, k, 0, mask=mask)
        for qi in (0, 5, 6, 50, 95):
            er, es = oc.search(A, Q[qi], k, 0, mask=mask, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi], er) and np.all(scores[qi] == es), qi


@pytest.mark.parametrize("d,metric", [(256, 0), (128, 1), (256, 2)])
def test_one_launch_sweep_with_the_running_bound_matches_oracle(d, metric):
    """Shards of >= 32 768 tiles sweep a batch in ONE launch (round 6): the bound that gates the score stores rises inside the sweep —
    every workgroup publishes its running maximum, waves re-read the maxima and publish the k-th largest (ScanParams::run_*,
    nmn_scan_mfma_kernel.h).  All three streamed matrices (f32 rows, bf16 mirror, 8-bit mirror — cosine batches there on ONE query
    plane), every list against the oracle; planted near-copies of two queries sit in the LAST tiles, where the bound is tightest."""
    from neumann_amd import GpuFlatIndex
    n, k, nq = 2_150_000, 10, 8          # 33 594 tiles
    A = oc.synth(7700 + d, 0, n, d, nthreads=8)
    Q = oc.synth(7800 + d, 0, nq, d)
    rng = np.random.default_rng(d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(7700 + d, n)
        for j in range(6):
            row = n - 1 - 97 * j
            v = (Q[j % 2] * np.float32(1.0 + 0.1 * j) + rng.standard_normal(d).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
            idx.set_row(row, v)
            A[row] = v
        want = [oc.search(A, Q[i], k, metric, nthreads=8, partial=True, native=True) for i in range(nq)]
        for mode, nbytes in ((0, 4), (2, 2), (1, 1)):
            if mode == 1 and d % 256:
                continue                  # (the 8-bit matrix-core sweep takes strides that are multiples of 256)
            idx.set_mirror(mode)
            rows, scores, counts, st = idx.search(Q, k, metric, with_stats=True)
            assert st.bytes_scanned == n * d * nbytes and st.sweep.startswith("mfma_"), (mode, st.sweep, st.bytes_scanned)
            if not os.environ.get("NMN_NO_RUN_BOUND"):  # (the A/B switch brings the sampling pass and its launches back)
                assert st.sweep_launches == 1, (mode, st.sweep_launches)
            assert st.fallback_queries == 0
            for i in range(nq):
                er, es = want[i]
                assert counts[i] == er.size and np.array_equal(rows[i, :er.size], er) and np.all(scores[i, :er.size] == es), (mode, i)
