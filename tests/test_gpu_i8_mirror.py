"""The 8-bit mirror (neumann_amd/csrc/nmn_scan_i8.hip): sweeps of 1-2 queries read int8 codes with a scale per row — one
byte per corpus element — and the answer must still be the reference's, bit for bit: rows identical to the oracle's
(vector_engine/src/lib.rs:1950-2101 restated in oracle/nmn_oracle.c), scores equal as u32 bits.  What makes that true is
the measured quantization error in the candidate margin plus the exact rescore; what these tests attack is exactly that:
every supported row length and metric, bitmaps dense and sparse, near-ties and duplicates around the cut, rows the
quantization serves badly (one huge element: the margin becomes useless and the sweep must notice), non-finite rows,
overwrites and appends after the mirror exists."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def check(idx, A, Q, k, metric, mask=None, expect_bytes=None):
    Q = np.atleast_2d(Q)
    rows, scores, counts, stats = idx.search(Q, k, metric, mask=mask, with_stats=True)
    for qi in range(Q.shape[0]):
        er, es = oc.search(A, Q[qi], k, metric, mask=mask, nthreads=8, partial=True, native=True)
        c = er.size
        assert counts[qi] == c, (qi, counts[qi], c)
        assert np.array_equal(rows[qi, :c], er), (metric, qi, rows[qi, :8], er[:8])
        assert np.array_equal(scores[qi, :c].view(np.uint32), es.view(np.uint32)), (metric, qi)
        assert np.all(rows[qi, c:] == U64_MAX) and np.all(np.isneginf(scores[qi, c:]))
    if expect_bytes is not None:
        assert stats.bytes_scanned == stats.rows_scanned * A.shape[1] * expect_bytes, (stats.bytes_scanned, stats.rows_scanned)
    return stats


@pytest.mark.parametrize("d", [256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096,
                               128, 384, 640, 896, 1152, 2176, 3968])   # ... and rows of an odd number of 128-element halves
def test_every_supported_row_length_matches_the_oracle(d):
    from neumann_amd import GpuFlatIndex
    n = 70_000 if d <= 1536 else 20_000
    A = oc.synth(0x18 + d, 0, n, d, nthreads=8)
    Q = oc.synth(0x19 + d, 0, 2, d)
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            check(idx, A, Q[0], 100, metric, expect_bytes=1)       # one query
            check(idx, A, Q, 37, metric, expect_bytes=1)           # two per sweep
        keep = np.random.default_rng(d).random(n) < 0.5
        check(idx, A, Q[0], 100, 0, mask=oc.mask_from_bool(keep), expect_bytes=1)
        keep = np.random.default_rng(d + 1).random(n) < 0.04       # sparse tiles: compacted steps
        check(idx, A, Q, 50, 1, mask=oc.mask_from_bool(keep), expect_bytes=1)


def test_mirror_modes_read_1_2_and_4_bytes_per_element_and_agree():
    from neumann_amd import GpuFlatIndex
    n, d, k = 150_000, 768, 100
    A = oc.synth(0x21, 0, n, d, nthreads=8)
    q = oc.synth(0x22, 0, 1, d)[0]
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x21, n)
        got = {}
        for mode, nbytes in ((1, 1), (2, 2), (0, 4), (1, 1)):
            idx.set_mirror(mode)
            st = check(idx, A, q, k, 0, expect_bytes=nbytes)
            got[mode] = idx.search(q, k, 0)
        for mode in (2, 0):
            assert np.array_equal(got[mode][0], got[1][0]) and np.array_equal(got[mode][1].view(np.uint32), got[1][1].view(np.uint32))
        # a 3-query batch is the matrix-core sweep's — over the 8-bit mirror too, with or without a bitmap (round 3);
        # a row length outside the 256-element groups is the bf16 VALU sweep's
        idx.set_mirror(1)
        Q3 = oc.synth(0x23, 0, 3, d)
        check(idx, A, Q3, k, 0, expect_bytes=1)
        keep = np.random.default_rng(3).random(n) < 0.5
        check(idx, A, Q3, k, 0, mask=oc.mask_from_bool(keep), expect_bytes=1)
        for metric in (1, 2):
            check(idx, A, Q3, 17, metric, mask=oc.mask_from_bool(keep), expect_bytes=1)
        idx.set_mirror(2)
        check(idx, A, Q3, k, 0, expect_bytes=2)
    with GpuFlatIndex(320, 80_000, single_launch=False) as idx:
        B = oc.synth(0x24, 0, 80_000, 320, nthreads=8)
        idx.upload(B)
        check(idx, B, oc.synth(0x25, 0, 1, 320)[0], 10, 0, expect_bytes=2)


def test_planted_near_ties_and_duplicates_around_the_cut():
    """rows within a few ulps of each other around rank k, and exact copies straddling it: the order must be the oracle's
    (score descending, row ascending) although the 8-bit sweep cannot tell any of them apart"""
    from neumann_amd import GpuFlatIndex
    n, d, k = 120_000, 768, 64
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    near = np.flatnonzero(rng.random(n) < 0.001)[:90]
    for t, r in enumerate(near):                       # 90 rows close to the query, differing in the last bits
        A[r] = q * np.float32(1.0 + 1e-7 * (t % 5)) + np.float32(1e-4) * rng.standard_normal(d).astype(np.float32) * np.float32(t % 3 == 0)
    A[near[10]] = A[near[3]]
    A[near[40]] = A[near[3]]                           # exact copies: ties by row id
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for metric in (0, 1, 2):
            for kk in (k, 89, 90, 91, 500):
                check(idx, A, q, kk, metric, expect_bytes=1)


def test_rows_the_quantization_serves_badly():
    """Every row carries one element 1000x the others: the per-row scale is set by it, the other 255 elements round to 0 or
    +-1 and the measured relative error is several per cent — the margin admits ~10 000 rows, far beyond cand_cap.  The
    crowd list takes them (a crowd, not a useless margin: well under an eighth of the shard); the answer is the oracle's."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 300_000, 256, 20
    rng = np.random.default_rng(9)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[np.arange(n), rng.integers(0, d, n)] *= np.float32(1000.0)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for i in range(6):
            for metric in (0, 1, 2):
                check(idx, A, Q[i], k, metric)


def test_the_8_bit_mirror_switches_itself_off_when_its_margin_is_useless():
    """All rows point the same way (a common vector plus 5 % noise): cosine scores differ in the third decimal.  The bf16
    mirror's measured error (~0.0015) still separates them; the 8-bit mirror's (~0.01) puts EVERY row within the margin —
    not a crowd but a useless margin: each query is re-swept in f32 (and is still exact).  After enough of those the shard
    leaves the 8-bit mirror alone: its sweeps read 2 bytes per element again."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 300_000, 256, 20
    rng = np.random.default_rng(10)
    c = rng.standard_normal(d).astype(np.float32)
    A = (c[None, :] + np.float32(0.05) * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        st = check(idx, A, Q[0], k, 0, expect_bytes=1)
        for i in range(6):
            check(idx, A, Q[i], k, 0)
            check(idx, A, Q[i], k, 2)
        for i in range(300):                           # the switch looks at its counters every 256th call
            idx.search(Q[i % 6], k, 0)
        st = check(idx, A, Q[0], k, 0)
        assert st.bytes_scanned == st.rows_scanned * d * 2, "the shard should have left the 8-bit mirror alone by now"


def test_a_shard_leaves_the_8_bit_mirror_when_its_data_drifts_and_returns_when_it_drifts_back():
    """The switch under drift.  A healthy shard (isotropic rows) is overwritten, piece by piece, with rows that
    all point the same way — the 8-bit margin then covers every row, each query pays the 8-bit pass AND an f32 retry, and
    within a few hundred searches the shard must go back to the bf16 mirror (built on demand at that moment: a shard keeps one
    mirror until it needs the other).  The rows are then overwritten with healthy data again: the off period (8192 searches)
    runs out, the shard re-enters the 8-bit mirror — which was kept current through every overwrite while nobody read it — and
    stays there.  Every answer on the way is the oracle's, whichever mirror served it; the resident bytes are checked too."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 300_000, 256, 20
    rng = np.random.default_rng(77)
    H = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal(d).astype(np.float32)
    U = (c[None, :] + np.float32(0.05) * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    H2 = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(H)
        elems = idx.hbm_bytes()[0] // 4
        assert idx.hbm_bytes()[1] == elems + 12 * (idx.hbm_bytes()[2] // 8), "one mirror while the rows arrive: the 8-bit one"
        for i in range(6):
            check(idx, H, Q[i], k, i % 3, expect_bytes=1)
        # drift in: four uploads of 75 000 rows each, searches in between (a mixed shard is still answered exactly)
        A = H.copy()
        for part in range(4):
            lo, hi = part * 75_000, (part + 1) * 75_000
            A[lo:hi] = U[lo:hi]
            idx.upload(U[lo:hi], row0=lo)
            check(idx, A, Q[part], k, 0)
        for i in range(600):                           # the switch looks at its counters every 256th call
            idx.search(Q[i % 6], k, 0)
        st = check(idx, U, Q[0], k, 0)
        assert st.bytes_scanned == st.rows_scanned * d * 2, "the shard should have left the 8-bit mirror by now"
        assert idx.hbm_bytes()[1] == 3 * elems + 12 * (idx.hbm_bytes()[2] // 8), "... and built the bf16 mirror when it did"
        for i in range(6):
            check(idx, U, Q[i], k, i % 3)
        # drift back: healthy rows again, written while the 8-bit mirror is switched off (it is re-quantized in place all the same)
        for part in range(4):
            lo, hi = part * 75_000, (part + 1) * 75_000
            A[lo:hi] = H2[lo:hi]
            idx.upload(H2[lo:hi], row0=lo)
            check(idx, A, Q[part], k, 1)
        st = check(idx, H2, Q[0], k, 0)
        assert st.bytes_scanned == st.rows_scanned * d * 2, "still inside the off period"
        for i in range(8192 + 256):                    # the off period is counted in searches
            idx.search(Q[i % 6], k, 0)
        for i in range(6):
            check(idx, H2, Q[i], k, i % 3, expect_bytes=1)   # back on the 8-bit mirror, and exact
        for i in range(600):                           # ... and it stays: healthy data does not trip the switch
            idx.search(Q[i % 6], k, 0)
        check(idx, H2, Q[0], k, 0, expect_bytes=1)


def test_non_finite_rows_and_queries():
    from neumann_amd import GpuFlatIndex
    n, d, k = 90_000, 512, 30
    rng = np.random.default_rng(13)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[1234, 7] = np.inf
    A[40_000, 100] = -np.inf
    q = np.abs(rng.standard_normal(d)).astype(np.float32)
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.upload(A)
        rows, scores, counts = idx.search(q, k, 2)     # dot product: row 1234 scores +inf and must come first
        er, es = oc.search(A, q, k, 2)
        assert np.array_equal(rows[0], er) and np.array_equal(scores[0].view(np.uint32), es.view(np.uint32))
    A = rng.standard_normal((n, d)).astype(np.float32)
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.upload(A)
        qbad = q.copy()
        qbad[3] = np.inf
        rows, scores, counts = idx.search(qbad, k, 1)  # Euclidean: every distance is inf, score 0 — ties by row id
        er, es = oc.search(A, qbad, k, 1)
        assert np.array_equal(rows[0], er) and np.array_equal(scores[0].view(np.uint32), es.view(np.uint32))
        check(idx, A, q, k, 0, expect_bytes=1)


def test_overwrites_and_appends_after_the_mirror_exists():
    from neumann_amd import GpuFlatIndex
    n, d, k = 100_000, 768, 40
    rng = np.random.default_rng(21)
    A = rng.standard_normal((n + 5000, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with GpuFlatIndex(d, n + 5000, single_launch=False) as idx:
        idx.upload(A[:n])
        check(idx, A[:n], q, k, 0, expect_bytes=1)                 # builds the 8-bit mirror
        A[777] = q * np.float32(3.0)                               # set_row: a new best row
        idx.set_row(777, A[777])
        A[5000:5200] = rng.standard_normal((200, d)).astype(np.float32) + q * np.float32(0.5)
        idx.upload(A[5000:5200], row0=5000)                        # overwrite a run of rows
        for metric in (0, 1, 2):
            check(idx, A[:n], q, k, metric, expect_bytes=1)
        A[n:n + 5000] = rng.standard_normal((5000, d)).astype(np.float32) + q * np.float32(0.2)
        idx.upload(A[n:n + 5000], row0=n)                          # append: the mirror is extended at the next search
        for metric in (0, 1, 2):
            check(idx, A, q, k, metric, expect_bytes=1)


def test_k_1000_euclidean_with_masks_like_config_5():
    """config 5's shape at 1/50 scale: 1536-element rows, L2, TOP-1000, bitmaps of selectivity 1.0 / 0.5 / 0.1"""
    from neumann_amd import GpuFlatIndex
    n, d, k = 200_000, 1536, 1000
    A = oc.synth(0x5EED0005, 0, n, d, nthreads=8)
    q = oc.synth(0x5EED0002, 0, 1, d)[0]
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x5EED0005, n)
        check(idx, A, q, k, 1, expect_bytes=1)
        for sel in (0.5, 0.1, 0.01):
            keep = np.random.default_rng(int(sel * 100)).random(n) < sel
            check(idx, A, q, k, 1, mask=oc.mask_from_bool(keep), expect_bytes=1)


@pytest.mark.parametrize("d", [256, 512, 768, 1024, 1280, 1536])
def test_batches_on_the_matrix_cores_over_the_8_bit_mirror(d):
    """3 .. 128 queries per sweep: v_mfma_i32_16x16x64_i8 over the int8 codes, the queries as two int8 planes.  Every metric,
    batch sizes on both sides of the 64-query block (two query blocks folded into one launch), k up to 1000."""
    from neumann_amd import GpuFlatIndex
    n = 150_000 if d <= 768 else 80_000
    A = oc.synth(0x31 + d, 0, n, d, nthreads=8)
    Q = oc.synth(0x32 + d, 0, 100, d)
    Q[5] = A[777] * np.float32(1.5)                 # a query proportional to a stored row
    Q[6] = A[778] + np.float32(1e-3)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x31 + d, n)
        for metric in (0, 1, 2):
            check(idx, A, Q[:5 if d < 768 else 3], 50, metric, expect_bytes=1)
            check(idx, A, Q[:64], 100, metric, expect_bytes=1)
        check(idx, A, Q, 100, 0, expect_bytes=1)    # 100 queries: two blocks of 64 in one launch
        check(idx, A, Q[:17], 1000, 1, expect_bytes=1)


def test_large_shard_batch_with_sampling_pass_and_planted_ties():
    """1.2M rows: the sampling pass and the score-write suppression are active (n_sample >= 1024 tiles); near-duplicates of the
    queries planted around the cut; cosine and Euclidean."""
    from neumann_amd import GpuFlatIndex
    n, d, k, nq = 1_200_000, 768, 100, 64
    A = oc.synth(0x41, 0, n, d, nthreads=8)
    Q = oc.synth(0x42, 0, nq, d)
    rng = np.random.default_rng(4)
    plant = rng.choice(n, 64, replace=False)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x41, n)
        for t, r in enumerate(plant):
            A[r] = Q[t % 8] * np.float32(1.0 + 1e-6 * t) + np.float32(2e-4) * rng.standard_normal(d).astype(np.float32)
            idx.set_row(int(r), A[r])
        A[plant[9]] = A[plant[1]]
        idx.set_row(int(plant[9]), A[plant[9]])
        for metric in (0, 1):
            rows, scores, counts, st = idx.search(Q, k, metric, with_stats=True)
            assert st.bytes_scanned == st.rows_scanned * d
            for qi in (0, 1, 7, 8, 33, 63):
                er, es = oc.search(A, Q[qi], k, metric, nthreads=8, partial=True, native=True)
                assert np.array_equal(rows[qi], er) and np.array_equal(scores[qi].view(np.uint32), es.view(np.uint32)), (metric, qi)


def test_euclidean_estimator_follows_the_data():
    """qprep picks the 8-bit sweep's Euclidean estimator per query from the shard's previous threshold distance: between the
    stored representations (error bounded in distance space: tight for near neighbours) or |q|^2 + |v|^2 - 2 q~.v~ with exact
    magnitudes (bounded in squared-distance space: tighter when the k-th neighbour is farther than |q|, i.e. on uncorrelated
    rows).  Either way the answer is the oracle's; on uncorrelated rows the second query must need fewer candidates than
    the first (which had no history), on clustered rows the choice must not cost exactness."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 300_000, 768, 200
    A = oc.synth(0x51, 0, n, d, nthreads=8)
    Q = oc.synth(0x52, 0, 4, d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x51, n)
        # counted through the asynchronous API (one pass per query, the whole chain): a host-buffer search whose candidate list
        # overflows is followed up by a second pass (short chain, round 4) — and that pass already has the first one's history
        import torch
        qd = torch.from_numpy(Q).cuda()
        c = []
        for i in range(4):
            idx.search_device(qd[i:i + 1], k, 1)
            c.append(idx.last_stats().candidates_rescored)
        assert c[1] < c[0] and c[2] < c[0] and c[3] < c[0], c
        for i in range(4):
            check(idx, A, Q[i], k, 1, expect_bytes=1)
        keep = np.random.default_rng(2).random(n) < 0.2
        check(idx, A, Q[0], k, 1, mask=oc.mask_from_bool(keep), expect_bytes=1)
        check(idx, A, Q[:2], k, 1, expect_bytes=1)      # two queries per sweep
    rng = np.random.default_rng(8)
    centres = (rng.standard_normal((64, d)) * 3.0).astype(np.float32)
    B = (centres[rng.integers(0, 64, n)] + np.float32(0.05) * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(B)
        for i in range(4):
            q = centres[i] + np.float32(0.02) * rng.standard_normal(d).astype(np.float32)
            check(idx, B, q, k, 1, expect_bytes=1)
        check(idx, B, Q[0], k, 1, expect_bytes=1)       # a far query after near ones, and back
        check(idx, B, centres[9], k, 1, expect_bytes=1)


# ---- the survivor walk of masked sweeps (round 3) -----------------------------------------------------------------------
_WALK_CHILD = r'''
import sys, numpy as np
from oracle import oracle_c as oc
from neumann_amd import GpuFlatIndex
n, d = int(sys.argv[1]), int(sys.argv[2])
A = oc.synth(0x51 + d, 0, n, d, nthreads=8)
A[1000:1040] = A[999]                                   # copies inside and across tiles
Q = oc.synth(0x52 + d, 0, 2, d)
Q[1] = A[999] * np.float32(1.25)
rng = np.random.default_rng(d)
tiles = (n + 63) // 64
masks = {}
masks["random 0.3"] = rng.random(n) < 0.3
masks["random 0.02"] = rng.random(n) < 0.02
masks["random 0.002"] = rng.random(n) < 0.002
runs = np.zeros(tiles * 64, bool).reshape(tiles, 64)
runs[(np.arange(tiles) % 40) < 10] = True              # ten full tiles, thirty empty ones
masks["runs of full tiles"] = runs.reshape(-1)[:n].copy()
masks["all"] = np.ones(n, bool)
tail = np.zeros(n, bool); tail[-37:] = True; tail[5] = True
masks["tail rows and one more"] = tail
mixed = rng.random(n) < 0.01
mixed[: n // 3] = rng.random(n // 3) < 0.9             # a third nearly full, the rest sparse
masks["dense third, sparse rest"] = mixed
masks["none"] = np.zeros(n, bool)
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)
with GpuFlatIndex(d, n, single_launch=False) as idx:
    idx.upload(A)
    for name, keep in masks.items():
        m = oc.mask_from_bool(keep)
        for metric, k, QQ in ((0, 100, Q[:1]), (1, 40, Q), (2, 7, Q[1:])):
            rows, scores, counts, stats = idx.search(QQ, k, metric, mask=m, with_stats=True)
            assert stats.bytes_scanned == stats.rows_scanned * d, (name, "not the 8-bit sweep")
            for qi in range(QQ.shape[0]):
                er, es = oc.search(A, QQ[qi], k, metric, mask=m, nthreads=8, partial=True, native=True)
                c = er.size
                assert counts[qi] == c, (name, metric, qi, counts[qi], c)
                assert np.array_equal(rows[qi, :c], er), (name, metric, qi, rows[qi, :8], er[:8])
                assert np.array_equal(scores[qi, :c].view(np.uint32), es.view(np.uint32)), (name, metric, qi)
                assert np.all(rows[qi, c:] == U64_MAX)
print("WALK-OK")
'''


@pytest.mark.parametrize("n,d,waves", [(300_000, 256, "256"), (300_000, 1536, "256"), (200_000, 2048, "256"), (1_200_000, 256, "256"),
                                        (300_000, 128, "256"), (300_000, 128, ""), (300_000, 384, "256"),   # eight-lane steps (128), the guarded last group (384)
                                        (300_000, 768, ""), (300_000, 768, "nowalk")])
def test_survivor_walk_of_masked_sweeps_matches_the_oracle(n, d, waves):
    """The masked 8-bit sweep lists the participating rows of up to 64 tiles of a wave and reads them four per step across
    tile borders (nmn_scan_i8.hip).  With NMN_SCAN_WAVES=256 a wave owns 19-74 tiles, so the lists are cut into sub-ranges
    of at most 512 rows and a wave walks more than one 64-tile range — the paths a 300k-row shard under the default 4096
    waves (2 tiles per wave) never reaches.  Masks: random dense / sparse / very sparse, runs of full and empty tiles,
    everything, nothing, the ragged tail, a dense third; rows and score bits must be the oracle's."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if waves == "nowalk":
        env["NMN_NO_WALK"] = "1"
    elif waves:
        env["NMN_SCAN_WAVES"] = waves
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _WALK_CHILD, str(n), str(d)], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WALK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("d", [256, 768, 1280])   # 16-KiB stages: a tile is 1 / 3 / 5 stages of a ring of eight
def test_bitmap_batches_on_a_small_shard_one_tile_per_workgroup(d):
    """64-query batches under a bitmap on the 8-bit mirror where a workgroup's range (one tile) is shorter than the DMA ring:
    the ring's prologue must still put kRing - 1 stages in flight, or the loop's counted wait says nothing about stage 0
    (round 3: 47 of 64 answers lost a row at 20 000 x 768 under a Euclidean metric until the prologue issued its dummies)."""
    from neumann_amd import GpuFlatIndex
    n, nq = 20_000, 64
    A = oc.synth(0x61 + d, 0, n, d, nthreads=8)
    Q = oc.synth(0x62 + d, 0, nq, d)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for sel in (1.0, 0.9, 0.3):
            keep = np.random.default_rng(d).random(n) < sel
            for metric, k in ((1, 100), (0, 17), (2, 40)):
                check(idx, A, Q, k, metric, mask=oc.mask_from_bool(keep), expect_bytes=1)
