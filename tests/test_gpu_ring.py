"""The single-query sweep over the f32 rows through the LDS-DMA ring (nmn_scan_ring.hip: f32 arithmetic, rows streamed like the
matrix-core sweep's) — what an unmasked nq = 1 search takes on a shard of >= 4096 tiles when no mirror serves it (the headline
configuration of bench.py).  Rows and scores must be the oracle's for every metric and row length it is built for; shards below
the threshold, bitmaps and two-query calls stay on scan_kernel and must give the same answers."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _check(idx, A, q, k, metric, mask=None):
    rows, scores, counts, st = idx.search(q, k, metric, mask=mask, with_stats=True)
    er, es = oc.search(A, q, k, metric, mask=mask, nthreads=8, partial=True, native=True)
    c = er.size
    assert counts[0] == c, (counts[0], c)
    assert np.array_equal(rows[0, :c], er), (rows[0, :8], er[:8])
    assert np.all(scores[0, :c] == es)
    assert np.all(rows[0, c:] == U64_MAX)
    return st


@pytest.mark.parametrize("n,d,k", [(300_000, 768, 100), (270_011, 128, 10), (600_000, 256, 50), (262_144 + 77, 384, 100),
                                   (300_000, 1000, 20),   # stride padded to 1024
                                   (280_000, 1536, 1000), (262_200, 2048, 100), (263_000, 640, 7)])
def test_ring_sweep_matches_oracle(n, d, k):
    from neumann_amd import GpuFlatIndex
    A = oc.synth(9000 + n + d, 0, n, d, nthreads=8)
    Q = oc.synth(9100 + d, 0, 4, d)
    Q[1] = A[n - 3]                     # a stored row (in the ragged last tile)
    Q[2] = 0.0                          # the zero query (defined under Euclidean only)
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(0)
        idx.fill_synthetic(9000 + n + d, n)
        for metric in (0, 1, 2):
            for qi in range(4):
                if metric != 1 and qi == 2:
                    continue            # (lib.rs:2066: a zero query is the host's Ok([]) for every metric but Euclidean)
                st = _check(idx, A, Q[qi], k, metric)
                assert st.bytes_scanned == st.rows_scanned * d * 4 and st.fallback_queries == 0
        # a bitmap, and a two-query call: scan_kernel's business, same answers
        keep = np.random.default_rng(n).random(n) < 0.3
        _check(idx, A, Q[0], k, 0, mask=oc.mask_from_bool(keep))
        rows, scores, counts = idx.search(Q[:2], k, 2)
        for qi in range(2):
            er, es = oc.search(A, Q[qi], k, 2, nthreads=8, partial=True, native=True)
            assert np.array_equal(rows[qi, :er.size], er) and np.all(scores[qi, :er.size] == es)
        assert idx.hbm_bytes()[1] == 0


def test_ring_sweep_with_planted_near_ties_and_a_tail_of_zero_rows():
    """Near-copies of the query differing in the last ulp, exact duplicates, zero rows (cosine 0 by the zero-magnitude rule) and a
    shard that ends in the middle of a tile."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(17)
    n, d, k = 270_000 + 13, 768, 60
    A = (rng.standard_normal((n, d)) * 0.05).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    for j in range(40):
        v = q.copy()
        pos = rng.integers(0, d, 3)
        v[pos] = np.nextafter(v[pos], np.float32(np.inf if j % 2 else -np.inf))
        A[rng.integers(0, n)] = v
    A[100_000:100_008] = A[99_999]
    A[n - 5:] = 0.0
    A[5] = 0.0
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(0)
        idx.upload(A)
        for metric in (0, 1, 2):
            _check(idx, A, q, k, metric)


@pytest.mark.parametrize("mirror", [0, 2])
@pytest.mark.parametrize("n,d,k", [(300_000, 768, 100), (270_011, 128, 10), (262_144 + 77, 384, 50), (280_000, 1536, 1000),
                                   (300_000, 1000, 20), (1_200_000, 256, 30)])
def test_masked_sweep_over_f32_and_bf16_rows_matches_oracle(n, d, k, mirror):
    """scan_kernel under a bitmap, including its survivor walk (sparse bitmaps: the participating rows of 64 tiles listed and read four per
    step): random bitmaps of every density, runs of rows (a time range, an IVF list), a single kept row, kept rows in a few tiles only
    plus the ragged last tile, nobody, all but every 97th — one and two queries per call, every metric."""
    from neumann_amd import GpuFlatIndex
    A = oc.synth(9500 + n + d, 0, n, d, nthreads=8)
    Q = oc.synth(9600 + d, 0, 2, d)
    Q[1] = A[n - 2]
    rng = np.random.default_rng(n + d)
    masks = {}
    for s in (0.9, 0.5, 0.1, 0.01, 0.0007):
        masks["random %g" % s] = rng.random(n) < s
    runs = np.zeros(n, bool)
    for r in range(5):
        a = int(rng.integers(0, n - 9000))
        runs[a:a + int(rng.integers(100, 9000))] = True
    masks["runs"] = runs
    one = np.zeros(n, bool)
    one[n - 2] = True
    masks["one row"] = one
    few = np.zeros(n, bool)
    few[64 * 1000:64 * 1003] = True
    few[n - 70:] = True
    masks["few tiles + the ragged end"] = few
    masks["nobody"] = np.zeros(n, bool)
    allbut = np.ones(n, bool)
    allbut[::97] = False
    masks["all but every 97th"] = allbut
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(mirror)
        idx.fill_synthetic(9500 + n + d, n)
        for name, keep in masks.items():
            m = oc.mask_from_bool(keep)
            for metric in (0, 1, 2):
                for qi in range(2):
                    st = _check(idx, A, Q[qi], k, metric, mask=m)
                    assert st.fallback_queries == 0, name
            # two queries in one call (NQ = 2)
            rows, scores, counts = idx.search(Q, k, 0, mask=m)
            for qi in range(2):
                er, es = oc.search(A, Q[qi], k, 0, mask=m, nthreads=8, partial=True, native=True)
                assert counts[qi] == er.size and np.array_equal(rows[qi, :er.size], er) and np.all(scores[qi, :er.size] == es), name
