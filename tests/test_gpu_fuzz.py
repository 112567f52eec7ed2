"""Randomised parity sweep through the C ABI: odd shapes (dimensions that are not multiples of 4 / 8 / 128, shards smaller
than a tile, k beyond the shard, duplicates, zero rows, masks of every density, 1-140 queries per call) against the
oracle.  Deterministic seeds; each case is small, the point is the number of shapes."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _case(rng):
    d = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 31, 64, 100, 120, 127, 128, 129, 200, 250, 256, 300, 384, 720, 768]))
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 500, 1000, 2500, 4097, 4097, 20000]))
    nq = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 17, 64, 65, 140]))
    k = int(rng.choice([1, 2, 5, 10, 64, 100, 333]))
    metric = int(rng.integers(0, 3))
    A = rng.standard_normal((n, d)).astype(np.float32)
    style = int(rng.integers(0, 6))
    if style == 1 and n > 4:                      # duplicates
        A[rng.integers(0, n, n // 3)] = A[0]
    elif style == 2:                              # a few zero rows
        A[rng.integers(0, n, max(1, n // 10))] = 0.0
    elif style == 3:                              # wide dynamic range
        A *= np.exp(rng.uniform(-6, 6, (n, 1))).astype(np.float32)
    elif style == 4:                              # coarse values: many exact ties
        A = np.round(A * 2).astype(np.float32) / 2
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    if rng.random() < 0.3:
        Q[0] = A[rng.integers(0, n)]
    if rng.random() < 0.15:
        Q[-1] = 0.0
    mask = None
    r = rng.random()
    if r < 0.5:
        density = float(rng.choice([0.0, 0.02, 0.3, 0.9, 1.0]))
        mask = oc.mask_from_bool(rng.random(n) < density)
    return A, Q, k, metric, mask


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("NMN_FUZZ_SEEDS", "60"))))
def test_random_shapes_match_the_oracle(seed):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(1000 + seed)
    for _ in range(14):
        A, Q, k, metric, mask = _case(rng)
        n, d = A.shape
        # every other index stores its rows with NMN_INDEX_WIDE_ROWS (stride padded to the matrix-core sweep's)
        with GpuFlatIndex(d, n + int(rng.integers(0, 70)), wide_rows=bool(rng.integers(0, 2))) as idx:
            idx.upload(A)
            rows, scores, counts = idx.search(Q, k, metric, mask=mask)
            for qi in range(Q.shape[0]):
                if metric != 1 and not Q[qi].any():
                    # the zero-query rule (`Ok([])` unless Euclidean, lib.rs:2066) lives in the facade, which the oracle's
                    # search restates; the C ABI scores the rows: every score is 0.0, so the order is by row
                    part = np.arange(A.shape[0]) if mask is None else np.nonzero(np.unpackbits(
                        mask.view(np.uint8), bitorder="little")[:A.shape[0]])[0]
                    er = part[:k].astype(np.uint64)
                    es = np.zeros(er.size, np.float32)
                else:
                    er, es = oc.search(A, Q[qi], k, metric, mask=mask)
                c = er.size
                ctx = (seed, n, d, Q.shape[0], k, metric, None if mask is None else int(np.unpackbits(mask.view(np.uint8)).sum()), qi)
                assert counts[qi] == c, ctx
                assert np.array_equal(rows[qi, :c], er), ctx
                assert np.array_equal(scores[qi, :c].view(np.uint32), es.view(np.uint32)) or np.all(scores[qi, :c] == es), ctx
                assert np.all(rows[qi, c:] == U64_MAX), ctx
