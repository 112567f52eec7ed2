"""nmn_search_stats.sweep_kind: the library REPORTS which kernel streamed the rows for a search (include/neumann_gpu.h NMN_SWEEP_*)
— bench.py prints that instead of re-deriving search_enqueue()'s dispatch.  One search on each of the eight
paths, the reported kind and the bytes per element that go with it asserted, the answers checked against the oracle as always."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu


def _same(rows, scores, counts, A, Q, k, metric, mask=None):
    for qi in range(Q.shape[0]):
        er, es = oc.search(A, Q[qi], k, metric, mask=mask, nthreads=8, partial=True, native=True)
        assert counts[qi] == er.size and np.array_equal(rows[qi, :er.size], er) and np.all(scores[qi, :er.size] == es), qi


def test_every_sweep_reports_its_kind():
    from neumann_amd import GpuFlatIndex, _capi
    n, d, k = 270_000, 256, 20           # >= 4096 tiles (ring), stride a multiple of 256 (8-bit matrix-core sweep)
    A = oc.synth(4242, 0, n, d, nthreads=8)
    Q = oc.synth(4243, 0, 8, d)
    keep = np.random.default_rng(3).random(n) < 0.4
    mask = oc.mask_from_bool(keep)
    seen = {}
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.fill_synthetic(4242, n)
        cases = [  # (mirror mode, queries, bitmap) -> (kind, bytes per element)
            (0, Q[:1], None, "ring_f32", 4), (0, Q[:1], mask, "valu_f32", 4), (0, Q[:2], None, "valu_f32", 4),
            (2, Q[:1], None, "valu_bf16", 2), (1, Q[:1], None, "valu_i8", 1),
            (0, Q, None, "mfma_f32", 4), (2, Q, None, "mfma_bf16", 2), (1, Q, None, "mfma_i8", 1)]
        for mode, qq, m, kind, eb in cases:
            idx.set_mirror(mode)
            rows, scores, counts, st = idx.search(qq, k, 0, mask=m, with_stats=True)
            _same(rows, scores, counts, A, qq, k, 0, mask=m)
            assert st.sweep == kind, (mode, qq.shape[0], m is not None, st.sweep, kind)
            rows_read = int(keep.sum()) if m is not None else n
            assert st.rows_scanned == rows_read and st.bytes_scanned == rows_read * d * eb
            assert st.sweep_launches >= 1 and st.fallback_queries == 0
            seen[st.sweep_kind] = kind
        # k above NMN_MAX_TOP_K: the large-k path — exact scores of every row, no approximate sweep
        idx.set_mirror(1)
        rows, scores, counts, st = idx.search(Q[:1], 5000, 0, with_stats=True)
        _same(rows, scores, counts, A, Q[:1], 5000, 0)
        assert st.sweep == "exact" and st.sweep_kind == _capi.SWEEP_EXACT and st.bytes_scanned == n * d * 4
    assert seen == {_capi.SWEEP_RING_F32: "ring_f32", _capi.SWEEP_VALU_F32: "valu_f32", _capi.SWEEP_VALU_BF16: "valu_bf16",
                    _capi.SWEEP_VALU_I8: "valu_i8", _capi.SWEEP_MFMA_F32: "mfma_f32", _capi.SWEEP_MFMA_BF16: "mfma_bf16",
                    _capi.SWEEP_MFMA_I8: "mfma_i8"}
    # a small shard, one query: tiny_search_kernel (one launch, exact scores) reports "exact" too; an empty shard "none"
    with GpuFlatIndex(d, 5000) as small:
        small.fill_synthetic(4242, 5000)
        rows, scores, counts, st = small.search(Q[:1], k, 0, with_stats=True)
        _same(rows, scores, counts, A[:5000], Q[:1], k, 0)
        assert st.sweep == "exact" and st.sweep_launches == 1
    with GpuFlatIndex(d, 16) as empty:
        rows, scores, counts, st = empty.search(Q[:1], k, 0, with_stats=True)
        assert counts[0] == 0 and st.sweep == "none" and st.rows_scanned == 0


def test_device_api_last_stats_reports_the_kind_too():
    import torch
    from neumann_amd import GpuFlatIndex
    n, d, k = 270_000, 128, 10
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(77, n)
        idx.set_mirror(0)
        q = torch.from_numpy(oc.synth(78, 0, 1, d)).cuda()
        idx.search_device(q, k, 0)
        torch.cuda.synchronize()
        st = idx.last_stats(None)
        assert st.sweep == "ring_f32" and st.sweep_launches == 1
        # the read ceiling of that sweep's own data movement: the ring with nothing behind it
        gbps = idx.read_probe(2)
        assert gbps > 500.0


@pytest.mark.parametrize("mode", [1, 0])
def test_split_selection_on_a_large_shard_matches_oracle(mode):
    """Lone calls on shards of more than 16 384 tiles select with SEVERAL workgroups per query (round 6, SelectParams::split): each
    holds a part of the tile maxima, the parts meet once on the super-group maxima and pick the same bound, every part appends its
    candidates.  1.7M rows x 128 = 26 563 tiles = two parts: one and two queries per call, three metrics, a bitmap, planted
    near-copies of the query at both ends of the shard and across the parts' border, duplicates (ties by row id across parts)."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 1_700_000, 128, 50
    A = oc.synth(5150, 0, n, d, nthreads=8)
    Q = oc.synth(5151, 0, 3, d)
    rng = np.random.default_rng(11)
    per = ((n // 64 + 1 + 1) // 2 + 3) // 4 * 4 * 64   # rows of the first part (tiles per part rounded up to a multiple of four)
    with GpuFlatIndex(d, n) as idx:
        idx.set_mirror(mode)
        idx.fill_synthetic(5150, n)
        for j, row in enumerate((0, 63, per - 1, per, per + 64, n - 1, n // 3, n // 3 + 1)):
            v = (Q[0] * np.float32(1.0 + 0.05 * (j % 3))).astype(np.float32)
            if j >= 6:
                v = (Q[0] * np.float32(2.0)).astype(np.float32)      # two identical rows: the tie is broken by the row id
            else:
                v[rng.integers(0, d, size=2)] += np.float32(1e-3 * j)
            idx.set_row(row, v)
            A[row] = v
        keep = rng.random(n) < 0.2
        keep[[0, per - 1, per, n - 1]] = True
        for metric in (0, 1, 2):
            for qq, mask in ((Q[:1], None), (Q[:2], None), (Q[:1], oc.mask_from_bool(keep))):
                rows, scores, counts, st = idx.search(qq, k, metric, mask=mask, with_stats=True)
                assert st.fallback_queries == 0
                for i in range(qq.shape[0]):
                    er, es = oc.search(A, qq[i], k, metric, mask=mask, nthreads=8, partial=True, native=True)
                    assert counts[i] == er.size and np.array_equal(rows[i, :er.size], er) and np.all(scores[i, :er.size] == es), (mode, metric, i)
