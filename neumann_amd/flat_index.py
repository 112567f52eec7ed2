"""GpuFlatIndex — host-side handle of one row-range shard resident on one MI355X.

Python mirror of the safe Rust wrapper a GPU-enabled `vector_engine` crate would put over the C ABI
(`GpuFlatIndex::search(&self, q, k, metric, mask) -> Vec<(usize, f32)>`): the
same shape `HNSWIndex::search` returns into `search_similar`'s cache hook
(vector_engine/src/lib.rs:1977-2001), i.e. (row, score) pairs that the caller maps back to keys.
All compute happens in libneumann_gpu.so; this file only marshals buffers.
"""
import ctypes as C
import enum

import numpy as np

from . import _capi


class DistanceMetric(enum.IntEnum):
    """vector_engine::DistanceMetric (lib.rs:268-289), same discriminant order."""
    Cosine = 0
    Euclidean = 1
    DotProduct = 2


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class GpuFlatIndex:
    def __init__(self, dim, capacity_rows, row_base=0, device=-1, cand_cap=0, wide_rows=False, single_launch=True):
        # wide_rows: NMN_INDEX_WIDE_ROWS — rows padded up to the next multiple of 128 elements even at up to 1/2 more
        # bytes (300 -> 384), so query batches and concurrent callers take the matrix-core sweep
        self._lib = _capi.load()
        self._h = C.c_void_p()
        # single_launch=False: NMN_INDEX_NO_SINGLE_LAUNCH — small shards go through the general pipeline too (tests, A/B)
        desc = _capi.IndexDesc(dim=int(dim), flags=(1 if wide_rows else 0) | (0 if single_launch else 2), capacity_rows=int(capacity_rows),
                               row_base=int(row_base), device=int(device), cand_cap=int(cand_cap))
        _capi.check(self._lib.nmn_index_create(C.byref(desc), C.byref(self._h)))
        self.dim = int(dim)
        self.capacity_rows = int(capacity_rows)
        self.row_base = int(row_base)

    # -- persistence of the device layout ------------------------------------
    def save(self, path):
        """Rows (stride removed) + their magnitudes -> `path` (nmn_index_save)."""
        _capi.check(self._lib.nmn_index_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path, device=-1, capacity_rows=0, row_base=0, cand_cap=0, wide_rows=False, max_file_bytes=0, max_entries=0):
        """nmn_index_load: a shard from a file written by `save`; the limits are the reference's max_index_file_bytes /
        max_index_entries (0 = none).  The upload's magnitudes must match the stored ones bit for bit."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = C.c_void_p()
        desc = _capi.IndexDesc(dim=0, flags=1 if wide_rows else 0, capacity_rows=int(capacity_rows), row_base=int(row_base),
                               device=int(device), cand_cap=int(cand_cap))
        _capi.check(self._lib.nmn_index_load(str(path).encode(), C.byref(desc), int(max_file_bytes), int(max_entries),
                                             C.byref(self._h)))
        self.dim = int(self._lib.nmn_index_dim(self._h))
        self.capacity_rows = max(int(capacity_rows), self.rows)
        self.row_base = int(self._lib.nmn_index_row_base(self._h))
        return self

    @classmethod
    def _view(cls, handle, owner):
        """A non-owning face of a shard some other object owns (nmn_sharded_shard): close() leaves the handle alone."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        self._h = C.c_void_p(handle)
        self._owner = owner
        self.dim = int(self._lib.nmn_index_dim(self._h))
        self.capacity_rows = self.rows
        self.row_base = int(self._lib.nmn_index_row_base(self._h))
        return self

    # -- lifecycle --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_owner", None) is not None:
            self._h = C.c_void_p()
            return
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.nmn_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def rows(self):
        return int(self._lib.nmn_index_rows(self._h))

    @property
    def row_stride(self):
        """Elements per stored row including the zero padding."""
        return int(self._lib.nmn_index_row_stride(self._h))

    # -- data -------------------------------------------------------------------------------
    def upload(self, rows, row0=None):
        """Append (or overwrite from row0) row-major f32 rows held in host memory."""
        rows = _f32(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise _capi.NeumannGpuError(_capi.ERR_DIMENSION_MISMATCH, f"expected [n,{self.dim}], got {rows.shape}")
        if row0 is None:
            row0 = self.rows
        _capi.check(self._lib.nmn_index_upload(self._h, _ptr(rows), int(row0), rows.shape[0]))

    def upload_device(self, rows_t, row0=None, stream=None):
        """Same from a CUDA/HIP torch tensor [n, dim] f32 (contiguous); asynchronous."""
        if row0 is None:
            row0 = self.rows
        assert rows_t.is_cuda and rows_t.is_contiguous() and rows_t.shape[1] == self.dim
        _capi.check(self._lib.nmn_index_upload_device(self._h, C.c_void_p(rows_t.data_ptr()), int(row0),
                                                      rows_t.shape[0], _stream_ptr(stream)))

    def fill_synthetic(self, seed, n, row0=None):
        if row0 is None:
            row0 = self.rows
        _capi.check(self._lib.nmn_index_fill_synthetic(self._h, int(seed), int(row0), int(n)))

    def set_row(self, row, vec):
        vec = _f32(vec)
        assert vec.size == self.dim
        _capi.check(self._lib.nmn_index_set_row(self._h, int(row), _ptr(vec)))

    def set_rows(self, n):
        _capi.check(self._lib.nmn_index_set_rows(self._h, int(n)))

    # -- search -----------------------------------------------------------------------------
    def search(self, queries, k, metric=DistanceMetric.Cosine, mask=None, with_stats=False):
        """SIMILAR TOP-K with host buffers.  Returns (rows u64 [nq,k], scores f32 [nq,k], counts u32 [nq])."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.size == 0 or q.shape[1] == 0:
            raise _capi.NeumannGpuError(_capi.ERR_EMPTY_VECTOR)
        if q.shape[1] != self.dim:
            raise _capi.NeumannGpuError(_capi.ERR_DIMENSION_MISMATCH, f"expected {self.dim}, got {q.shape[1]}")
        nq = q.shape[0]
        k = int(k)
        out_rows = np.empty((nq, max(k, 1)), dtype=np.uint64)
        out_scores = np.empty((nq, max(k, 1)), dtype=np.float32)
        out_counts = np.empty(nq, dtype=np.uint32)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint64)
            need = (self.rows + 63) // 64
            if m.size < need:
                raise _capi.NeumannGpuError(_capi.ERR_BUFFER_TOO_SMALL, f"mask needs {need} words")
        stats = _capi.SearchStats()
        _capi.check(self._lib.nmn_index_search(self._h, _ptr(q), nq, k, int(metric),
                                               None if m is None else _ptr(m), _ptr(out_rows),
                                               _ptr(out_scores), _ptr(out_counts), C.byref(stats) if with_stats else None))
        if with_stats:  # (asked for only then: collecting them is a synchronous copy of the queries' state, ~20 us a call)
            return out_rows, out_scores, out_counts, stats
        return out_rows, out_scores, out_counts

    def search_dmask(self, queries, k, metric, mask_device_ptr):
        """`search` with the selection bitmap already in device memory (e.g. GpuColumns.mask_device)."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq, k = q.shape[0], int(k)
        out_rows = np.empty((nq, max(k, 1)), dtype=np.uint64)
        out_scores = np.empty((nq, max(k, 1)), dtype=np.float32)
        out_counts = np.empty(nq, dtype=np.uint32)
        _capi.check(self._lib.nmn_index_search_dmask(self._h, _ptr(q), nq, k, int(metric), mask_device_ptr,
                                                     _ptr(out_rows), _ptr(out_scores), _ptr(out_counts), None))
        return out_rows, out_scores, out_counts

    def search_pred(self, columns, ops, consts, queries, k, metric=DistanceMetric.Cosine):
        """Filtered search in one call: evaluate the predicate program over `columns` (a GpuColumns covering this
        shard's rows) and search what it selects.  Returns (rows, scores, counts, selected)."""
        from ._capi import PredOp
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        nq, k = q.shape[0], int(k)
        arr = (PredOp * max(len(ops), 1))()
        for i, o in enumerate(ops):
            arr[i] = o if isinstance(o, PredOp) else PredOp(*o)
        cs = np.ascontiguousarray(np.asarray(list(consts), dtype=np.uint64))
        out_rows = np.empty((nq, max(k, 1)), dtype=np.uint64)
        out_scores = np.empty((nq, max(k, 1)), dtype=np.float32)
        out_counts = np.empty(nq, dtype=np.uint32)
        sel = C.c_uint64()
        _capi.check(self._lib.nmn_index_search_pred(self._h, columns._h, arr, len(ops),
                                                    C.c_void_p(cs.ctypes.data) if cs.size else None, cs.size, _ptr(q), nq, k,
                                                    int(metric), _ptr(out_rows), _ptr(out_scores), _ptr(out_counts),
                                                    C.byref(sel), None))
        return out_rows, out_scores, out_counts, int(sel.value)

    def search_device(self, queries_t, k, metric=DistanceMetric.Cosine, mask_t=None, out=None, stream=None):
        """Asynchronous search with torch device tensors.

        queries_t: [nq, dim] f32 cuda.  mask_t: optional int64 cuda tensor holding the u64 bitmap words.
        Returns (rows int64 [nq,k] — bit pattern of the u64 ids, -1 = unused slot; scores f32 [nq,k];
        counts int32 [nq]) allocated on the same device unless `out` supplies them.
        """
        import torch

        assert queries_t.is_cuda and queries_t.dtype == torch.float32 and queries_t.is_contiguous()
        if queries_t.dim() == 1:
            queries_t = queries_t[None, :]
        nq = queries_t.shape[0]
        if queries_t.shape[1] != self.dim:
            raise _capi.NeumannGpuError(_capi.ERR_DIMENSION_MISMATCH, f"expected {self.dim}")
        if out is None:
            rows = torch.empty((nq, k), dtype=torch.int64, device=queries_t.device)
            scores = torch.empty((nq, k), dtype=torch.float32, device=queries_t.device)
            counts = torch.empty((nq,), dtype=torch.int32, device=queries_t.device)
        else:
            rows, scores, counts = out
        _capi.check(self._lib.nmn_index_search_device(
            self._h, C.c_void_p(queries_t.data_ptr()), nq, int(k), int(metric),
            None if mask_t is None else C.c_void_p(mask_t.data_ptr()),
            C.c_void_p(rows.data_ptr()), C.c_void_p(scores.data_ptr()), C.c_void_p(counts.data_ptr()),
            _stream_ptr(stream)))
        return rows, scores, counts

    def set_mirror(self, enabled):
        """What approximate sweeps read.  True / 1 (default): the smallest mirror that serves the shape — the 8-bit mirror for
        1-2 queries over rows whose stride is a multiple of 128 elements up to 4096 (batches: of 256 up to 1536, 2048, 3072), else the bf16
        mirror; a shard keeps one mirror and builds the other only when a call needs it (include/neumann_gpu.h).  2: the bf16 mirror only.  False / 0: the
        f32 corpus itself (rows*dim*4 bytes per query — or per batch of up to 64-128 queries on the matrix cores —, SURVEY §8(d)'s pricing;
        set before the rows arrive, no mirror is ever built).  Results are identical in every mode."""
        _capi.check(self._lib.nmn_index_set_mirror(self._h, int(enabled)))

    def scan_history(self, stream=None):
        """Sweep durations (ms) of the timed searches enqueued on `stream` since the last call (at most the 64 most recent)."""
        buf = (C.c_float * 64)()
        n = C.c_uint32(0)
        _capi.check(self._lib.nmn_index_scan_history(self._h, _stream_ptr(stream), buf, 64, C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def hbm_bytes(self):
        """(corpus_bytes, mirror_bytes, per_row_bytes) the shard holds in device memory right now (nmn_index_hbm_bytes)."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _capi.check(self._lib.nmn_index_hbm_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def retry_declined(self):
        """Forget the shard's out-of-memory verdicts (a declined mirror, a shrunk query pass): the next search asks the device again."""
        _capi.check(self._lib.nmn_index_retry_declined(self._h))

    def set_timing(self, enabled):
        """0 / False: off; 1 / True: every event (scan_ms and total_ms of last_stats); 2: the sweep's two events only."""
        _capi.check(self._lib.nmn_index_set_timing(self._h, int(enabled)))

    def last_stats(self, stream=None):
        st = _capi.SearchStats()
        _capi.check(self._lib.nmn_index_last_stats(self._h, _stream_ptr(stream), C.byref(st)))
        return st

    # -- exact helpers ----------------------------------------------------------------------
    def score_rows(self, queries, local_rows, metric=DistanceMetric.Cosine):
        """Reference-order scores of explicit rows: f32 [nq, len(local_rows)]."""
        q = _f32(queries)
        if q.ndim == 1:
            q = q[None, :]
        r = np.ascontiguousarray(local_rows, dtype=np.uint64)
        out = np.empty((q.shape[0], r.size), dtype=np.float32)
        _capi.check(self._lib.nmn_index_score_rows(self._h, _ptr(q), q.shape[0], int(metric), _ptr(r), r.size,
                                                   _ptr(out)))
        return out

    def read_probe(self, reps=3):
        """GB/s of a pure read sweep over this shard (the scan's access pattern without arithmetic)."""
        out = C.c_double()
        _capi.check(self._lib.nmn_index_read_probe(self._h, int(reps), C.byref(out)))
        return float(out.value)

    def callers_probe(self, queries, k, metric=DistanceMetric.Cosine, seconds=1.0):
        """len(queries) native threads each calling the single-query host API in a loop: dict(calls_per_s,
        merged_batches, merged_calls, mismatches)."""
        q = _f32(queries)
        qps, b, r, bad = C.c_double(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _capi.check(self._lib.nmn_index_callers_probe(self._h, _ptr(q), q.shape[0], int(k), int(metric), float(seconds),
                                                      C.byref(qps), C.byref(b), C.byref(r), C.byref(bad)))
        return {"calls_per_s": qps.value, "merged_batches": b.value, "merged_calls": r.value, "mismatches": bad.value}

    def coalesce_stats(self):
        """(batches that merged >= 2 concurrent host-buffer searches, searches merged)."""
        b, r = C.c_uint64(), C.c_uint64()
        _capi.check(self._lib.nmn_index_coalesce_stats(self._h, C.byref(b), C.byref(r)))
        return b.value, r.value

    def count_exact(self, query, score, metric=DistanceMetric.Cosine, mask=None):
        """(#rows with exact score > score, #rows with exact score == score) — a full exact pass."""
        q = _f32(query).reshape(-1)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint64)
        gt, eq = C.c_uint64(), C.c_uint64()
        _capi.check(self._lib.nmn_index_count_exact(self._h, _ptr(q), int(metric),
                                                    None if m is None else _ptr(m), C.c_float(float(score)),
                                                    C.byref(gt), C.byref(eq)))
        return gt.value, eq.value


def _stream_ptr(stream):
    """None -> torch's current stream if torch is imported and CUDA is up, else the null stream."""
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        return C.c_void_p(0)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


def merge_topk_host(rows, scores, counts, k):
    """ResultMerger::merge_top_k (query_router/src/distributed.rs:413-433) over [lists][nq][k] host arrays."""
    lib = _capi.load()
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    scores = _f32(scores)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n_lists, nq = counts.shape
    o_r = np.empty((nq, k), dtype=np.uint64)
    o_s = np.empty((nq, k), dtype=np.float32)
    o_c = np.empty(nq, dtype=np.uint32)
    _capi.check(lib.nmn_merge_topk_host(_ptr(rows), _ptr(scores), _ptr(counts), n_lists, nq, int(k), _ptr(o_r),
                                        _ptr(o_s), _ptr(o_c)))
    return o_r, o_s, o_c


def merge_topk_device(rows_t, scores_t, counts_t, k, stream=None):
    """Device merge of an all-gathered [lists][nq][k] block (torch tensors: int64, f32, int32)."""
    import torch

    lib = _capi.load()
    n_lists, nq = counts_t.shape
    o_r = torch.empty((nq, k), dtype=torch.int64, device=rows_t.device)
    o_s = torch.empty((nq, k), dtype=torch.float32, device=rows_t.device)
    o_c = torch.empty((nq,), dtype=torch.int32, device=rows_t.device)
    _capi.check(lib.nmn_merge_topk_device(C.c_void_p(rows_t.data_ptr()), C.c_void_p(scores_t.data_ptr()),
                                          C.c_void_p(counts_t.data_ptr()), n_lists, nq, int(k),
                                          C.c_void_p(o_r.data_ptr()), C.c_void_p(o_s.data_ptr()),
                                          C.c_void_p(o_c.data_ptr()), _stream_ptr(stream)))
    return o_r, o_s, o_c


def packed_layout(nq, k):
    """Byte layout of one rank's packed result block: (size, offset of scores, offset of counts)."""
    off_s = nq * k * 8
    off_c = off_s + nq * k * 4
    return (off_c + nq * 4 + 15) // 16 * 16, off_s, off_c


def merge_topk_device_packed(gathered_t, n_lists, nq, k, out=None, stream=None):
    """Device merge of `n_lists` PACKED blocks ([rows i64 | scores f32 | counts i32], see packed_layout)
    laid end to end in the uint8 tensor `gathered_t` — the output of ONE all-gather."""
    import torch

    lib = _capi.load()
    size, off_s, off_c = packed_layout(nq, k)
    assert gathered_t.dtype == torch.uint8 and gathered_t.numel() >= n_lists * size
    if out is None:
        out = (torch.empty((nq, k), dtype=torch.int64, device=gathered_t.device),
               torch.empty((nq, k), dtype=torch.float32, device=gathered_t.device),
               torch.empty((nq,), dtype=torch.int32, device=gathered_t.device))
    g = gathered_t.data_ptr()
    _capi.check(lib.nmn_merge_topk_device_strided(
        C.c_void_p(g), C.c_void_p(g + off_s), C.c_void_p(g + off_c), size, int(n_lists), int(nq), int(k),
        C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
        _stream_ptr(stream)))
    return out


def synth_rows(seed, row0, n, dim):
    """Host copy of the synthetic generator (bit-identical to the device fill)."""
    out = np.empty((n, dim), dtype=np.float32)
    _capi.check(_capi.load().nmn_synth_fill_host(_ptr(out), int(seed), int(row0), int(n), int(dim)))
    return out
