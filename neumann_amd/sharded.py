"""Row-range sharding of one corpus over the GPUs of a node — one process per GPU.

The reference's only "collective" on this path is the application-level scatter-gather of the
distributed planner: every shard runs the same SIMILAR locally and `ResultMerger::merge_top_k`
concatenates, sorts by score descending and truncates to k (query_router/src/distributed.rs:173-180,
413-433).  Here the shards are GPUs: rank g owns global rows [g*ceil(N/G), ...), queries are
replicated, each rank produces its exact local top-k padded to k with (-inf, u64::MAX), ONE all-gather
(RCCL over xGMI; payload nq*k*12 B per rank, latency-bound) brings every block to every rank, and the
merge kernel ranks the G*k candidates by (score desc, row asc).  Global top-k is a subset of the
union of local top-k lists, so recall stays 1.0 by construction.

The same class drives the CPU/gloo tests: with `local_search=` injected the local step is whatever
the test supplies, the gather runs over gloo and the merge is the host (`router-side`) merge.
"""
import numpy as np

from . import flat_index

U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_range(total_rows, world_size, rank):
    """Contiguous row range [r0, r1) of `rank` (GPU g holds rows [g*ceil(N/G), ...))."""
    per = (total_rows + world_size - 1) // world_size
    r0 = min(rank * per, total_rows)
    r1 = min(r0 + per, total_rows)
    return r0, r1


class GpuShardedIndex:
    """ONE process, several devices: the `nmn_sharded` handle of the C ABI (include/neumann_gpu.h) — what a Rust
    `vector_engine` holding an Arc<VectorEngine> binds to use every GPU of the node.  Shard g holds the global rows
    [g*ceil(N/G), ...) on devices[g]; `search` replicates the queries, runs every shard's pipeline on its own stream,
    gathers the per-shard top-k blocks with one RCCL all-gather (distinct devices) or peer copies (logical shards on one
    device) and merges them on shard 0's device — `ResultMerger::merge_top_k` (distributed.rs:413-433)."""

    def __init__(self, dim, capacity_rows, n_shards, devices=None, row_base=0, gather=0, cand_cap=0, wide_rows=False,
                 cyclic=False):
        import ctypes as C
        from . import _capi
        self._lib = _capi.load()
        self._h = C.c_void_p()
        self.dim, self.capacity_rows, self.n_shards, self.row_base = int(dim), int(capacity_rows), int(n_shards), int(row_base)
        devs = None
        if devices is not None:
            assert len(devices) == n_shards
            devs = (C.c_int32 * n_shards)(*[int(d) for d in devices])
        desc = _capi.ShardedDesc(dim=self.dim, flags=1 if wide_rows else 0, capacity_rows=self.capacity_rows,
                                 row_base=self.row_base, n_shards=self.n_shards, gather=int(gather),
                                 devices=devs, cand_cap=int(cand_cap), layout=1 if cyclic else 0)
        _capi.check(self._lib.nmn_sharded_create(C.byref(desc), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            import ctypes as C
            self._lib.nmn_sharded_destroy(self._h)
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
        return int(self._lib.nmn_sharded_rows(self._h))

    @property
    def rccl_ranks(self):
        """communicator ranks the create-time self-test saw answer the all-gather (0: peer-copy gather)"""
        return int(self._lib.nmn_sharded_rccl_ranks(self._h))

    @property
    def layout(self):
        return int(self._lib.nmn_sharded_layout(self._h))

    @property
    def gather_mode(self):
        """1 = RCCL all-gather, 2 = peer copies (what nmn_sharded_create chose)."""
        return int(self._lib.nmn_sharded_gather_mode(self._h))

    def device_of(self, shard):
        return int(self._lib.nmn_sharded_device(self._h, int(shard)))

    def shard_rows(self, shard):
        return int(self._lib.nmn_index_rows(self._lib.nmn_sharded_shard(self._h, int(shard))))

    def shard(self, g):
        """Shard g as a GpuFlatIndex VIEW (the handle stays owned by this object): its exact helpers (score_rows,
        count_exact) serve certificates over a sharded corpus."""
        return flat_index.GpuFlatIndex._view(self._lib.nmn_sharded_shard(self._h, int(g)), self)

    def global_row(self, shard, local_row):
        return int(self._lib.nmn_sharded_global_row(self._h, int(shard), int(local_row)))

    def upload(self, rows, row0=None):
        import ctypes as C
        from . import _capi
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.dim
        if row0 is None:
            row0 = self.rows
        _capi.check(self._lib.nmn_sharded_upload(self._h, C.c_void_p(rows.ctypes.data), int(row0), rows.shape[0]))

    def fill_synthetic(self, seed, n, row0=None):
        from . import _capi
        if row0 is None:
            row0 = self.rows
        _capi.check(self._lib.nmn_sharded_fill_synthetic(self._h, int(seed), int(row0), int(n)))

    def set_timing(self, enabled):
        from . import _capi
        _capi.check(self._lib.nmn_sharded_set_timing(self._h, 1 if enabled else 0))

    def set_mirror(self, enabled):
        from . import _capi
        _capi.check(self._lib.nmn_sharded_set_mirror(self._h, int(enabled)))

    def last_gather_ms(self):
        import ctypes as C
        from . import _capi
        ms = C.c_float()
        _capi.check(self._lib.nmn_sharded_last_gather_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def coalesce_stats(self):
        """(sweeps that carried two or more concurrent calls, calls in them)"""
        import ctypes as C
        from . import _capi
        b, c = C.c_uint64(), C.c_uint64()
        _capi.check(self._lib.nmn_sharded_coalesce_stats(self._h, C.byref(b), C.byref(c)))
        return int(b.value), int(c.value)

    def search(self, queries, k, metric=0, mask=None, with_stats=False):
        """As GpuFlatIndex.search: (rows u64 [nq,k], scores f32 [nq,k], counts u32 [nq]); `mask` covers the GLOBAL rows."""
        import ctypes as C
        from . import _capi
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.size == 0 or q.shape[1] == 0:
            raise _capi.NeumannGpuError(_capi.ERR_EMPTY_VECTOR)
        if q.shape[1] != self.dim:
            raise _capi.NeumannGpuError(_capi.ERR_DIMENSION_MISMATCH, f"expected {self.dim}, got {q.shape[1]}")
        nq, k = q.shape[0], int(k)
        out_rows = np.empty((nq, max(k, 1)), dtype=np.uint64)
        out_scores = np.empty((nq, max(k, 1)), dtype=np.float32)
        out_counts = np.empty(nq, dtype=np.uint32)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint64)
            if m.size < (self.capacity_rows + 63) // 64 and m.size < (self.rows + 63) // 64:
                raise _capi.NeumannGpuError(_capi.ERR_BUFFER_TOO_SMALL, "mask must cover the global rows")
        stats = _capi.SearchStats()
        _capi.check(self._lib.nmn_sharded_search(self._h, C.c_void_p(q.ctypes.data), nq, k, int(metric),
                                                 None if m is None else C.c_void_p(m.ctypes.data),
                                                 C.c_void_p(out_rows.ctypes.data), C.c_void_p(out_scores.ctypes.data),
                                                 C.c_void_p(out_counts.ctypes.data), C.byref(stats) if with_stats else None))
        if with_stats:
            return out_rows, out_scores, out_counts, stats
        return out_rows, out_scores, out_counts


class ShardedSearcher:
    """Per-rank driver: local shard search -> all-gather -> merge.  World size 1 skips the collective."""

    def __init__(self, index, world_size=1, rank=0, k=10, nq=1, device=None, group=None, local_search=None,
                 always_gather=False):
        self.index = index
        self.always_gather = bool(always_gather)  # run the all-gather + merge even with one rank (1-GPU RCCL test)
        self.world_size = int(world_size)
        self.rank = int(rank)
        self.k = int(k)
        self.nq = int(nq)
        self.device = device
        self.group = group
        self.local_search = local_search
        self._bufs = None

    # ---- GPU path ------------------------------------------------------------------------------
    def _alloc(self):
        """One packed block per rank: [rows nq*k i64 | scores nq*k f32 | counts nq i32], padded to 16 B, so
        that ONE all-gather moves all three fields (the step is latency-bound: nq*k*12 B per rank)."""
        import torch
        dev, nq, k, w = self.device, self.nq, self.k, self.world_size
        size, off_s, off_c = flat_index.packed_layout(nq, k)
        block = torch.empty(size, dtype=torch.uint8, device=dev)
        self._bufs = {
            "block": block, "size": size, "off_s": off_s, "off_c": off_c,
            "rows": block[:off_s].view(torch.int64).view(nq, k),
            "scores": block[off_s:off_c].view(torch.float32).view(nq, k),
            "counts": block[off_c:off_c + nq * 4].view(torch.int32),
        }
        if w > 1 or self.always_gather:
            self._bufs["gathered"] = torch.empty(w * size, dtype=torch.uint8, device=dev)
            self._bufs["out"] = (torch.empty((nq, k), dtype=torch.int64, device=dev),
                                 torch.empty((nq, k), dtype=torch.float32, device=dev),
                                 torch.empty((nq,), dtype=torch.int32, device=dev))

    def search_device(self, queries_t, metric, mask_t=None):
        """queries_t [nq, dim] f32 on this rank's GPU (replicated on every rank).  Everything is
        enqueued on the current stream; returns (rows int64 [nq,k], scores f32, counts int32) tensors
        holding the GLOBAL top-k on every rank."""
        if self._bufs is None:
            self._alloc()
        b = self._bufs
        self.index.search_device(queries_t, self.k, metric, mask_t=mask_t,
                                 out=(b["rows"], b["scores"], b["counts"]))
        if self.world_size == 1 and not self.always_gather:
            return b["rows"], b["scores"], b["counts"]
        return self.gather_merge()

    def gather_merge(self):
        """The collective step on its own: all-gather of this rank's packed block (whatever the last search left in it),
        then the device merge.  RCCL (backend "nccl") moves the device buffers directly; under "gloo" — the control-flow
        check of the N > 1 path with every rank on ONE GPU, where RCCL refuses to run — the blocks are staged through the host."""
        import torch
        import torch.distributed as dist
        b = self._bufs
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(b["gathered"], b["block"], group=self.group)
        else:
            torch.cuda.current_stream().synchronize()
            mine = b["block"].cpu()
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            dist.all_gather(parts, mine, group=self.group)
            b["gathered"].copy_(torch.cat(parts), non_blocking=False)
        return flat_index.merge_topk_device_packed(b["gathered"], self.world_size, self.nq, self.k, out=b["out"])

    # ---- host path (gloo tests, router-side merge) -----------------------------------------------
    def search_host(self, queries, metric, mask=None):
        """Host-buffer variant: local search through `local_search` (or the index's host API), gather
        over the process group (any backend that moves CPU tensors), merge with the host merge."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        fn = self.local_search or (lambda qq, kk, mm, mk: self.index.search(qq, kk, mm, mask=mk))
        rows, scores, counts = fn(q, self.k, metric, mask)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        if self.world_size == 1:
            return rows, scores, counts
        import torch
        import torch.distributed as dist
        w = self.world_size
        t_rows = torch.from_numpy(rows.view(np.int64))
        t_scores = torch.from_numpy(scores)
        t_counts = torch.from_numpy(counts.view(np.int32))
        g_rows = [torch.empty_like(t_rows) for _ in range(w)]
        g_scores = [torch.empty_like(t_scores) for _ in range(w)]
        g_counts = [torch.empty_like(t_counts) for _ in range(w)]
        dist.all_gather(g_rows, t_rows, group=self.group)
        dist.all_gather(g_scores, t_scores, group=self.group)
        dist.all_gather(g_counts, t_counts, group=self.group)
        R = np.stack([t.numpy().view(np.uint64) for t in g_rows])
        S = np.stack([t.numpy() for t in g_scores])
        Cn = np.stack([t.numpy().view(np.uint32) for t in g_counts])
        return flat_index.merge_topk_host(R, S, Cn, self.k)
