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
        import torch.distributed as dist
        dist.all_gather_into_tensor(b["gathered"], b["block"], group=self.group)
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
