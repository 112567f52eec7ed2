"""Worker of tests/test_distributed_gloo.py: one rank of a world_size-N gloo group on CPU.

Exercises the N>1 path of neumann_amd.sharded.ShardedSearcher (shard ranges, padding, all-gather,
router-side merge through the C ABI's nmn_merge_topk_host).  The per-shard local search is supplied by
the CPU oracle here because this container has no GPU; on the GPU box the same class runs the HIP
path (tests/test_gpu_sharded.py, bench.py).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from neumann_amd.sharded import ShardedSearcher, shard_range
    from oracle import oracle_c as oc

    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, d, k, nq = 3001, 48, 37, 3          # 3001 rows: the last shard is ragged
    A = oc.synth(0xD157, 0, n, d)
    A[100:140] = A[100]                    # duplicates that straddle... nothing: all inside shard 0
    A[n - 5:] = A[100]                     # ...and copies of them in the LAST shard: cross-shard ties
    Q = oc.synth(0xD158, 0, nq, d)
    Q[1] = A[100]
    r0, r1 = shard_range(n, world, rank)
    keep = (np.arange(n) % 3 != 1)
    results = {}
    for metric in (0, 1, 2):
        for tag, mask_all in (("all", None), ("masked", keep)):
            def local_search(q, kk, m, _mask, _lo=r0, _hi=r1, _keep=mask_all):
                rows = np.full((q.shape[0], kk), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
                scores = np.full((q.shape[0], kk), -np.inf, dtype=np.float32)
                counts = np.zeros(q.shape[0], dtype=np.uint32)
                mk = None if _keep is None else oc.mask_from_bool(_keep[_lo:_hi])
                for i in range(q.shape[0]):
                    r, s = oc.search(A[_lo:_hi], q[i], kk, m, mask=mk, row_base=_lo)
                    rows[i, :r.size], scores[i, :r.size], counts[i] = r, s, r.size
                return rows, scores, counts

            ss = ShardedSearcher(None, world_size=world, rank=rank, k=k, nq=nq, local_search=local_search)
            rows, scores, counts = ss.search_host(Q, metric)
            ok = True
            for i in range(nq):
                er, es = oc.search(A, Q[i], k, metric, mask=None if mask_all is None else oc.mask_from_bool(mask_all))
                ok &= bool(counts[i] == er.size and np.array_equal(rows[i, :er.size], er)
                           and np.array_equal(scores[i, :er.size].view(np.uint32), es.view(np.uint32)))
            results[f"m{metric}_{tag}"] = ok
    # empty shard: more ranks than rows
    r0e, r1e = shard_range(1, world, rank)
    tiny = oc.synth(1, 0, 1, d)

    def local_tiny(q, kk, m, _mask):
        rows = np.full((q.shape[0], kk), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        scores = np.full((q.shape[0], kk), -np.inf, dtype=np.float32)
        counts = np.zeros(q.shape[0], dtype=np.uint32)
        if r1e > r0e:
            for i in range(q.shape[0]):
                r, s = oc.search(tiny, q[i], kk, m)
                rows[i, :r.size], scores[i, :r.size], counts[i] = r, s, r.size
        return rows, scores, counts

    ss = ShardedSearcher(None, world_size=world, rank=rank, k=4, nq=1, local_search=local_tiny)
    rows, scores, counts = ss.search_host(Q[:1], 0)
    results["tiny"] = bool(counts[0] == 1 and rows[0, 0] == 0 and rows[0, 1] == np.uint64(0xFFFFFFFFFFFFFFFF))
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(results, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
