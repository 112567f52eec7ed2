"""ShardedSearcher on the GPU (world size 1 on the 1-GPU box; the N>1 collective path is covered over
gloo in tests/test_distributed_gloo.py and its packed merge in tests/test_gpu_golden.py)."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu


def test_sharded_searcher_world1_matches_oracle():
    import torch
    from neumann_amd import GpuFlatIndex
    from neumann_amd.sharded import ShardedSearcher, shard_range
    n, d, k, nq = 30000, 128, 50, 3
    A = oc.synth(11, 0, n, d)
    Q = oc.synth(12, 0, nq, d)
    r0, r1 = shard_range(n, 1, 0)
    with GpuFlatIndex(d, r1 - r0, row_base=r0) as idx:
        idx.fill_synthetic(11, r1 - r0)
        ss = ShardedSearcher(idx, world_size=1, rank=0, k=k, nq=nq, device=torch.device("cuda", 0))
        for metric in (0, 1, 2):
            rows, scores, counts = ss.search_device(torch.from_numpy(Q).cuda(), metric)
            torch.cuda.synchronize()
            rows, scores, counts = rows.cpu().numpy().view(np.uint64), scores.cpu().numpy(), counts.cpu().numpy()
            for qi in range(nq):
                er, es = oc.search(A, Q[qi], k, metric)
                assert counts[qi] == k and np.array_equal(rows[qi], er) and np.all(scores[qi] == es)
        # host path of the same class (router-side merge with world size 1 = identity)
        hr, hs, hc = ss.search_host(Q, 0)
        er, es = oc.search(A, Q[0], k, 0)
        assert np.array_equal(hr[0], er) and np.all(hs[0] == es)
