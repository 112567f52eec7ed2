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


def test_rccl_gather_and_device_merge_single_rank():
    """The N>1 device path end to end on one GPU: an RCCL ("nccl") process group of one rank, the packed
    all-gather on the search stream and the device merge — the exact calls every rank makes at N>1."""
    import os
    import torch
    import torch.distributed as dist
    from neumann_amd import GpuFlatIndex
    from neumann_amd.sharded import ShardedSearcher
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, d, k, nq = 50000, 256, 100, 5
        A = oc.synth(21, 1000, n, d)  # shard rows are generated from their GLOBAL ids
        Q = oc.synth(22, 0, nq, d)
        with GpuFlatIndex(d, n, row_base=1000) as idx:
            idx.fill_synthetic(21, n)
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            ss = [ShardedSearcher(idx, world_size=1, rank=0, k=k, nq=nq, device=dev, always_gather=True) for _ in streams]
            qd = torch.from_numpy(Q).to(dev)
            outs = []
            for i in range(6):  # pipelined over two streams like bench.py
                with torch.cuda.stream(streams[i % 2]):
                    outs.append(ss[i % 2].search_device(qd, i % 3))
            dist.barrier()
            torch.cuda.synchronize()
            for i in (4, 5):  # the last result of each stream is still in its buffers
                rows, scores, counts = (t.cpu().numpy() for t in outs[i])
                for qi in range(nq):
                    er, es = oc.search(A, Q[qi], k, i % 3, row_base=1000)
                    assert counts[qi] == k
                    assert np.array_equal(rows[qi].view(np.uint64), er) and np.all(scores[qi] == es)
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_two_ranks_control_flow_on_one_gpu():
    """bench.py's N>1 path end to end — two processes, row-range shards, per-step gather + device merge, max-over-ranks
    timing, the cross-shard exactness certificate — with both ranks pinned to GPU 0 and gloo as the transport (RCCL
    refuses two ranks on one device; its own calls are covered by the single-rank test above)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMN_BENCH_DEVICE="0", NMN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "300000", "--steps", "6",
           "--warmup", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    # (under a launcher bench.py uses the ranks it is given; weak scaling is the default: --rows rows PER GPU, BASELINE config 4)
    assert d["n_gpus"] == 2 and d["config"]["rows_per_gpu"] == 300000 and d["config"]["rows_total"] == 600000 and d["scaling"] == "weak"
    assert d["parity"]["exact_topk_certified"] and d["parity"]["returned"] == 100
    assert d["value"] > 0 and d["cpu_baseline"] is None
    assert d["multi_gpu"]["ranks"] == 2 and d["multi_gpu"]["rows_per_gpu"] == [300000, 300000]
