"""world_size-2 (and 3) gloo runs of the sharded search path on CPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_rows():
    from neumann_amd.sharded import shard_range
    for total, world in ((10, 1), (10, 3), (3001, 2), (1, 4), (0, 2), (80_000_000, 8)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(0 <= lo <= hi for lo, hi in spans)
    assert shard_range(80_000_000, 8, 3) == (30_000_000, 40_000_000)  # config 4: GPU g owns [g*10M, (g+1)*10M)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_over_gloo(world, tmp_path):
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        res = json.load(open(f"{out}.{rank}"))
        assert res and all(res.values()), (rank, res)
