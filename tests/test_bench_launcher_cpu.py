"""bench.py as its own launcher: `python bench.py --gpus N` with no torchrun around it must start the N ranks
itself, and must REFUSE — non-zero exit, a reason on stderr, no JSON line — when fewer than N devices are visible instead of
quietly measuring one GPU.  This box has no GPU at all, which is the refusal case; the run with two real ranks is the
`-m gpu` test tests/test_gpu_bench_multi.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NMN_BENCH_DEVICE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300, env=e)


def _has_gpu():
    import ctypes as C
    from neumann_amd import _capi
    n = C.c_int32(0)
    _capi.load().nmn_device_count(C.byref(n))
    return n.value


def test_refuses_more_gpus_than_are_visible():
    want = _has_gpu() + 2
    r = _run(["--gpus", str(want), "--steps", "2", "--warmup", "1", "--rows", "1000"])
    assert r.returncode != 0
    assert "refusing" in r.stderr and f"--gpus {want}" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no number may be printed for a refused run"


def test_refuses_a_world_size_that_is_not_the_gpu_count_asked_for():
    # a launcher that started 3 ranks for `--gpus 2`: every rank leaves, rank 0 says why (checked before any device is touched)
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "1000"],
             env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
