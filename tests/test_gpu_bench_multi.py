"""bench.py's N > 1 path end to end on ONE GPU: plain `python bench.py --gpus 2` — no torchrun — must start
two ranks itself, shard the corpus by row range, all-gather rank ids and top-k blocks, merge, certify, and say n_gpus == 2.
RCCL refuses two ranks on one device, so the ranks are pinned to device 0 and the collective runs over gloo (staged through
the host): the launcher, the rendezvous, the sharding, the packed gather + device merge and the certificate are the real
ones; the xGMI transport is the driver's 8-GPU run.  The one-process handle leg runs too (logical shards on device 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_bench_gpus_2_starts_two_ranks_and_certifies():
    env = dict(os.environ, NMN_BENCH_DEVICE="0", NMN_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--rows", "300000", "--rebuilds", "2"], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    assert d["config"]["rows_total"] == 600000 and d["config"]["rows_per_gpu"] == 300000
    m = d["multi_gpu"]
    assert m["ranks"] == 2 and m["rows_per_gpu"] == [300000, 300000] and m["gather_plus_merge_ms"] > 0
    assert m["rccl_ranks"] == 0 and m["collective_backend"] == "gloo"   # (2 under RCCL: one rank per device)
    assert d["parity"]["exact_topk_certified"] is True and d["parity"]["returned"] == 100
    assert len(d["rebuilds"]["queries_per_s"]) == 2 and d["value"] > 0
    # `value` is queries/s over the WHOLE corpus at every N (BASELINE's metric): steps / elapsed; the aggregate of shard scans
    # (the quantity that grows with N under weak scaling) sits under multi_gpu only
    assert d["unit"] == "queries/s" and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert abs(m["shard_scans_per_s"] - 2 * d["value"]) < 1e-6 * d["value"] and "whole corpus" in d["value_counts"]
    # the kernel time is OF the timed loop: it cannot exceed the step time
    assert 0 < d["roofline"]["avg_kernel_ms"] <= d["ms_per_step"] and d["roofline"]["kernel_launches_timed"] == 6
    # a future SCALE_r*.json must be checkable from the scalars the driver keeps (`config`): the sweep every rank's LIBRARY
    # reported, the ranks the collective saw, the cost of gather + merge per step
    c = d["config"]
    assert c["sweep_kind"] == "ring_f32" and c["sweep_kind_by_rank"] == "ring_f32,ring_f32" and d["roofline"]["kernel"] == "nmn::scan_ring_kernel"
    assert c["rccl_ranks"] == 0 and c["gather_plus_merge_ms"] == m["gather_plus_merge_ms"] and c["shards"] == 2
    assert list(d["roofline"])[:24] == ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms",
                                        "algorithmic_bytes_per_launch", "c3_f32_frac", "c3_f32_ms_per_batch", "c3_f32_qps", "c2_f32_frac",
                                        "c2_f32_qps", "c5_mask1.0_f32_frac", "c5_mask0.5_f32_frac", "c5_mask0.1_f32_frac",
                                        "c5_mask1.0_f32_qps", "c5_mask0.1_f32_qps", "i8_mirror_queries_per_s",
                                        "i8_mirror_frac_on_mirror_bytes", "c3_i8_frac_on_mirror_bytes", "c3_i8_qps", "ring_only_read_ceiling"]
    h = m["one_process_handle"]
    assert "error" not in h, h
    assert h["rows_per_gpu"] == [300000, 300000] and h["exact_topk_certified"] is True and h["gather"] == "peer copies"


def test_strong_scaling_splits_the_rows():
    env = dict(os.environ, NMN_BENCH_DEVICE="0", NMN_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "1",
                        "--rows", "200000", "--rebuilds", "1", "--scaling", "strong", "--no-handle-leg"],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["rows_total"] == 200000
    assert d["multi_gpu"]["rows_per_gpu"] == [66667, 66667, 66666] and d["parity"]["exact_topk_certified"] is True
    assert d["config"]["sweep_kind_by_rank"] == "valu_f32,valu_f32,valu_f32" and d["config"]["gather_plus_merge_ms"] > 0
