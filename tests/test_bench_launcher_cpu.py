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


def test_roofline_keeps_every_config_inside_the_first_24_keys():
    """The driver's record keeps the FIRST 24 keys of `roofline` (scalars; strings cut at 120 characters): the figures that answer
    "what fraction of the roofline on every BASELINE.json config" must be those 24, in a fixed order, whatever else the run adds
    (VERDICT r05 #1a: 102 scalars were emitted and every c2 / c3 / c5 figure fell off the end)."""
    sys.path.insert(0, ROOT)
    import bench
    canned = {"bound": "hbm", "achieved": 6800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.85, "traffic": 30.8e9,
              "traffic_source": "x" * 300, "traffic_read_write": [1.0, 2.0], "kernel": "nmn::scan_ring_kernel", "avg_kernel_ms": 4.5,
              "kernel_launches_timed": 50, "avg_kernel_ms_from": "y" * 300, "avg_kernel_ms_alone": 4.6, "pricing": "z" * 200,
              "bytes_per_corpus_element": 4, "algorithmic_bytes_per_launch": 30720000000, "candidates_rescored": 102,
              "sweep_kind": "ring_f32", "sweep_launches": 1, "ring_only_read_ceiling": 7100.0, "frac_of_read_ceiling": 0.95}
    for pfx in ("i8_mirror_", "bf16_mirror_"):
        for k in ("queries_per_s", "ms_per_step", "avg_kernel_ms", "frac_on_mirror_bytes", "traffic", "exact_and_same_answer"):
            canned[pfx + k] = 1.0
    for pfx in ("c3_f32_", "c3_i8_", "c2_f32_", "c2_i8_", "c5_mask1.0_f32_", "c5_mask0.5_f32_", "c5_mask0.1_f32_", "c5_mask0.1_i8_"):
        for k in ("qps", "ms_per_batch", "ms_per_step", "frac", "frac_on_mirror_bytes", "exact", "step_frac"):
            canned[pfx + k] = 1.0
    roof, notes = bench.order_roofline(canned)
    first = list(roof)[:24]
    assert first == list(bench.ROOFLINE_FIRST) and len(bench.ROOFLINE_FIRST) == 24
    for must in ("frac", "traffic", "kernel", "c3_f32_frac", "c2_f32_frac", "c5_mask1.0_f32_frac", "c5_mask0.5_f32_frac",
                 "c5_mask0.1_f32_frac", "c3_i8_frac_on_mirror_bytes", "i8_mirror_frac_on_mirror_bytes", "ring_only_read_ceiling"):
        assert must in first
    # no prose and no list in the kept window; nothing lost
    assert all(not isinstance(roof[k], (list, dict)) and (not isinstance(roof[k], str) or len(roof[k]) <= 120) for k in first)
    assert set(notes) == set(bench.ROOFLINE_PROSE) and not (set(notes) & set(roof))
    assert set(roof) | set(notes) == set(canned)
    # a leg that did not run leaves its slot (None), the positions never shift
    roof2, _ = bench.order_roofline({"bound": "hbm", "frac": 0.8})
    assert list(roof2)[:24] == list(bench.ROOFLINE_FIRST) and roof2["c3_f32_frac"] is None


def test_bench_names_a_kernel_for_every_sweep_kind_the_library_can_report():
    """bench.py prints the sweep the LIBRARY reports (nmn_search_stats.sweep_kind) — every NMN_SWEEP_* name has its kernel."""
    sys.path.insert(0, ROOT)
    import bench
    from neumann_amd import _capi
    lib = _capi.load()
    names = [lib.nmn_sweep_kind_str(i).decode() for i in range(9)]
    assert names == ["none", "ring_f32", "valu_f32", "valu_bf16", "valu_i8", "mfma_f32", "mfma_bf16", "mfma_i8", "exact"]
    assert lib.nmn_sweep_kind_str(99).decode() == "unknown"
    assert set(names) == set(bench.KERNEL_OF_SWEEP)
    import ctypes as C
    assert C.sizeof(_capi.SearchStats) == 40
