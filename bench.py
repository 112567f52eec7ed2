#!/usr/bin/env python
"""bench.py — SIMILAR TOP-K throughput of the MI355X-native vector_engine hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): 10M x 768 f32 cosine brute-force TOP-100.  One "step" = one pass of the hot path over
one batch of `--nq` synthetic queries (default 1: the HBM-bound single-query scan the roofline target is quoted on).  The
corpus and the queries are resident in HBM before the timed region.

N > 1: ONE PROCESS PER GPU.  `python bench.py --gpus N` with no launcher around it starts the N ranks ITSELF (rank r on
device r, rendezvous on 127.0.0.1) and refuses to run when fewer than N devices are visible; under torch.distributed.run it
uses the ranks it is given and refuses a WORLD_SIZE that is not N.  The corpus is row-range sharded (rank g owns
[g*ceil(T/N), ...)); every step each rank scans its shard, the packed per-shard top-k blocks are all-gathered over RCCL and
merged on-device.  `--scaling weak` (default; BASELINE config 4) keeps `--rows` rows PER GPU — 80M x 768 on 8 GPUs;
`--scaling strong` splits `--rows` rows over the GPUs.  At N = 1 the two are the same run.

Prints ONE JSON line (rank 0).  The driver's record keeps the top-level scalars and the SCALARS of `roofline`, `config` and
`cpu_baseline`; everything a checker needs is therefore a flat scalar there (nested objects carry the detail for a reader):
  value / ms_per_step   queries per second over the WHOLE corpus (rows_total rows) at every N — BASELINE.json's metric — with the
                sweep reading the ROW-MAJOR F32 CORPUS (`--mirror 0`, the default: SURVEY §8(d) prices the path on rows*dim*4
                bytes per query and north_star asks for coalesced reads of the f32 rows).  Under weak scaling the corpus grows
                with N, so `value` staying level is linear scaling; shard scans/s (N x value) is under `multi_gpu`.  The median
                over `--rebuilds` index rebuilds (each: W warmup steps, then EXACTLY K timed steps between a barrier +
                synchronize on both sides, max over ranks); `rebuilds` lists every draw
  dtype         "f32": what the headline sweep reads and computes in
  roofline      the dominant kernel of the timed loop (nmn::scan_ring_kernel over the f32 rows; nmn::scan_kernel for bitmaps / small shards).  `avg_kernel_ms` = HIP events the
                library records on the launch stream around that kernel in EVERY timed step (nmn_index_scan_history), of the
                very loop `value` comes from; sweeps of a shard never run side by side (the next one starts, on the device, when
                the previous one ends; the selection / rescore tail of a step runs under the next step's sweep), so kernel <=
                step.  `achieved` = algorithmic bytes (rows x dim x 4 per query) / avg_kernel_ms; `traffic` = PMC bytes per launch
  roofline.i8_mirror_* / bf16_mirror_*   the SAME loop (index, queries, streams, steps) on the library's own default: the sweep
                reads the shard's 8-bit (or bf16) mirror and every candidate is re-scored from the f32 rows in the reference's
                order — the same answer bit for bit (…_exact_and_same_answer), an exact acceleration priced on the bytes the
                mirror sweep is asked to read (…_frac_on_mirror_bytes), never a §8(d) figure
  roofline.c3_f32_* / c3_i8_*   config 3 (64 queries per step) on the f32 corpus (…_frac: 30.72 GB per batch / sweep time / 8 TB/s)
                and on the mirror; roofline.c2_* / c5_mask*_*: configs 2 and 5 (child runs), f32 sweep and mirror sweep each
  config.shards_with_mirror   how many of the N shards served the mirror leg from a mirror (a shard short of HBM declines its
                mirror and sweeps f32 rows: correct, slower — it must be visible); config.bytes_per_corpus_element: the headline's
  cpu_baseline  the CPU oracle (oracle/nmn_oracle.c, -O3 -march=native, all host cores; cpu_model says which) on the WHOLE
                corpus when the host's RAM holds its twin (else a 1M-row sample, flagged `extrapolated`)
  parity        size-independent exactness certificate of the last timed result (see certificate())
  mirror_legs, batched, concurrent_callers, other_configs, next_rows   the detail behind the flat scalars, and further legs of
                the default single-GPU run (64 / 128 host threads; SURVEY §8(f): filtered SIMILAR end to end, IVF probe, upload,
                index load)
  multi_gpu     (N > 1) rccl_ranks (ranks seen by a real all-gather of rank ids), rows_per_gpu[], gather_plus_merge_ms,
                shard_scans_per_s, bytes per corpus element of every rank's sweeps, and `one_process_handle`: the same GPUs
                driven by ONE process through the C ABI's nmn_sharded
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable by a copy)
MFMA_I8_PEAK_TOPS = 3944.0     # dense int8, v_mfma_i32_16x16x64_i8 (MI355X_MICROARCH.md, matrix-core table)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (same table)
SEED_CORPUS = 0x5EED0003
SEED_QUERY = 0x5EED0002
METRICS = {"cosine": 0, "euclidean": 1, "dot": 2}
SWEEP = {1: ("i8", "8-bit mirror (int8 codes, one f32 scale per row), int32 accumulate", "rows*dim*1, the 8-bit mirror"),
         2: ("bf16", "bf16 mirror of the corpus, f32 accumulate", "rows*dim*2, the bf16 mirror"),
         4: ("f32", "f32 corpus", "SURVEY §8(d): rows*dim*4, the f32 corpus")}


# Kernel behind each nmn_search_stats.sweep_kind (include/neumann_gpu.h NMN_SWEEP_*): the library REPORTS which sweep served a
# search; bench.py prints that, it does not re-derive the dispatch.
KERNEL_OF_SWEEP = {"ring_f32": "nmn::scan_ring_kernel", "valu_f32": "nmn::scan_kernel", "valu_bf16": "nmn::scan_kernel",
                   "valu_i8": "nmn::scan_i8_kernel", "mfma_f32": "nmn::scan_mfma_kernel", "mfma_bf16": "nmn::scan_mfma_kernel",
                   "mfma_i8": "nmn::scan_mfma_kernel", "exact": "nmn::exact_scan_kernel", "none": None}
# queries one corpus sweep of each kind serves (what `passes`, the sweeps per step, follows from)
QUERIES_PER_SWEEP = {"ring_f32": 1, "valu_f32": 4, "valu_bf16": 4, "valu_i8": 2}  # (mfma_*: a whole pass; exact: the first query's scan is the timed launch)

# The driver's record keeps the FIRST 24 keys of `roofline` (scalars; strings cut at 120 characters) — so the 24 figures a checker
# needs to answer "what fraction of the roofline on EVERY BASELINE.json config" come first, in this order; everything else follows,
# prose goes to the top-level `notes` (tests/test_bench_launcher_cpu.py asserts the order on a canned run).
ROOFLINE_FIRST = (
    "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms", "algorithmic_bytes_per_launch",
    "c3_f32_frac", "c3_f32_ms_per_batch", "c3_f32_qps", "c2_f32_frac", "c2_f32_qps",
    "c5_mask1.0_f32_frac", "c5_mask0.5_f32_frac", "c5_mask0.1_f32_frac", "c5_mask1.0_f32_qps", "c5_mask0.1_f32_qps",
    "i8_mirror_queries_per_s", "i8_mirror_frac_on_mirror_bytes", "c3_i8_frac_on_mirror_bytes", "c3_i8_qps",
    "ring_only_read_ceiling")
ROOFLINE_PROSE = ("traffic_source", "avg_kernel_ms_from", "pricing", "traffic_read_write")


def order_roofline(roof):
    """(ordered roofline dict, notes dict): ROOFLINE_FIRST's keys first and in that order — present in EVERY line, None where a leg
    did not run, so that the positions never shift —, then the remaining scalars; prose and lists move out into `notes`."""
    notes = {k: roof[k] for k in ROOFLINE_PROSE if k in roof}
    out = {k: roof.get(k) for k in ROOFLINE_FIRST}
    for k, v in roof.items():
        if k not in out and k not in notes:
            out[k] = v
    return out, notes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows per GPU (weak) / in total (strong)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--nq", type=int, default=1, help="queries per step")
    ap.add_argument("--metric", default="cosine", choices=sorted(METRICS))
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="weak (default, BASELINE config 4): --rows rows PER GPU; strong: --rows rows split over the GPUs")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are pipelined over (each stream runs whole steps in order; with 2 "
                         "the select/rescore/gather tail of one query overlaps the next query's scan)")
    ap.add_argument("--rebuilds", type=int, default=3,
                    help="index builds the timed loop is repeated over; value = the median (buffer placement moves one draw by several %%)")
    ap.add_argument("--mask", type=float, default=1.0,
                    help="selectivity of a synthetic WHERE-predicate bitmap (config 5); 1.0 = no mask")
    ap.add_argument("--mirror", type=int, default=0, choices=[0, 1, 2],
                    help="nmn_index_set_mirror of the HEADLINE loop: 0 = the row-major f32 corpus (default: SURVEY §8(d) prices the path on "
                         "rows*dim*4 bytes, and `value` / `roofline` are that sweep), 1 = the smallest mirror that serves the call (the "
                         "library's own default; reported beside the headline as roofline.i8_mirror_*), 2 = bf16 mirror only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--callers", type=int, default=64, help="host threads of the concurrent-callers leg (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline leg")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-published-shapes", action="store_true",
                    help="skip cpu_baseline's pub_* scalars (the reference's own bench shapes, 1 000 x 128 ... 10 000 x 128, top-10)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra legs of the default single-GPU run: BASELINE.json's other single-GPU "
                         "configurations (config 2, config 5 with mask 1.0 / 0.5 / 0.1) and the SURVEY §8(f) legs")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (then it comes from profiles/pmc_traffic.json)")
    ap.add_argument("--no-mirror-legs", "--no-f32-leg", dest="no_mirror_legs", action="store_true",
                    help="skip the roofline.f32_corpus / roofline.bf16_mirror legs")
    ap.add_argument("--legs", default="i8,bf16", help="mirror sweeps measured beside an f32 headline (comma list of i8, bf16)")
    ap.add_argument("--always-gather", action="store_true",
                    help="run the all-gather + device merge even with one rank (what the N>1 step adds, on a 1-GPU box)")
    ap.add_argument("--batched", type=int, default=64,
                    help="also measure config 3 (this many queries per step on the MFMA sweep) and report it under \"batched\" "
                         "(single-GPU runs only; 0 = skip)")
    ap.add_argument("--no-handle-leg", action="store_true", help="N > 1: skip the one-process nmn_sharded leg")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--next-rows-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--handle-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def _synth(seed, row0, n, dim):
    from neumann_amd import synth_rows
    return synth_rows(seed, row0, n, dim)


def _visible_devices():
    import ctypes as C
    from neumann_amd import _capi
    n = C.c_int32(0)
    _capi.load().nmn_device_count(C.byref(n))
    return int(n.value)


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process becomes the launcher
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    n = args.gpus
    visible = _visible_devices()
    pinned = os.environ.get("NMN_BENCH_DEVICE")  # debugging aid: every rank on ONE device (needs NMN_BENCH_BACKEND=gloo)
    if pinned is None and visible < n:
        print(f"[bench] --gpus {n} but only {visible} HIP device(s) are visible: refusing to measure fewer GPUs than asked for",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
        rc = rc or p.returncode
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    return rc or 0


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, metric, total_rows, device, gpu_answer=None):
    """Time the CPU oracle (and, as the checker, compare the GPU path with it on the same rows); returns the cpu_baseline
    object.  The WHOLE configured corpus when the host's RAM holds its twin (SURVEY §8(d) allows extrapolation only where it
    does not): `gpu_answer` = (query index, rows, scores, counts) of the resident GPU index for one of the timed queries.
    Otherwise a 1M-row sample, scaled linearly and flagged as extrapolated."""
    from oracle import oracle_c as oc
    from neumann_amd import GpuFlatIndex
    cores, quota = _effective_cores()
    try:
        ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError):
        ram = 0
    full_bytes = total_rows * args.dim * 4
    full = ram >= 2 * full_bytes + (16 << 30) and not os.environ.get("NMN_BENCH_CPU_SAMPLE")
    sample_rows = total_rows if full else min(1_000_000, total_rows)
    t0 = time.perf_counter()
    A = oc.synth(SEED_CORPUS, 0, sample_rows, args.dim, nthreads=cores)
    synth_s = time.perf_counter() - t0
    Q = oc.synth(SEED_QUERY, 0, 4, args.dim)
    oc.search(A, Q[0], args.k, metric, partial=True, nthreads=cores, native=True)  # warm (page-in, threads)
    t0 = time.perf_counter()
    er, es = oc.search(A, Q[1], args.k, metric, partial=True, nthreads=cores, native=True)
    t1 = time.perf_counter() - t0
    reps = int(max(3, min(2000, args.cpu_seconds / max(t1, 1e-4))))
    t0 = time.perf_counter()
    for i in range(reps):
        oc.search(A, Q[i % 4], args.k, metric, partial=True, nthreads=cores, native=True)
    dt = (time.perf_counter() - t0) / reps
    # single-thread figure on a smaller slice, for the record
    t0 = time.perf_counter()
    n1 = min(50_000, sample_rows)
    oc.search(A[:n1], Q[0], args.k, metric, partial=True, nthreads=1, native=True)
    dt1 = (time.perf_counter() - t0) * (1_000_000 / float(n1))
    # checker: the HIP path on the same rows must return the oracle's rows and scores
    if full and gpu_answer is not None:
        qi, gr, gs, gc = gpu_answer
        er, es = oc.search(A, Q[qi], args.k, metric, partial=True, nthreads=cores, native=True)
    else:
        with GpuFlatIndex(args.dim, sample_rows, row_base=0, device=device) as small:
            small.fill_synthetic(SEED_CORPUS, sample_rows)
            gr, gs, gc = small.search(Q[1], args.k, metric)
    sample_ok = bool(gc[0] == er.size and np.array_equal(gr[0, :er.size].astype(np.uint64), er.astype(np.uint64)) and np.all(gs[0, :er.size] == es))
    del A
    qps_full = 1.0 / (dt * (total_rows / sample_rows))
    return {
        "value": qps_full, "unit": "queries/s", "cores": cores, "cpu_model": _cpu_model(), "kind": "port",
        "host_hardware_threads": os.cpu_count(), "cgroup_cpu_quota": quota,
        "sample": (f"{reps} queries x the WHOLE corpus, {sample_rows} rows x {args.dim} (same generator/seed; host twin built in "
                   f"{synth_s:.1f} s), {dt * 1e3:.1f} ms/query on {cores} threads"
                   f"{f' (the cgroup CPU quota of this container: {quota:g} CPUs of the {os.cpu_count()} hardware threads)' if quota else ''}, nothing extrapolated" if full else
                   f"{reps} queries x {sample_rows} rows x {args.dim} (same generator/seed), {dt * 1e3:.2f} ms/query on {cores} threads, "
                   f"EXTRAPOLATED linearly to {total_rows} rows (host RAM {ram >> 30} GiB does not hold two corpus twins)") +
                  f"; 1 thread: {dt1 * 1e3:.0f} ms per 1M rows. Optimistic for the reference (flat array, per-thread partial "
                  f"top-k; the Rust path also pays a BTreeMap lookup, two clones per row and a full sort, published 193-367 ns/row)",
        "extrapolated": not full, "rows_timed": sample_rows,
        "gbps": sample_rows * args.dim * 4 / dt / 1e9,
        "gpu_matches_oracle_on_sample": sample_ok,
        **({} if getattr(args, "no_published_shapes", False) else published_shapes(cores)),
    }


# The only numbers the reference publishes (vector_engine/benches/vector_engine_bench.rs:40-77, `search_similar`
# top-10 over uniform(-1,1) vectors): mean time per call on the authors' CPU.
PUBLISHED_SEARCH_US = {"1000x128": 242.0, "1000x768": 367.0, "10000x128": 1930.0}


def published_shapes(cores):
    """The reference's own bench shapes through the host mirror of its API: nmn_engine_search_similar top-10 (host buffers: query
    H2D, result D2H and key strings inside every call), p50 microseconds over 1 000 native back-to-back calls, beside the CPU oracle
    on the same data (1 thread and all granted cores) and the reference's published figure; and the row count from which one GPU
    call beats the 1-thread oracle at d = 128 / 768.  Flat scalars (they go under `cpu_baseline`, which the driver keeps)."""
    from oracle import oracle_c as oc
    from neumann_amd.engine import VectorEngine
    out = {}
    rng = np.random.default_rng(0x5EED0006)

    def one(n, d, calls):
        A = rng.uniform(-1.0, 1.0, (n, d)).astype(np.float32)
        Q = rng.uniform(-1.0, 1.0, (16, d)).astype(np.float32)
        eng = VectorEngine()
        try:
            eng.batch_store_embeddings([f"v{i}" for i in range(n)], A)
            eng.search_probe(Q, 10, 20)  # builds the resident shard, warms the path
            us = eng.search_probe(Q, 10, calls)
            res = eng.search_similar(Q[0], 10)
        finally:
            eng.close() if hasattr(eng, "close") else None
        er, es = oc.search(A, Q[0], 10, 0)
        same = [r.key for r in res] == [f"v{int(i)}" for i in er] and np.array_equal(np.array([r.score for r in res], np.float32), es)
        cpu = {}
        for nt in sorted({1, cores}):
            reps = max(20, min(calls, int(2e5 / max(n * d / 1e5, 1))))
            oc.search(A, Q[0], 10, 0, partial=True, nthreads=nt, native=True)
            t = []
            for i in range(reps):
                t0 = time.perf_counter()
                oc.search(A, Q[i % 16], 10, 0, partial=True, nthreads=nt, native=True)
                t.append((time.perf_counter() - t0) * 1e6)
            cpu[nt] = float(np.median(t))
        return float(np.median(us)), float(np.percentile(us, 99)), cpu, bool(same)

    same_all, detail = True, {}
    for n, d in ((1000, 128), (1000, 768), (10000, 128)):
        key = f"{n}x{d}"
        try:
            p50, p99, cpu, same = one(n, d, 1000)
            out[f"pub_{key}_gpu_p50_us"] = p50
            out[f"pub_{key}_oracle_1t_us"] = cpu[1]
            out[f"pub_{key}_reference_published_us"] = PUBLISHED_SEARCH_US[key]
            same_all = same_all and same
            detail[key] = {"gpu_p99_us": p99, **{f"oracle_{t}t_us": v for t, v in cpu.items()}}
        except Exception as e:
            same_all = False
            detail[key] = {"error": f"{type(e).__name__}: {e}"[:200]}
    for d in (128, 768):  # the smallest corpus (powers of two) from which one GPU call is faster than the 1-thread oracle
        cross = None
        try:
            for n in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
                p50, _, cpu, _ = one(n, d, 200)
                if p50 < cpu[1]:
                    cross = n
                    break
        except Exception:
            cross = None
        out[f"pub_crossover_rows_d{d}"] = cross
    out["pub_same_answers_as_oracle"] = same_all
    out["pub_detail"] = detail  # (nested: for a reader; the flat scalars above are what the driver's record keeps — 24 keys with the 12 before them)
    return out


def _effective_cores():
    """Host threads this process may actually run at once: the smaller of the visible CPUs, the affinity mask and the cgroup's CPU
    quota (cpu.max = "quota period": the GPU boxes show 256 hardware threads and grant 16 CPUs — 256 oracle threads on such a quota
    run at HALF the rate of 16: the baseline is timed with what it is allowed to use, and says so)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def _cpu_model():
    """Model name of the host CPU (stated next to the core count), sockets x model when there are several."""
    try:
        names, phys = [], set()
        cur = None
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                cur = l.split(":", 1)[1].strip()
                names.append(cur)
            elif l.startswith("physical id"):
                phys.add(l.split(":", 1)[1].strip())
        if not names:
            return None
        return (f"{len(phys)} x " if len(phys) > 1 else "") + names[0] + f" ({len(names)} hardware threads)"
    except OSError:
        return None


def _child_json(cmd, timeout):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError(f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}")
    return json.loads(lines[-1])


def other_configs():
    """BASELINE.json configs 2 and 5 as child runs (each needs its own resident corpus: 3 GB and 61 GB).  Every child is this
    script on that workload: headline = the f32-corpus sweep, the library's default mirror sweep beside it — both come back."""
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-configs", "--batched", "0", "--callers", "0",
            "--legs", "i8", "--no-live-pmc", "--warmup", "3", "--rebuilds", "1"]
    runs = [("config2_1Mx768_cosine_top100", ["--rows", "1000000", "--steps", "200", "--rebuilds", "3"]),
            ("config5_10Mx1536_l2_top1000_mask1.0", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "12"]),
            ("config5_10Mx1536_l2_top1000_mask0.5", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "12",
                                                     "--mask", "0.5"]),
            ("config5_10Mx1536_l2_top1000_mask0.1", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "30",
                                                     "--mask", "0.1"])]
    out = {}
    for name, extra in runs:
        try:
            d = _child_json(base + extra, 300)
            r = d["roofline"]
            o = {"workload": d["config"]["workload"],
                 # the f32-corpus sweep, priced as SURVEY §8(d): kept rows x dim x 4 (+ the bitmap) per query
                 "f32_qps": d["value"], "f32_ms_per_step": d["ms_per_step"], "f32_avg_kernel_ms": r["avg_kernel_ms"],
                 "f32_frac": r["frac"], "f32_achieved_GBps": r["achieved"], "f32_kernel": r["kernel"], "f32_sweep_kind": r.get("sweep_kind"),
                 "f32_bytes_per_corpus_element": r["bytes_per_corpus_element"], "f32_step_frac": r["step_priced_as_survey_8d_frac"],
                 "f32_exact": d["parity"]["exact_topk_certified"] if d["parity"] else None}
            for tag in ("i8", "bf16"):  # the library's default sweep on this shape (the 8-bit mirror, else the bf16 one)
                if f"{tag}_mirror_queries_per_s" in r:
                    o.update({f"{tag}_qps": r[f"{tag}_mirror_queries_per_s"], f"{tag}_ms_per_step": r[f"{tag}_mirror_ms_per_step"],
                              f"{tag}_avg_kernel_ms": r[f"{tag}_mirror_avg_kernel_ms"],
                              f"{tag}_frac_on_mirror_bytes": r[f"{tag}_mirror_frac_on_mirror_bytes"],
                              f"{tag}_exact": r[f"{tag}_mirror_exact_and_same_answer"],
                              f"{tag}_candidates_rescored": d["mirror_legs"][tag]["candidates_rescored"]})
            out[name] = o
        except Exception as e:  # a child failing must not take the headline line down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def pmc_traffic(rows_per_gpu, args, elem_bytes=4):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if this
    exact workload was profiled (the fallback where rocprofv3 cannot be run around a child)."""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["entries"]
    except (OSError, ValueError, KeyError):
        return None, None
    for e in ent:
        if (e["rows_per_gpu"], e["dim"], e["metric"], e["nq"], e["mask"], e.get("bytes_per_corpus_element", 4)) == (
                rows_per_gpu, args.dim, args.metric, args.nq, args.mask, elem_bytes):
            return e["hbm_bytes_per_launch"], e["source"]
    return None, None


def live_pmc(args):
    """HBM bytes per launch of the nq=1 sweeps, measured NOW: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, no other trace
    domain — MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"), each a handful of sweeps of the same synthetic shard over
    the 8-bit mirror, the bf16 mirror and the f32 corpus.  Corrections as that guide prescribes: both counters are KiB; on
    gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read, so read bytes = FETCH_SIZE * 1024 * 2.
    Returns {"i8": {...}, "bf16": {...}, "f32": {...}} or None when rocprofv3 is not usable here."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--rows", str(args.rows), "--dim", str(args.dim),
             "--k", str(args.k), "--metric", args.metric]
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    tmp = tempfile.mkdtemp(prefix="nmn_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "p", "--"] + child,
                               capture_output=True, text=True, timeout=240, cwd="/tmp", env=env)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            rows = list(db.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)))
            for tag in ("i8", "bf16", "f32"):
                # scan_kernel<METRIC, MASKED, NQ, CHUNKS, .., .., HALF> / scan_i8_kernel<...>: the sweeps proper are the launches
                # within 2x of the largest (the f32 retry launches of a mirror pass share the f32 template and return at once)
                if tag == "i8":
                    v = [val for name, val in rows if "scan_i8_kernel" in name]
                else:
                    want = "true" if tag == "bf16" else "false"
                    v = [val for name, val in rows
                         if "scan_i8" not in name and (m := re.search(r"scan_kernel<[^>]*?(true|false)>", name)) and m.group(1) == want]
                    if tag == "f32":  # one unmasked query over the f32 rows of a large shard: the ring sweep (nmn_scan_ring.hip)
                        v += [val for name, val in rows if "scan_ring_kernel" in name]
                if not v:
                    continue
                big = [x for x in v if x * 2 >= max(v)]
                per.setdefault(tag, {})[counter] = (float(np.mean(big)), len(big))
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for tag, d in per.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        rd = d["FETCH_SIZE"][0] * 1024 * 2
        wr = d["WRITE_SIZE"][0] * 1024
        res[tag] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr,
                    "launches_counted": d["FETCH_SIZE"][1],
                    "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace and --pmc WRITE_SIZE "
                              "--kernel-trace (separate passes) around child runs of bench.py --pmc-child; "
                              "FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE*1024"}
    return res or None


def pmc_child(args):
    """What live_pmc() profiles: a few nq=1 sweeps over the 8-bit mirror, the bf16 mirror, the f32 corpus; no output."""
    import torch
    from neumann_amd import GpuFlatIndex
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx = GpuFlatIndex(args.dim, args.rows, device=0)
    idx.fill_synthetic(SEED_CORPUS, args.rows)
    q = torch.from_numpy(_synth(SEED_QUERY, 0, 4, args.dim)).to(dev)
    for mode in (1, 2, 0):
        idx.set_mirror(mode)
        for i in range(8):
            idx.search_device(q[i % 4:i % 4 + 1], args.k, METRICS[args.metric])
        torch.cuda.synchronize()
    idx.close()


def certificate(idx, q_host, metric, rows, scores, counts, world, dev, mask_host=None, reduce=None):
    """Size-independent proof that (rows, scores) is the exact top-k of the whole sharded corpus, using
    only the product's exact (reference-order) kernels, which tests/ pin bit-for-bit to the oracle:
      1. every returned score equals the exact score of its row (owner shard recomputes it);
      2. the list is ordered (score desc, row asc);
      3. #rows anywhere with exact score > s_k  ==  #returned scores > s_k, and the returned ties at
         s_k do not exceed the corpus-wide number of rows scoring exactly s_k.
    `idx` may be a list of shards held by this process (the one-process handle)."""
    shards = idx if isinstance(idx, (list, tuple)) else [idx]
    cnt = int(counts[0])
    r = rows[0, :cnt].astype(np.uint64)
    s = scores[0, :cnt]
    sk = float(s[-1]) if cnt else float("inf")
    gt = eq = bad = 0
    for sh in shards:
        base, n_local = sh.row_base, sh.rows
        mine = (r >= base) & (r < base + n_local)
        if mine.any():
            ex = sh.score_rows(q_host, (r[mine] - np.uint64(base)), metric)[0]
            bad += 0 if bool(np.all(ex == s[mine])) else 1
        if cnt:
            g, e = sh.count_exact(q_host, sk, metric, mask=mask_host)
            gt += g
            eq += e
    ordered = bool(np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (r[:-1] < r[1:])))) if cnt > 1 else True
    if world > 1:
        import torch
        import torch.distributed as dist
        agg = torch.tensor([gt, eq, bad], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(agg)
        gt, eq, bad = (int(x) for x in agg.tolist())
    n_gt_ret = int(np.sum(s > np.float32(sk)))
    n_eq_ret = int(np.sum(s == np.float32(sk)))
    exact = bad == 0 and ordered and gt == n_gt_ret and n_eq_ret <= eq and (cnt == 0 or n_eq_ret >= 1)
    return {"exact_topk_certified": bool(exact), "scores_bit_equal_exact_kernel": bad == 0, "ordered": ordered,
            "rows_above_kth": gt, "rows_equal_kth": eq, "returned": cnt, "recall_at_k": 1.0 if exact else None}


def measure_batched(args, idx, dev, metric, total_rows, torch, certify=True):
    """Config 3 on the resident corpus: nq queries per step through the MFMA sweep (one corpus sweep per 64 queries), two
    steps in flight; the last batch is certified query by query with the exact kernels."""
    from neumann_amd.sharded import ShardedSearcher
    nq = args.batched
    q_host = np.stack([_synth(SEED_QUERY + 1, s * nq, nq, args.dim) for s in range(4)])
    q_dev = torch.from_numpy(q_host).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    searchers = [ShardedSearcher(idx, world_size=1, rank=0, k=args.k, nq=nq, device=dev) for _ in range(2)]

    def step(i):
        with torch.cuda.stream(streams[i % 2]):
            return searchers[i % 2].search_device(q_dev[i % 4], metric)

    steps = max(6, min(args.steps, 16))
    idx.set_timing(2)  # HIP events around the sweep (its launches and the bound kernels between them), read back after the loop
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    for st_ in streams:
        idx.scan_history(st_)
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows, scores, counts = (t.cpu().numpy().copy() for t in out)
    sweep_ms = [x for st_ in streams for x in idx.scan_history(st_) if x > 0]
    st = idx.last_stats(streams[(steps - 1) % 2])
    elem_bytes = int(st.bytes_scanned // (st.rows_scanned * args.dim)) if st.rows_scanned else 4
    sweep_kind, sweep_launches = st.sweep, int(st.sweep_launches)
    idx.set_timing(False)
    torch.cuda.synchronize()
    qh = q_host[(steps - 1) % 4]
    ok = None
    if certify and not args.no_parity:
        ok = True
        for qi in (0, nq // 2, nq - 1):  # three of the batch's queries: the exact certificate is a full pass each
            c = certificate(idx, qh[qi], metric, rows.view(np.uint64)[qi:qi + 1], scores[qi:qi + 1], counts[qi:qi + 1], 1, dev)
            ok = ok and c["exact_topk_certified"]
    sweep = float(np.mean(sweep_ms)) if sweep_ms else float("nan")
    return {"value": nq * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "sweep_ms": sweep,
            "elem_bytes": elem_bytes, "certified": ok, "sweep_kind": sweep_kind, "sweep_launches": sweep_launches}


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY §8(f) rows, driver-visible: filtered SIMILAR end to end, IVF probe, upload, index-file load (child process)
# ---------------------------------------------------------------------------------------------------------------------
def next_rows_child(args):
    import tempfile
    import torch
    from neumann_amd import GpuFlatIndex
    from neumann_amd import columns as g
    from neumann_amd.ivf import GpuIvfFlat
    torch.cuda.set_device(0)
    out = {}
    metric = METRICS[args.metric]
    n, d, k = args.rows, args.dim, args.k
    # ---- (f2) filtered SIMILAR at selectivity 0.1 through the C ABI: predicate program -> bitmap -> masked sweep -> top-k ----
    try:
        with GpuFlatIndex(d, n, device=0) as idx, g.GpuColumns(n) as cols:
            idx.fill_synthetic(SEED_CORPUS, n)
            col = cols.add_column()
            bucket = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(10)
            cols.write(col, 0, np.full(n, g.CELL_INT, np.uint8), bucket)
            cols.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
            prog = [(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, col, 3, 0)]
            Q = _synth(SEED_QUERY + 7, 0, 8, d)
            rows, scores, counts, selected = idx.search_pred(cols, prog, [], Q[0], k, metric)
            _, _, _, st = idx.search(Q[0], k, metric, with_stats=True)
            eb = int(st.bytes_scanned // max(1, st.rows_scanned * d))
            reps = 40
            t0 = time.perf_counter()
            for i in range(reps):
                rows, scores, counts, selected = idx.search_pred(cols, prog, [], Q[i % 8], k, metric)
            ms = (time.perf_counter() - t0) / reps * 1e3
            keep = bucket == 3
            words = (n + 63) // 64
            padded = np.zeros(words * 64, dtype=bool)
            padded[:n] = keep
            mask = np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
            cert = certificate(idx, Q[(reps - 1) % 8], metric, rows, scores, counts, 1, None, mask)
            alg = selected * d * eb + n * 9 + n // 8  # kept rows of the sweep + (kind u8, payload u64) per row + the bitmap
            out["filtered_similar_sel0.1"] = {
                "what": f"nmn_index_search_pred (host buffers: query H2D, predicate kernel over {n} rows, masked sweep, select, "
                        f"rescore, result D2H) — search_with_pre_filter end to end (lib.rs:3514-3557), WHERE bucket = 3 of 10",
                "rows": n, "selected": int(selected), "selectivity": selected / n, "ms_per_query_wall": ms, "value": 1e3 / ms,
                "unit": "queries/s", "bytes_per_corpus_element": eb, "algorithmic_bytes": alg,
                "roofline": {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": alg / ms / 1e6 / HBM_PEAK_GBS, "note": "wall time of the whole call, PCIe and launches included"},
                "exact_topk_certified": cert["exact_topk_certified"]}
    except Exception as e:
        out["filtered_similar_sel0.1"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- (f1) upload and (f4) index-file load, GB/s end to end ----
    try:
        un = 500_000
        A = np.empty((un, d), dtype=np.float32)
        A[:] = (np.arange(d, dtype=np.float32) % 7 - 3.0)[None, :]
        A += (np.arange(un, dtype=np.float32) % 1013 * np.float32(1e-3))[:, None]
        with GpuFlatIndex(d, un, device=0) as idx:
            idx.upload(A[:1000])  # warm (allocations, first-touch)
            t0 = time.perf_counter()
            idx.upload(A, row0=0)
            dt = time.perf_counter() - t0
            out["upload_host_rows"] = {"what": "nmn_index_upload: pageable host rows -> HBM + magnitudes in reference order + bf16 mirror (one ingest kernel)",
                                       "rows": un, "bytes": A.nbytes, "seconds": dt, "value": A.nbytes / dt / 1e9, "unit": "GB/s",
                                       "bound": "PCIe (the copy); the ingest kernel behind it is HBM-bound"}
            path = os.path.join(tempfile.mkdtemp(prefix="nmn_bench_", dir="/tmp"), "shard.nmnidx")
            idx.save(path)
            fbytes = os.path.getsize(path)
        t0 = time.perf_counter()
        idx2 = GpuFlatIndex.load(path, device=0)
        dt = time.perf_counter() - t0
        ok = idx2.rows == un
        idx2.close()
        os.remove(path)
        out["index_file_load"] = {"what": "nmn_index_load: sequential read through pinned chunks, H2D, magnitudes recomputed and compared bit for bit, checksum",
                                  "rows": un, "file_bytes": fbytes, "seconds": dt, "value": fbytes / dt / 1e9, "unit": "GB/s",
                                  "rows_restored": bool(ok), "bound": "file system / PCIe"}
    except Exception as e:
        out["upload_and_load"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- (f4) IVF-Flat probe: 2M x 768, 256 lists, nprobe 8 ----
    try:
        rn, C_, nprobe, tn = 2_000_000, 256, 8, 200_000
        rng = np.random.default_rng(5)
        centers = rng.standard_normal((C_, d)).astype(np.float32) * np.float32(2.0)
        pool = rng.standard_normal((8192, d)).astype(np.float32)

        def rows_of(a, b):  # cluster centre + a noise vector from a pool + a per-row nudge (no two rows identical: exact ties make a
            i = np.arange(a, b, dtype=np.int64)  # probe re-fetch its list until the run of equal distances ends)
            r = centers[(i * 2654435761 >> 9) % C_] + pool[(i * 40503 + 17) % 8192]
            r[:, 0] += ((i % 100003) * np.float32(1e-5)).astype(np.float32)
            return r

        t0 = time.perf_counter()
        ivf = GpuIvfFlat.build(rows_of(0, tn), C_, nprobe=nprobe, max_iterations=3, seed=42, init_method="kmeans++", capacity_rows=rn)
        t_train = time.perf_counter() - t0
        with ivf:
            batches = [rows_of(a, min(a + 300_000, rn)) for a in range(tn, rn, 300_000)]  # (generated before the clock starts)
            t0 = time.perf_counter()
            for b in batches:
                ivf.add(b)
            t_add = time.perf_counter() - t0
            del batches
            Q = rows_of(12345, 12345 + 32) + np.float32(0.05)
            ivf.search(Q[0], k)
            reps = 64
            t0 = time.perf_counter()
            for i in range(reps):
                ids, dist, cnt = ivf.search(Q[i % 32], k)
            ms = (time.perf_counter() - t0) / reps * 1e3
            ivf.search(Q, k)                      # 32 queries in ONE call: chunks of 16 share the centroid phase and its round trip
            t0 = time.perf_counter()
            for i in range(4):
                bi, bd, bc = ivf.search(Q, k)
            ms_b = (time.perf_counter() - t0) / (4 * 32) * 1e3
            same = bool(np.array_equal(bi[(reps - 1) % 32], ids[0]) and np.array_equal(bd[(reps - 1) % 32], dist[0]))
            Q128 = rows_of(54321, 54321 + 128) + np.float32(0.05)   # 128 per call: two chunks of 64, each ONE batched pass per part
            ivf.search(Q128, k)
            t0 = time.perf_counter()
            for i in range(3):
                ivf.search(Q128, k)
            ms_b128 = (time.perf_counter() - t0) / (3 * 128) * 1e3
            sizes = ivf.cluster_sizes()
            ex_ids, ex_dist, _ = ivf.search(Q[(reps - 1) % 32], k, nprobe=C_)  # every list: the exhaustive answer
            recall = len(set(ids[0].tolist()) & set(ex_ids[0].tolist())) / float(k)
            probed = float(np.sort(sizes)[::-1][:nprobe].sum())  # upper bound on the rows one probe scans
            out["ivf_probe"] = {"what": "nmn_ivf_search (tensor_store/src/ivf.rs:325-406): rank 256 centroids, scan the 8 nearest lists, top-k",
                                "rows": rn, "dim": d, "clusters": C_, "nprobe": nprobe, "k": k, "ms_per_query_wall": ms, "value": 1e3 / ms,
                                "unit": "queries/s", "ms_per_query_wall_32_per_call": ms_b, "value_32_per_call": 1e3 / ms_b,
                                "ms_per_query_wall_128_per_call": ms_b128, "value_128_per_call": 1e3 / ms_b128,
                                "batched_call_equals_single_calls": same, "list_major_rows": int(ivf.list_major_rows), "train_seconds": t_train, "add_rows_per_s": (rn - tn) / t_add,
                                "list_size_min_mean_max": [int(sizes.min()), float(sizes.mean()), int(sizes.max())],
                                "rows_in_8_largest_lists": probed, "recall_vs_exhaustive_probe_one_query": recall,
                                "note": "a single-query call is launch-bound at this size (list scan ~50 us of the call); with many queries per call the centroid phase is shared and the list scans of a chunk run as one batched sweep with a bitmap per query; IVF itself is approximate in the reference too"}
    except Exception as e:
        out["ivf_probe"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# N GPUs driven by ONE process through the C ABI (nmn_sharded): what a Rust host holding one Arc<VectorEngine> binds
# ---------------------------------------------------------------------------------------------------------------------
def handle_child(args):
    from neumann_amd import GpuShardedIndex
    n = args.gpus
    pinned = os.environ.get("NMN_BENCH_DEVICE")
    devices = [int(pinned)] * n if pinned is not None else list(range(n))
    per_gpu = min(args.rows, 2_000_000)
    total = per_gpu * n
    metric = METRICS[args.metric]
    out = {"what": "nmn_sharded_*: ONE process, one shard per GPU, queries replicated, per-shard pipelines on per-device streams "
                   "(one host thread per shard), one grouped ncclAllGather of the packed top-k blocks, merge on device 0",
           "devices": devices, "rows_total": total}
    with GpuShardedIndex(args.dim, total, n, devices=devices) as sh:  # create ends with the collective's self-test
        sh.fill_synthetic(SEED_CORPUS, total)
        out["rccl_ranks"] = sh.rccl_ranks
        out["gather"] = {1: "rccl all-gather", 2: "peer copies"}.get(sh.gather_mode, str(sh.gather_mode))
        out["rows_per_gpu"] = [sh.shard_rows(g) for g in range(n)]
        Q = _synth(SEED_QUERY + 11, 0, 8, args.dim)
        sh.set_timing(True)
        for i in range(3):
            sh.search(Q[i], args.k, metric)
        reps = 30
        gm = []
        t0 = time.perf_counter()
        for i in range(reps):
            rows, scores, counts = sh.search(Q[i % 8], args.k, metric)
            gm.append(sh.last_gather_ms())
        dt = (time.perf_counter() - t0) / reps
        cert = certificate([sh.shard(g) for g in range(n)], Q[(reps - 1) % 8], metric, rows, scores, counts, 1, None)
        out.update({"value": 1.0 / dt, "unit": "queries/s (host-buffer API: query H2D and result D2H inside every call)",
                    "ms_per_query_wall": dt * 1e3, "gather_plus_merge_ms": float(np.median(gm)),
                    "exact_topk_certified": cert["exact_topk_certified"]})
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.next_rows_child:
        return next_rows_child(args)
    if args.handle_child:
        return handle_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a number "
                  f"for a GPU count that was not asked for", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # NMN_BENCH_DEVICE pins every rank to one device: lets the N>1 code path run on a 1-GPU box with NMN_BENCH_BACKEND=gloo
    # (RCCL refuses two ranks on one device); a debugging aid, never used by the driver
    pinned = os.environ.get("NMN_BENCH_DEVICE")
    dev_index = int(pinned) if pinned is not None else local_rank
    if dev_index >= torch.cuda.device_count():
        print(f"[bench] rank {rank}: device {dev_index} does not exist ({torch.cuda.device_count()} visible); --gpus {args.gpus} "
              f"needs one device per rank", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = None
    rccl_ranks = 0
    if world > 1 or args.always_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        backend = os.environ.get("NMN_BENCH_BACKEND", "nccl")  # "gloo": control-flow check of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # a REAL all-gather of rank ids before anything is measured: every rank must see 0..world-1 in order
        me = torch.tensor([rank], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        got = [torch.empty_like(me) for _ in range(world)]
        dist.all_gather(got, me)
        seen = [int(t.item()) for t in got]
        if seen != list(range(world)):
            print(f"[bench] rank {rank}: the all-gather of rank ids returned {seen}", file=sys.stderr)
            sys.exit(3)
        rccl_ranks = dist.get_world_size()

    from neumann_amd import GpuFlatIndex
    from neumann_amd.sharded import ShardedSearcher, shard_range

    metric = METRICS[args.metric]
    total_rows = args.rows * world if args.scaling == "weak" else args.rows
    r0, r1 = shard_range(total_rows, world, rank)
    local_rows = r1 - r0
    rows_per_gpu = [shard_range(total_rows, world, g)[1] - shard_range(total_rows, world, g)[0] for g in range(world)]

    n_query_sets = 16
    q_host = np.stack([_synth(SEED_QUERY, s * args.nq, args.nq, args.dim) for s in range(n_query_sets)])
    q_dev = torch.from_numpy(q_host).to(dev)  # [sets, nq, dim] resident in HBM
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    mask_host = mask_dev = None
    kept_rows = local_rows
    if args.mask < 1.0:
        # relational_engine-style selection bitmap (bit i of word i/64, LSB first), resident in HBM
        keep = np.random.default_rng(0x5EED0005 + rank).random(local_rows) < args.mask
        kept_rows = int(keep.sum())
        words = (local_rows + 63) // 64
        padded = np.zeros(words * 64, dtype=bool)
        padded[:local_rows] = keep
        mask_host = np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
        mask_dev = torch.from_numpy(mask_host.view(np.int64)).to(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state = {}

    def build_index():
        if state.get("idx") is not None:
            state["idx"].close()
        idx = GpuFlatIndex(args.dim, local_rows, row_base=r0, device=dev_index)
        t = time.perf_counter()
        idx.fill_synthetic(SEED_CORPUS, local_rows)
        torch.cuda.synchronize()
        state["fill_s"] = time.perf_counter() - t
        idx.set_mirror(args.mirror)
        state["idx"] = idx
        # one set of result buffers (and one library workspace) per stream
        state["searchers"] = [ShardedSearcher(idx, world_size=world, rank=rank, k=args.k, nq=args.nq, device=dev,
                                              always_gather=args.always_gather) for _ in range(n_streams)]
        return idx

    def step(i):
        with torch.cuda.stream(streams[i % n_streams]):
            return state["searchers"][i % n_streams].search_device(q_dev[i % n_query_sets], metric, mask_t=mask_dev)

    def timed_loop():
        """W untimed steps, EXACTLY K timed steps between fences; the elapsed time is the max over ranks.  The library's HIP
        events around the dominant kernel (recorded on the stream it is launched on) are on for the whole loop; the durations
        of the K timed steps are read back after the closing fence — the kernel time is OF the timed region, so it cannot
        exceed the step time (sweeps of a large shard never run side by side: nmn_index.h, sweep chain)."""
        idx = state["idx"]
        idx.set_timing(2)  # the two events around the sweep, nothing else
        for i in range(args.warmup):
            step(i)
        fence()
        for st_ in streams:
            idx.scan_history(st_)  # drop the warm-up's entries
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        fence()
        elapsed = time.perf_counter() - t0
        scan_ms = [x for st_ in streams for x in idx.scan_history(st_) if x > 0]
        st = idx.last_stats(streams[(args.steps - 1) % n_streams])
        eb = int(st.bytes_scanned // (st.rows_scanned * args.dim)) if st.rows_scanned else 4
        idx.set_timing(False)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        info = {"scan_ms": scan_ms, "elem_bytes": eb, "candidates": int(st.candidates_rescored),
                "sweep": st.sweep, "sweep_launches": int(st.sweep_launches)}
        return elapsed, tuple(t.cpu().numpy().copy() for t in out), info  # result of the last timed step

    def isolated_kernel_ms(n=8):
        """The same kernel with nothing else on the device (one step at a time, waited for): what the timed loop's figure is
        compared with.  Not used for `roofline.achieved`."""
        idx = state["idx"]
        idx.set_timing(True)
        out = []
        for i in range(n):
            torch.cuda.synchronize()
            step(i)
            st = idx.last_stats(streams[i % n_streams])
            if st.scan_ms > 0:
                out.append(st.scan_ms)
        idx.set_timing(False)
        for st_ in streams:
            idx.scan_history(st_)
        fence()
        return float(np.mean(out)) if out else None

    default_workload = (args.rows == 10_000_000 and args.dim == 768 and args.k == 100 and args.nq == 1 and
                        args.metric == "cosine" and args.mask >= 1.0 and args.mirror == 0)
    extras = world == 1 and not args.always_gather
    leg_modes = [] if (args.no_mirror_legs or args.mirror != 0 or args.k > 4096 or args.nq != 1) else \
        [m for m, nm in ((1, "i8"), (2, "bf16")) if nm in args.legs.split(",")]

    # ---- the headline: median over index rebuilds --------------------------------------------------------------------
    draws, batched_draws, infos = [], {}, []
    last_out = None
    do_batched = extras and args.batched > 0 and args.nq == 1 and args.mask >= 1.0 and args.k <= 4096
    for b in range(max(1, args.rebuilds)):
        build_index()
        elapsed, last_out, info = timed_loop()
        draws.append(elapsed)
        infos.append(info)
        if b == 0:
            hbm0 = state["idx"].hbm_bytes()  # what the shard holds for the headline loop (before a leg below builds a mirror)
        if do_batched:
            # config 3 on the SAME resident index: first on the sweep the headline uses, then (headline = f32 corpus) on the
            # library's default — the smallest mirror that serves the batch, built on first use
            for mode in ([args.mirror] + ([1] if args.mirror == 0 and not args.no_mirror_legs else [])):
                state["idx"].set_mirror(mode)
                batched_draws.setdefault(mode, []).append(
                    measure_batched(args, state["idx"], dev, metric, total_rows, torch, certify=(b == max(1, args.rebuilds) - 1)))
            state["idx"].set_mirror(args.mirror)
    idx = state["idx"]
    med = int(np.argsort(draws)[len(draws) // 2])  # the median draw: `value`, `ms_per_step` AND the kernel time come from this one loop
    elapsed = float(draws[med])
    ms_per_step = elapsed / args.steps * 1e3
    value = args.nq * args.steps / elapsed
    scan_ms, elem_bytes, cands = infos[med]["scan_ms"], infos[med]["elem_bytes"], [i["candidates"] for i in infos]
    scan_avg = float(np.mean(scan_ms)) if scan_ms else float("nan")
    hbm_bytes_per_row = (hbm0[0] + hbm0[1] + hbm0[2]) / max(local_rows, 1)
    scan_alone = isolated_kernel_ms() if world == 1 else None
    # corpus sweeps per step, from the sweep the LIBRARY says it ran (nmn_search_stats.sweep_kind): the matrix-core sweeps serve a
    # whole pass of queries per corpus read, the VALU sweeps 4 (2 on the 8-bit mirror), the ring sweep and the large-k scan one
    def passes_of(sweep):
        per = QUERIES_PER_SWEEP.get(sweep)
        return 1 if per is None else (args.nq + per - 1) // per

    sweep_name = infos[med]["sweep"]
    passes = passes_of(sweep_name)
    if world > 1:
        leg_modes = [m for m in leg_modes if m == 1]  # N > 1: the library's default sweep only (a second mirror per rank buys nothing)

    def alg_bytes_for(eb, sweep=None):  # excluded rows are never read
        return (kept_rows * args.dim * eb + (local_rows // 8 if mask_dev is not None else 0)) * passes_of(sweep or sweep_name)

    alg_bytes = alg_bytes_for(elem_bytes)
    achieved = alg_bytes / (scan_avg * 1e-3) / 1e9 if scan_ms else float("nan")
    kernel_name = KERNEL_OF_SWEEP.get(sweep_name, sweep_name)

    # ---- read ceiling of this device: the scan's access pattern with the arithmetic removed ----
    read_ceiling = idx.read_probe(3) if local_rows else None

    # ---- parity of the last result ----
    parity = None
    q_last = q_host[(args.steps - 1) % n_query_sets][0]
    if not args.no_parity:
        o_rows, o_scores, o_counts = last_out
        parity = certificate(idx, q_last, metric, o_rows.view(np.uint64), o_scores, o_counts, world, dev, mask_host)

    # ---- N > 1: what the collective step costs, measured on its own ----
    multi = None
    if world > 1 or args.always_gather:
        s0 = state["searchers"][0]
        gm = []
        for i in range(12):
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[0]):
                e0.record()
                s0.gather_merge()
                e1.record()
            torch.cuda.synchronize()
            gm.append(e0.elapsed_time(e1))
        multi = {"rccl_ranks": rccl_ranks if backend == "nccl" else 0, "ranks": rccl_ranks,
                 "collective_backend": "nccl (RCCL over xGMI)" if backend == "nccl" else backend,
                 "rank_id_all_gather": "ranks 0..N-1 seen in order on every rank before the measurement",
                 "rows_per_gpu": rows_per_gpu, "gather_plus_merge_ms": float(np.median(gm[2:])),
                 "gather_bytes_per_rank": int(state["searchers"][0]._bufs["size"]), "host": "one process per GPU (torch.distributed)",
                 # how to read `value` across N: a query scans the WHOLE corpus, every GPU its shard.  Under weak scaling the corpus
                 # grows with N (10M rows per GPU), so queries/s staying level IS linear scaling — the quantity that grows with N is
                 # the shard scans (one GPU, one query, its rows) the job completes per second:
                 "shard_scans_per_s": None, "corpus_rows_scanned_per_s": None,
                 "reading": ("weak scaling: rows_total = N x rows_per_gpu; `value` = queries/s over all rows_total rows — staying level from "
                             "N = 1 up is 100 % efficiency; shard_scans_per_s = N x value is the aggregate that grows with N"
                             if args.scaling == "weak" else
                             "strong scaling: rows_total is fixed, every GPU scans 1/N of it; value(N) ~ N x value(1) is 100 % efficiency")}

    def per_rank(v):
        """[v of rank 0, ..., v of rank N-1] (an integer per rank) — which sweep every shard used must be visible in the line"""
        if world == 1:
            return [int(v)]
        t = torch.tensor([int(v)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [int(x.item()) for x in got]

    headline_eb = per_rank(elem_bytes)
    kinds = list(KERNEL_OF_SWEEP)  # (an integer per rank travels; the names come back)
    headline_kinds = [kinds[i] if 0 <= i < len(kinds) else "unknown" for i in per_rank(kinds.index(sweep_name) if sweep_name in kinds else -1)]

    # ---- the same loop on the shard's MIRRORS: the library's default sweep (1 B per element where the shape allows it, every
    # candidate re-scored from the f32 rows: the same answer, bit for bit) and the bf16 mirror.  Same index, queries, streams,
    # steps, warmup; at N > 1 every rank runs them (timed_loop's collectives), and which ranks got their mirror is reported ----
    legs = {}
    mirror_eb = None
    for mode in leg_modes:
        idx.set_mirror(mode)
        e2, out2, info2 = timed_loop()
        idx.set_mirror(args.mirror)
        scan2, eb2 = info2["scan_ms"], info2["elem_bytes"]
        ebs = per_rank(eb2)
        if mode == 1:
            mirror_eb = ebs
        # (every decision below is taken from the gathered lists, identical on all ranks: the certificate further down is a collective,
        #  and a rank whose mirror did not fit must not leave its neighbours waiting in it)
        name = SWEEP[min(ebs)][0]
        if name in legs or ebs == headline_eb:
            continue  # (a shape the 8-bit sweeps do not serve: mode 1 already was the bf16 mirror; or no mirror fitted anywhere)
        k_ms = float(np.mean(scan2)) if scan2 else float("nan")
        b2 = alg_bytes_for(eb2, info2["sweep"])
        ach = b2 / (k_ms * 1e-3) / 1e9 if scan2 else float("nan")
        cert2 = None
        if not args.no_parity:
            cert2 = certificate(idx, q_last, metric, out2[0].view(np.uint64), out2[1], out2[2], world, dev, mask_host)
        same = bool(np.array_equal(out2[0], last_out[0]) and np.array_equal(out2[1].view(np.uint32), last_out[1].view(np.uint32)))
        legs[name] = {"what": f"nmn_index_set_mirror({mode}): the sweep streams the " + SWEEP[eb2][1] +
                              "; same index, queries, streams, steps and warmup as the headline loop",
                      "queries_per_s": args.nq * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3,
                      "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": ach / HBM_PEAK_GBS if scan2 else None,
                      "avg_kernel_ms": k_ms, "kernel_launches_timed": len(scan2),
                      "algorithmic_bytes_per_launch": b2, "bytes_per_corpus_element": eb2, "bytes_per_corpus_element_by_rank": ebs,
                      "pricing": SWEEP[eb2][2], "kernel": KERNEL_OF_SWEEP.get(info2["sweep"], info2["sweep"]), "sweep_kind": info2["sweep"],
                      "frac_of_read_ceiling": (ach / read_ceiling) if (scan2 and read_ceiling) else None,
                      "candidates_rescored": info2["candidates"],
                      "traffic": pmc_traffic(local_rows, args, eb2)[0], "traffic_source": pmc_traffic(local_rows, args, eb2)[1],
                      "exact_topk_certified": cert2["exact_topk_certified"] if cert2 else None,
                      "same_answer_as_headline_sweep": same}
    hbm1 = idx.hbm_bytes()

    def batched_summary(draws_, mode):
        vals = [b["value"] for b in draws_]
        sw = [b["sweep_ms"] for b in draws_]
        eb3 = draws_[-1]["elem_bytes"]
        nq = args.batched
        per_sweep = 128 if ((args.dim // 128 <= 6 or args.dim // 128 in (8, 10)) and nq > 64) else 32 if args.dim // 128 in (16, 24, 32) else 64
        sweep_med = float(np.median(sw))
        gbps = idx.rows * args.dim * eb3 / (sweep_med * 1e-3) / 1e9
        planes, peak, pname = (2, MFMA_I8_PEAK_TOPS, "mfma_i8") if eb3 == 1 else (1, MFMA_BF16_PEAK_TFLOPS, "mfma_bf16")
        ops = planes * 2.0 * idx.rows * args.dim * nq / (sweep_med * 1e-3) / 1e12
        return {"workload": f"{total_rows}x{args.dim} f32 {args.metric} TOP-{args.k}, nq={nq}/step", "mirror_mode": mode,
                "sweep": SWEEP[eb3][1], "value": float(np.median(vals)), "unit": "queries/s", "rebuilds": vals,
                "ms_per_step": float(np.median([b["ms_per_step"] for b in draws_])), "steps": draws_[-1]["steps"],
                "sweep_ms_incl_sampling_pass": sweep_med, "sweep_ms_rebuilds": sw, "query_blocks_per_launch": (nq + per_sweep - 1) // per_sweep,
                # config 3 is reported against BOTH bounds.  HBM: the streamed matrix's bytes once per 64-128 queries
                # (f32 corpus: rows*dim*4 — §8(d)'s own figure).  Matrix cores: 2*rows*dim*nq operations per query plane (the 8-bit sweep
                # multiplies two int8 planes of every query: twice the operations of the bf16 form) against the dense peak of the type.
                "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                             "kernel": f"{KERNEL_OF_SWEEP.get(draws_[-1]['sweep_kind'], draws_[-1]['sweep_kind'])} (every launch of the sweep)",
                             "sweep_kind": draws_[-1]["sweep_kind"], "sweep_launches": draws_[-1]["sweep_launches"], "bytes_per_corpus_element": eb3,
                             "frac_priced_as_survey_8d": gbps * 4 / eb3 / HBM_PEAK_GBS,
                             "mfma": {"bound": pname, "planes_per_query": planes, "achieved_tops": ops, "peak_tops": peak,
                                      "unit": "TOP/s" if eb3 == 1 else "TFLOP/s", "frac": ops / peak}},
                "exact_topk_certified_3_of_batch": draws_[-1]["certified"]}

    batched = {("f32_corpus" if mode == 0 else "mirror"): batched_summary(d_, mode) for mode, d_ in batched_draws.items()} or None

    # Single-query calls from many host threads at once (the reference's Arc<VectorEngine> under concurrent clients):
    # the C ABI merges callers that arrive while the shard is busy into one query batch.  Native threads, 1 second.  On the
    # library's default (mirror) configuration — this leg is about the request coalescer, not about a sweep.
    callers = None
    if extras and args.callers > 0 and args.nq == 1 and args.mask >= 1.0 and args.k <= 4096 and default_workload:
        idx.set_mirror(1)
        cq = _synth(SEED_QUERY + 2, 0, args.callers, args.dim)
        r = idx.callers_probe(cq, args.k, metric, seconds=1.0)
        callers = {"workload": f"{args.callers} host threads, each nmn_index_search(nq=1, k={args.k}) in a loop, "
                               f"{total_rows}x{args.dim} f32 {args.metric}, library default (mirror) sweep",
                   "value": r["calls_per_s"], "unit": "queries/s", "threads": args.callers,
                   "sweeps_carrying_2_or_more_calls": r["merged_batches"], "calls_in_them": r["merged_calls"],
                   "answers_differing_from_a_lone_call": r["mismatches"]}
        cq2 = _synth(SEED_QUERY + 3, 0, 2 * args.callers, args.dim)  # twice the threads: one sweep carries up to 128 callers
        r2 = idx.callers_probe(cq2, args.k, metric, seconds=1.0)
        callers["with_twice_the_threads"] = {"threads": 2 * args.callers, "value": r2["calls_per_s"], "unit": "queries/s",
                                             "answers_differing_from_a_lone_call": r2["mismatches"]}
        idx.set_mirror(args.mirror)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ga = None
        if args.mask >= 1.0 and args.nq == 1:  # the resident index's answer to one of the baseline's queries, for the checker
            ga = (1,) + tuple(idx.search(q_host[1][0], args.k, metric))
        cpu = cpu_baseline(args, metric, total_rows, dev_index, ga)

    traffic, traffic_src = pmc_traffic(local_rows, args, elem_bytes)
    traffic_rw = None
    # HBM traffic of the dominant kernel from PMC counters collected in THIS run (the committed profiles/pmc_traffic.json
    # figure above is the fallback where rocprofv3 is not usable)
    if rank == 0 and extras and default_workload and not args.no_live_pmc:
        live = live_pmc(args)
        if live:
            key = SWEEP[elem_bytes][0]
            if key in live:
                traffic, traffic_src = live[key]["hbm_bytes_per_launch"], live[key]["source"]
                traffic_rw = [live[key]["read_bytes_corrected"], live[key]["write_bytes"]]
            for name in legs:
                if name in live:
                    legs[name]["traffic"], legs[name]["traffic_source"] = live[name]["hbm_bytes_per_launch"], live[name]["source"]
                    legs[name]["traffic_read_write"] = [live[name]["read_bytes_corrected"], live[name]["write_bytes"]]
    fill_s = state["fill_s"]
    idx.close()  # the children (and the one-process handle) need the HBM
    state["idx"] = None
    others = next_rows = None
    if extras and default_workload and not args.no_other_configs:
        others = other_configs()
        try:
            next_rows = _child_json([sys.executable, os.path.abspath(__file__), "--next-rows-child"], 240)
        except Exception as e:
            next_rows = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 or args.always_gather:
        dist.barrier()  # every rank is done with its collectives before any of them tears the group down
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.no_handle_leg and multi is not None:
        # the same GPUs, ONE process, through the C ABI's nmn_sharded handle (its create runs the rank all-gather self-test)
        try:
            time.sleep(1.0)  # the other ranks are exiting and returning their HBM
            multi["one_process_handle"] = _child_json(
                [sys.executable, os.path.abspath(__file__), "--handle-child", "--gpus", str(world), "--rows", str(args.rows),
                 "--dim", str(args.dim), "--k", str(args.k), "--metric", args.metric], 240)
        except Exception as e:
            multi["one_process_handle"] = {"error": f"{type(e).__name__}: {e}"}
    # What `value` counts: queries per second against the WHOLE corpus, at every N (BASELINE.json's metric).  One step = one
    # batch of queries; every GPU scans its shard of all `rows_total` rows.  Under WEAK scaling (config 4) the corpus grows with
    # N — `--rows` rows per GPU — so `value` staying level from N = 1 up IS linear scaling; the quantity that grows with N, shard
    # scans per second (N x value), is reported under `multi_gpu` only.
    if rank == 0 and multi is not None and "reading" in multi:
        multi["shard_scans_per_s"] = value * world
        multi["corpus_rows_scanned_per_s"] = value * total_rows
        multi["headline_bytes_per_corpus_element_by_rank"] = headline_eb
        multi["mirror_leg_bytes_per_corpus_element_by_rank"] = mirror_eb
        multi["shards_with_mirror"] = None if mirror_eb is None else sum(1 for x in mirror_eb if x < 4)
    if rank == 0:
        sweep_key, sweep_txt, pricing = SWEEP[elem_bytes]
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if scan_ms else None,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_read_write": traffic_rw,
                "kernel": kernel_name, "avg_kernel_ms": scan_avg, "kernel_launches_timed": len(scan_ms),
                "avg_kernel_ms_from": "HIP events recorded by the library on the launch stream around the kernel, in EVERY step of the "
                                      "timed loop `value` comes from (nmn_index_scan_history); sweeps of one shard never overlap "
                                      "(sweep chain), so kernel <= step",
                "avg_kernel_ms_alone": scan_alone,
                "pricing": "bytes the kernel is asked to read: " + pricing,
                "bytes_per_corpus_element": elem_bytes,
                "algorithmic_bytes_per_launch": alg_bytes,
                # the same kernel time priced as SURVEY §8(d) writes it (N*d*4 per query): for the f32 headline the same number
                "achieved_priced_as_survey_8d": achieved * 4 / elem_bytes if scan_ms else None,
                "frac_priced_as_survey_8d": achieved * 4 / elem_bytes / HBM_PEAK_GBS if scan_ms else None,
                "step_priced_as_survey_8d_frac": (kept_rows * args.dim * 4 * passes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "candidates_rescored": int(np.median(cands)) if cands else None,
                # what the library says it ran (nmn_search_stats.sweep_kind / sweep_launches), not bench.py's guess
                "sweep_kind": sweep_name, "sweep_launches": infos[med]["sweep_launches"],
                # measured in this run by nmn_index_read_probe: the sweep's own data movement (the ring kernel's LDS-DMA pieces, same
                # workgroups and stages) with the arithmetic and every store removed — GB/s; the sweep cannot read faster than this
                "ring_only_read_ceiling": read_ceiling,
                "frac_of_read_ceiling": (achieved / read_ceiling) if (scan_ms and read_ceiling) else None}
        # ---- everything else a checker needs, as FLAT scalars (the driver's record keeps scalars of `roofline` / `config` /
        # `cpu_baseline` and drops nested objects): the mirror sweeps of the same loop; configs 3, 2, 5 on BOTH sweeps.
        # "<x>_frac" prices the f32 sweeps as SURVEY §8(d) does (rows*dim*4 per query or batch); "<x>_frac_on_mirror_bytes" prices a
        # mirror sweep on the bytes IT is asked to read (rows*dim*1 or *2) — an exact acceleration, never a §8(d) figure ----
        for name, leg in legs.items():
            pfx = f"{name}_mirror_"
            roof[pfx + "queries_per_s"] = leg["queries_per_s"]
            roof[pfx + "ms_per_step"] = leg["ms_per_step"]
            roof[pfx + "avg_kernel_ms"] = leg["avg_kernel_ms"]
            roof[pfx + "frac_on_mirror_bytes"] = leg["frac"]
            roof[pfx + "traffic"] = leg["traffic"]
            roof[pfx + "exact_and_same_answer"] = bool(leg["exact_topk_certified"] and leg["same_answer_as_headline_sweep"]) \
                if leg["exact_topk_certified"] is not None else None
        cname = "c3" if (args.rows == 10_000_000 and args.dim == 768 and args.batched == 64 and args.metric == "cosine" and args.k == 100) \
            else f"batch{args.batched}"
        for key, bsum in (batched or {}).items():
            tag = SWEEP[bsum["roofline"]["bytes_per_corpus_element"]][0]
            pfx = f"{cname}_{tag}_"
            roof[pfx + "qps"] = bsum["value"]
            roof[pfx + "ms_per_batch"] = bsum["ms_per_step"]
            roof[pfx + "sweep_ms"] = bsum["sweep_ms_incl_sampling_pass"]
            roof[pfx + ("frac" if tag == "f32" else "frac_on_mirror_bytes")] = bsum["roofline"]["frac"]
            roof[pfx + "mfma_frac"] = bsum["roofline"]["mfma"]["frac"]
            roof[pfx + "exact"] = bsum["exact_topk_certified_3_of_batch"]
            roof[pfx + "sweep_launches"] = bsum["roofline"]["sweep_launches"]
        for cfg_name, o in (others or {}).items():
            pfx = cfg_name.split("_")[0].replace("config", "c") + ("_mask" + cfg_name.rsplit("mask", 1)[1] if "mask" in cfg_name else "") + "_"
            if "error" in o:
                roof[pfx + "error"] = o["error"][:100]
                continue
            for k_, v_ in o.items():
                if isinstance(v_, (int, float, bool)) or v_ is None:
                    roof[pfx + k_] = v_
        roof, notes = order_roofline(roof)
        line = {
            "metric": "queries/sec, brute-force SIMILAR TOP-K (recall@K = 1.0 vs CPU oracle)",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_counts": "queries/s over the whole corpus (rows_total rows), at every N",
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": f"{sweep_key} sweep + f32 exact rescore (results bit-equal to the f32 reference path)" if sweep_key != "f32" else "f32",
            "data": "synthetic",
            "rebuilds": {"n": len(draws), "queries_per_s": [args.nq * args.steps / e for e in draws],
                         "spread": (max(draws) - min(draws)) / elapsed if len(draws) > 1 else 0.0,
                         "value_is": "the median draw; every draw is W warmup + exactly K timed steps on a freshly built index"},
            "config": {"workload": f"{total_rows}x{args.dim} f32 {args.metric} TOP-{args.k}, nq={args.nq}/step"
                                   + (f", WHERE mask selectivity {args.mask}" if args.mask < 1.0 else ""),
                       "rows_total": total_rows, "rows_per_gpu": local_rows, "dim": args.dim, "k": args.k,
                       "nq": args.nq, "streams": n_streams,
                       "sweep": sweep_txt + ("; every candidate re-scored from the f32 corpus in the reference's order" if elem_bytes < 4 else
                                             " (row-major, resident in HBM), approximate score per row; candidates re-scored in the reference's order"),
                       "bytes_per_corpus_element": elem_bytes,
                       "shards": world, "shards_on_f32_sweep_in_headline": sum(1 for x in headline_eb if x == 4),
                       "shards_with_mirror": None if mirror_eb is None else sum(1 for x in mirror_eb if x < 4),
                       "hbm_bytes_per_row": hbm_bytes_per_row, "hbm_bytes_per_element": hbm_bytes_per_row / args.dim,
                       "hbm_bytes_per_element_after_legs": (hbm1[0] + hbm1[1] + hbm1[2]) / max(local_rows, 1) / args.dim,
                       "parallelism": f"row-range shards x{world}, RCCL all-gather of top-k",
                       # flat scalars a SCALE_r*.json can be checked from (the driver keeps the scalars of `config`): the sweep every
                       # rank's library reported, the ranks a real RCCL all-gather saw, what the collective + merge cost per step
                       "sweep_kind": sweep_name, "sweep_kind_by_rank": ",".join(headline_kinds),
                       "rccl_ranks": (multi or {}).get("rccl_ranks"), "gather_plus_merge_ms": (multi or {}).get("gather_plus_merge_ms"),
                       # step-level (whole call, tail included) fractions of the other configs, priced as SURVEY §8(d)
                       "c2_f32_step_frac": roof.get("c2_f32_step_frac"), "c5_mask0.1_f32_step_frac": roof.get("c5_mask0.1_f32_step_frac"),
                       "filtered_similar_sel0.1_ms": ((next_rows or {}).get("filtered_similar_sel0.1") or {}).get("ms_per_query_wall"),
                       "c3_f32_sweep_launches": roof.get("c3_f32_sweep_launches")},
            "roofline": roof,
            "notes": notes,
            "cpu_baseline": cpu,
            "parity": parity,
            "mirror_legs": legs or None,
            "multi_gpu": multi,
            "batched": batched,
            "concurrent_callers": callers,
            "other_configs": others,
            "next_rows": next_rows,
            "fill_s": fill_s,
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
