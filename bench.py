#!/usr/bin/env python
"""bench.py — SIMILAR TOP-K throughput of the MI355X-native vector_engine hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): 10M x 768 f32 cosine brute-force TOP-100.  One "step" = one pass
of the hot path over one batch of `--nq` synthetic queries (default 1: the HBM-bound single-query
scan the roofline target is quoted on).  The corpus and the queries are resident in HBM before the
timed region.  With N > 1 the 10M-row corpus is row-range sharded over the N GPUs (one process per
GPU); every step each rank scans its shard, the per-shard top-k blocks are all-gathered over RCCL
and merged on-device (strong scaling: total work per query is fixed).  `--scaling weak` instead
keeps 10M rows PER GPU (config 4: 80M rows on 8 GPUs).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel of the headline configuration (scan_kernel over the shard's bf16 mirror: 2 bytes per corpus
                element; the exact f32 rescore keeps the answer bit-equal).  `achieved` / `frac` count the bytes that kernel is
                asked to read (rows*dim*2, `pricing` says so, `bytes_per_corpus_element` = 2); SURVEY.md §8(d) prices a query
                at rows*dim*4 (the f32 corpus): `achieved_priced_as_survey_8d` / `frac_priced_as_survey_8d` give the same
                kernel time under that pricing (an EFFECTIVE rate; it exceeds the HBM peak because half the bytes are moved)
  roofline_f32_corpus  (default single-GPU run) the same index, same queries, same 2-stream loop with
                nmn_index_set_mirror(0): the sweep of the row-major f32 corpus §8(d) describes, priced at rows*dim*4 —
                queries/s, HIP-event kernel average, achieved GB/s, frac of 8 TB/s, exactness certificate
  cpu_baseline  the CPU oracle (oracle/nmn_oracle.c, -O3 -march=native, all host cores) on a bounded
                row sample of the same workload, extrapolated linearly to the full row count
  parity        the GPU result of the last timed query checked against the oracle / exact certificate
  batched       (default single-GPU run) config 3: 64 queries per step on the MFMA sweep, same resident corpus
  concurrent_callers  (default single-GPU run) 64 native host threads, each a loop of single-query nmn_index_search calls on
                the same resident corpus: the C ABI merges callers that arrive while the shard is busy into one batch
  other_configs (default single-GPU run) config 2 (1M x 768) and config 5 (10M x 1536 L2 TOP-1000, mask 1.0 / 0.5 /
                0.1), each as a child run of this script with its own corpus, each with its exactness certificate
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable by a copy)
SEED_CORPUS = 0x5EED0003
SEED_QUERY = 0x5EED0002
METRICS = {"cosine": 0, "euclidean": 1, "dot": 2}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total corpus rows (strong) / rows per GPU (weak)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--nq", type=int, default=1, help="queries per step")
    ap.add_argument("--metric", default="cosine", choices=sorted(METRICS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are pipelined over (each stream runs whole steps in order; with 2 "
                         "the select/rescore/gather tail of one query overlaps the next query's scan)")
    ap.add_argument("--mask", type=float, default=1.0,
                    help="selectivity of a synthetic WHERE-predicate bitmap (config 5); 1.0 = no mask")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--callers", type=int, default=64, help="host threads of the concurrent-callers leg (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline leg")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra legs of the default single-GPU run: BASELINE.json's other single-GPU "
                         "configurations (config 2: 1M x 768 cosine TOP-100; config 5: 10M x 1536 L2 TOP-1000 with mask "
                         "1.0 / 0.5 / 0.1) measured as child runs of this script and reported under \"other_configs\"")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (then it comes from profiles/pmc_traffic.json)")
    ap.add_argument("--no-f32-leg", action="store_true",
                    help="skip the roofline_f32_corpus leg (the same loop with the bf16 mirror switched off)")
    ap.add_argument("--always-gather", action="store_true",
                    help="run the RCCL all-gather + device merge even with one rank (measures what the N>1 step adds "
                         "on a 1-GPU box; the group has one member)")
    ap.add_argument("--batched", type=int, default=64,
                    help="also measure config 3 (this many queries per step on the MFMA sweep) after the main "
                         "measurement and report it under \"batched\" (single-GPU runs only; 0 = skip)")
    return ap.parse_args()


def cpu_baseline(args, metric, total_rows, device):
    """Time the CPU oracle on a bounded row sample (and, as the checker, compare the GPU path with it
    on that same sample); returns the cpu_baseline object."""
    from oracle import oracle_c as oc
    from neumann_amd import GpuFlatIndex
    cores = os.cpu_count() or 1
    sample_rows = min(1_000_000, total_rows)  # 3 GB at dim 768: enough rows per thread on a 256-thread host
    A = oc.synth(SEED_CORPUS, 0, sample_rows, args.dim, nthreads=cores)
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
    dt1 = (time.perf_counter() - t0) * (sample_rows / float(n1))
    # checker: the HIP path on the same sample rows must return the oracle's rows and scores
    with GpuFlatIndex(args.dim, sample_rows, row_base=0, device=device) as small:
        small.fill_synthetic(SEED_CORPUS, sample_rows)
        gr, gs, gc = small.search(Q[1], args.k, metric)
    sample_ok = bool(gc[0] == er.size and np.array_equal(gr[0, :er.size], er) and np.all(gs[0, :er.size] == es))
    qps_full = 1.0 / (dt * (total_rows / sample_rows))
    return {
        "value": qps_full, "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"{reps} queries x {sample_rows} rows x {args.dim} (same generator/seed), "
                  f"{dt * 1e3:.2f} ms/query on {cores} threads, extrapolated linearly to {total_rows} rows; "
                  f"1 thread: {dt1 * 1e3:.1f} ms per {sample_rows} rows. Optimistic for the reference "
                  f"(flat array, per-thread partial top-k; the Rust path also pays a BTreeMap lookup, two clones "
                  f"per row and a full sort, published 193-367 ns/row)",
        "gbps": sample_rows * args.dim * 4 / dt / 1e9,
        "gpu_matches_oracle_on_sample": sample_ok,
    }


def other_configs():
    """BASELINE.json configs 2 and 5 as child runs (each needs its own resident corpus: 3 GB and 61 GB)."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-configs", "--batched", "0", "--callers", "0",
            "--no-f32-leg", "--no-live-pmc", "--warmup", "3"]
    runs = [("config2_1Mx768_cosine_top100", ["--rows", "1000000", "--steps", "200"]),
            ("config5_10Mx1536_l2_top1000_mask1.0", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "12"]),
            ("config5_10Mx1536_l2_top1000_mask0.5", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "12",
                                                     "--mask", "0.5"]),
            ("config5_10Mx1536_l2_top1000_mask0.1", ["--dim", "1536", "--metric", "euclidean", "--k", "1000", "--steps", "30",
                                                     "--mask", "0.1"])]
    out = {}
    for name, extra in runs:
        try:
            r = subprocess.run(base + extra, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            out[name] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                         "ms_per_step": d["ms_per_step"], "roofline_frac": d["roofline"]["frac"],
                         "achieved_GBps": d["roofline"]["achieved"], "kernel": d["roofline"]["kernel"],
                         "exact_topk_certified": d["parity"]["exact_topk_certified"] if d["parity"] else None}
        except Exception as e:  # a child failing must not take the headline line down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def pmc_traffic(rows_per_gpu, args, elem_bytes=4):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if this
    exact workload was profiled; bench.py cannot collect PMC counters itself (they need a rocprofv3 wrapper)."""
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
    """HBM bytes per launch of the two nq=1 sweeps, measured NOW: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, no other trace
    domain — MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"), each a handful of sweeps of the same synthetic shard over
    the bf16 mirror and over the f32 corpus.  Corrections as that guide prescribes: both counters are KiB; on gfx950
    FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read, so read bytes = FETCH_SIZE * 1024 * 2.
    Returns {"mirror": {...}, "f32": {...}} or None when rocprofv3 is not usable here."""
    import glob
    import shutil
    import sqlite3
    import subprocess
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
            for half, tag in ((True, "mirror"), (False, "f32")):
                # scan_kernel<METRIC, MASKED, NQ, CHUNKS, .., .., HALF>: the sweeps proper are the launches within 2x of the
                # largest (the f32 retry launches of a mirror pass share the f32 template and return at once)
                want = "true" if half else "false"
                v = [val for name, val in rows
                     if (m := re.search(r"scan_kernel<[^>]*?(true|false)>", name)) and m.group(1) == want]
                if not v:
                    return None
                big = [x for x in v if x * 2 >= max(v)]
                per.setdefault(tag, {})[counter] = (float(np.mean(big)), len(big))
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for tag, d in per.items():
        rd = d["FETCH_SIZE"][0] * 1024 * 2
        wr = d["WRITE_SIZE"][0] * 1024
        res[tag] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr,
                    "launches_counted": d["FETCH_SIZE"][1],
                    "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE --kernel-trace and --pmc WRITE_SIZE "
                              "--kernel-trace (separate passes) around child runs of bench.py --pmc-child; "
                              "FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE*1024"}
    return res


def pmc_child(args):
    """What live_pmc() profiles: a few nq=1 sweeps over the bf16 mirror, then over the f32 corpus; no output."""
    import torch
    from neumann_amd import GpuFlatIndex
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx = GpuFlatIndex(args.dim, args.rows, device=0)
    idx.fill_synthetic(SEED_CORPUS, args.rows)
    q = torch.from_numpy(_synth(SEED_QUERY, 0, 4, args.dim)).to(dev)
    for mirror in (True, False):
        idx.set_mirror(mirror)
        for i in range(8):
            idx.search_device(q[i % 4:i % 4 + 1], args.k, METRICS[args.metric])
        torch.cuda.synchronize()
    idx.close()


def certificate(idx, q_host, metric, rows, scores, counts, world, dev, mask_host=None):
    """Size-independent proof that (rows, scores) is the exact top-k of the whole sharded corpus, using
    only the product's exact (reference-order) kernels, which tests/ pin bit-for-bit to the oracle:
      1. every returned score equals the exact score of its row (owner shard recomputes it);
      2. the list is ordered (score desc, row asc);
      3. #rows anywhere with exact score > s_k  ==  #returned scores > s_k, and the returned ties at
         s_k do not exceed the corpus-wide number of rows scoring exactly s_k."""
    import torch
    import torch.distributed as dist
    cnt = int(counts[0])
    r = rows[0, :cnt].astype(np.uint64)
    s = scores[0, :cnt]
    base, n_local = idx.row_base, idx.rows
    mine = (r >= base) & (r < base + n_local)
    ok_scores = True
    if mine.any():
        ex = idx.score_rows(q_host, (r[mine] - np.uint64(base)), metric)[0]
        ok_scores = bool(np.all(ex == s[mine]))
    ordered = bool(np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (r[:-1] < r[1:])))) if cnt > 1 else True
    sk = float(s[-1]) if cnt else float("inf")
    gt, eq = idx.count_exact(q_host, sk, metric, mask=mask_host) if cnt else (0, 0)
    agg = torch.tensor([gt, eq, 0 if ok_scores else 1], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(agg)
    gt, eq, bad = (int(x) for x in agg.tolist())
    n_gt_ret = int(np.sum(s > np.float32(sk)))
    n_eq_ret = int(np.sum(s == np.float32(sk)))
    exact = bad == 0 and ordered and gt == n_gt_ret and n_eq_ret <= eq and (cnt == 0 or n_eq_ret >= 1)
    return {"exact_topk_certified": bool(exact), "scores_bit_equal_exact_kernel": bad == 0, "ordered": ordered,
            "rows_above_kth": gt, "rows_equal_kth": eq, "returned": cnt, "recall_at_k": 1.0 if exact else None}


def measure_batched(args, idx, dev, metric, total_rows, torch):
    """Config 3 on the same resident corpus: nq queries per step through the MFMA sweep (one corpus sweep per 64
    queries), two steps in flight; the last batch is certified query by query with the exact kernels."""
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
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows, scores, counts = (t.cpu().numpy().copy() for t in out)
    idx.set_timing(True)
    sweep_ms = []
    elem_bytes = 4
    for i in range(6):
        torch.cuda.synchronize()  # the sweep alone on the device: its HIP events must not span another step's kernels
        step(i)
        st = idx.last_stats(streams[i % 2])
        if st.scan_ms > 0:
            sweep_ms.append(st.scan_ms)
        if st.rows_scanned:
            elem_bytes = int(st.bytes_scanned // (st.rows_scanned * args.dim))
    idx.set_timing(False)
    torch.cuda.synchronize()
    qh = q_host[(steps - 1) % 4]
    ok = True
    if not args.no_parity:
        for qi in (0, nq // 2, nq - 1):  # three of the batch's queries: the exact certificate is a full pass each
            c = certificate(idx, qh[qi], metric, rows.view(np.uint64)[qi:qi + 1], scores[qi:qi + 1], counts[qi:qi + 1],
                            1, dev)
            ok = ok and c["exact_topk_certified"]
    # stationary queries of one matrix-core sweep (launch_metric in nmn_scan_mfma.hip): 128 when the pass holds more
    # than 64 queries and the rows are <= 768 elements long, else 64
    per_sweep = 128 if ((args.dim // 128 <= 6 or args.dim // 128 in (8, 10)) and nq > 64) else 32 if args.dim // 128 in (16, 24, 32) else 64
    sweeps = (nq + per_sweep - 1) // per_sweep
    sweep = float(np.mean(sweep_ms)) if sweep_ms else float("nan")
    # ONE launch carries every query block; the workgroups that stream the same tiles for different blocks sit on the
    # same XCD (ids 8 apart) and share its L2, so the algorithmic HBM bytes are one corpus read per launch.  The bytes
    # the workgroups REQUEST (query blocks x corpus) are reported beside it.
    gbps = idx.rows * args.dim * elem_bytes / (sweep * 1e-3) / 1e9
    return {"workload": f"{total_rows}x{args.dim} f32 {args.metric} TOP-{args.k}, nq={nq}/step (MFMA sweep)",
            "value": nq * steps / dt, "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "sweep_ms_incl_sampling_pass": sweep, "query_blocks_per_launch": sweeps,
            "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbps / HBM_PEAK_GBS, "kernel": "nmn::scan_mfma_kernel (+3% sampling pass)",
                         "bytes_per_corpus_element": elem_bytes, "requested_GBs_all_query_blocks": gbps * sweeps},
            "exact_topk_certified_3_of_batch": bool(ok) if not args.no_parity else None}


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # NMN_BENCH_DEVICE pins every rank to one device: lets the N>1 code path run on a 1-GPU box where the
    # collective library tolerates several ranks per GPU (debugging aid, never used by the driver)
    dev_index = int(os.environ.get("NMN_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or args.always_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        backend = os.environ.get("NMN_BENCH_BACKEND", "nccl")  # "gloo": control-flow check of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from neumann_amd import GpuFlatIndex
    from neumann_amd.sharded import ShardedSearcher, shard_range

    metric = METRICS[args.metric]
    if args.scaling == "weak":
        total_rows = args.rows * world
    else:
        total_rows = args.rows
    r0, r1 = shard_range(total_rows, world, rank)
    local_rows = r1 - r0

    idx = GpuFlatIndex(args.dim, local_rows, row_base=r0, device=dev_index)
    t_fill = time.perf_counter()
    idx.fill_synthetic(SEED_CORPUS, local_rows)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill

    n_query_sets = 16
    q_host = np.stack([_synth(SEED_QUERY, s * args.nq, args.nq, args.dim) for s in range(n_query_sets)])
    q_dev = torch.from_numpy(q_host).to(dev)  # [sets, nq, dim] resident in HBM
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    searchers = [ShardedSearcher(idx, world_size=world, rank=rank, k=args.k, nq=args.nq, device=dev,
                                 always_gather=args.always_gather)
                 for _ in range(n_streams)]  # one set of result buffers (and one library workspace) per stream
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

    def step(i):
        with torch.cuda.stream(streams[i % n_streams]):
            return searchers[i % n_streams].search_device(q_dev[i % n_query_sets], metric, mask_t=mask_dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run():
        """W untimed steps, K timed steps between fences (max over ranks), then the dominant kernel's HIP-event average
        over up to 30 more steps.  Returns (elapsed_s, last result, scan_ms list, total_ms list, bytes per element)."""
        for i in range(args.warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        last = tuple(t.cpu().numpy().copy() for t in out)  # result of the last timed step
        # dominant-kernel timing: HIP events on the launch stream, recorded inside the library
        idx.set_timing(True)
        scan_ms, total_ms = [], []
        eb = 4  # bytes per corpus element the dominant kernel reads: 4 (f32 corpus) or 2 (its bf16 mirror)
        for i in range(min(max(args.steps, 5), 30)):
            step(i)
            st = idx.last_stats(streams[i % n_streams])
            if st.scan_ms > 0:
                scan_ms.append(st.scan_ms)
                total_ms.append(st.total_ms)
            if st.rows_scanned:
                eb = int(st.bytes_scanned // (st.rows_scanned * args.dim))
        idx.set_timing(False)
        fence()
        return elapsed, last, scan_ms, total_ms, eb

    elapsed, last_out, scan_ms, total_ms, elem_bytes = timed_run()
    ms_per_step = elapsed / args.steps * 1e3
    value = args.nq * args.steps / elapsed
    scan_avg = float(np.mean(scan_ms)) if scan_ms else float("nan")
    # corpus sweeps per step: 64 queries per sweep on the MFMA path (>= 3
    # queries at dim >= 768, else >= 5; cosine/dot), else 4 (VALU) — mirrors search_enqueue() / scan_mfma_supported() in neumann_amd/csrc
    ld128 = (args.dim + 127) // 128 * 128  # nmn_index_create pads rows just short of a supported multiple of 128 up to it
    while ld128 <= 4096 and not (ld128 // 128 <= 6 or ld128 // 128 in (8, 10, 12, 16, 24, 32)):
        ld128 += 128
    kc = ld128 // 128 if (ld128 <= 4096 and (ld128 - args.dim) * 8 <= args.dim) else 0
    mfma_min = int(os.environ.get("NMN_MFMA_MIN_NQ") or 0) or (3 if args.dim >= 768 else 5)  # mfma_min_queries()
    mfma = (args.nq >= mfma_min and args.metric in ("cosine", "dot", "euclidean") and kc and (kc <= 6 or kc in (8, 10, 12, 16, 24, 32))
            and args.k <= 4096)
    # the matrix-core sweep is ONE launch whatever the number of query blocks (they share the tiles through the XCD's L2);
    # VALU sweeps are one launch per 4 queries
    passes = 1 if mfma else ((args.nq + 3) // 4 if args.nq >= 3 else 1)
    if args.k > 4096:
        passes = 1  # large-k path: one exact scan per query, the first one is the timed launch
    alg_bytes = (kept_rows * args.dim * elem_bytes + (local_rows // 8 if mask_dev is not None else 0)) * passes  # excluded rows are never read
    achieved = alg_bytes / (scan_avg * 1e-3) / 1e9 if scan_ms else float("nan")

    # ---- read ceiling of this device: the scan's access pattern with the arithmetic removed (rank 0 reports) ----
    read_ceiling = idx.read_probe(3) if local_rows else None

    # ---- parity of the last result ----------------------------------------------------------------
    parity = None
    if not args.no_parity:
        o_rows, o_scores, o_counts = last_out
        parity = certificate(idx, q_host[(args.steps - 1) % n_query_sets][0], metric,
                             o_rows.view(np.uint64), o_scores, o_counts, world, dev, mask_host)

    # ---- the sweep SURVEY §8(d) prices: the row-major f32 corpus itself, same index / queries / loop ----
    f32_leg = None
    if world == 1 and elem_bytes == 2 and not args.no_f32_leg and args.k <= 4096 and args.nq == 1:
        idx.set_mirror(False)
        e2, out2, scan2, total2, eb2 = timed_run()
        idx.set_mirror(True)
        k_ms = float(np.mean(scan2)) if scan2 else float("nan")
        bytes4 = (kept_rows * args.dim * 4 + (local_rows // 8 if mask_dev is not None else 0)) * passes
        ach = bytes4 / (k_ms * 1e-3) / 1e9 if scan2 else float("nan")
        cert2 = None
        if not args.no_parity:
            cert2 = certificate(idx, q_host[(args.steps - 1) % n_query_sets][0], metric, out2[0].view(np.uint64), out2[1],
                                out2[2], world, dev, mask_host)
        same = bool(np.array_equal(out2[0], last_out[0]) and np.array_equal(out2[1].view(np.uint32), last_out[1].view(np.uint32)))
        f32_leg = {"what": "nmn_index_set_mirror(0): scan_kernel streams the row-major f32 corpus (SURVEY §8(d): rows*dim*4 "
                           "bytes per query); same index, queries, streams, steps and warmup as the headline loop",
                   "value": args.nq * args.steps / e2, "unit": "queries/s", "ms_per_step": e2 / args.steps * 1e3,
                   "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS if scan2 else None,
                   "avg_kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes4, "bytes_per_corpus_element": eb2,
                   "pricing": "SURVEY §8(d): rows*dim*4",
                   "frac_of_read_ceiling": (ach / read_ceiling) if (scan2 and read_ceiling) else None,
                   "traffic": pmc_traffic(local_rows, args, 4)[0], "traffic_source": pmc_traffic(local_rows, args, 4)[1],
                   "exact_topk_certified": cert2["exact_topk_certified"] if cert2 else None,
                   "same_answer_as_mirror_sweep": same}

    batched = None
    if world == 1 and args.batched > 0 and args.nq == 1 and args.mask >= 1.0:
        batched = measure_batched(args, idx, dev, metric, total_rows, torch)

    # Single-query calls from many host threads at once (the reference's Arc<VectorEngine> under concurrent clients):
    # the C ABI merges callers that arrive while the shard is busy into one query batch.  Native threads, 1 second.
    callers = None
    if world == 1 and args.callers > 0 and args.nq == 1 and args.mask >= 1.0 and args.k <= 4096:
        cq = _synth(SEED_QUERY + 2, 0, args.callers, args.dim)
        r = idx.callers_probe(cq, args.k, metric, seconds=1.0)
        callers = {"workload": f"{args.callers} host threads, each nmn_index_search(nq=1, k={args.k}) in a loop, "
                               f"{total_rows}x{args.dim} f32 {args.metric}",
                   "value": r["calls_per_s"], "unit": "queries/s", "threads": args.callers,
                   "sweeps_carrying_2_or_more_calls": r["merged_batches"], "calls_in_them": r["merged_calls"],
                   "answers_differing_from_a_lone_call": r["mismatches"]}
        # twice the threads: one sweep carries up to 128 callers
        cq2 = _synth(SEED_QUERY + 3, 0, 2 * args.callers, args.dim)
        r2 = idx.callers_probe(cq2, args.k, metric, seconds=1.0)
        callers["with_twice_the_threads"] = {"threads": 2 * args.callers, "value": r2["calls_per_s"], "unit": "queries/s",
                                             "answers_differing_from_a_lone_call": r2["mismatches"]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, metric, total_rows, local_rank)

    traffic, traffic_src = pmc_traffic(local_rows, args, elem_bytes)
    others = None
    default_workload = (args.rows == 10_000_000 and args.dim == 768 and args.k == 100 and args.nq == 1 and
                        args.metric == "cosine" and args.mask >= 1.0)
    # HBM traffic of the dominant kernel from PMC counters collected in THIS run (the committed profiles/pmc_traffic.json
    # figure above is the fallback where rocprofv3 is not usable)
    if rank == 0 and world == 1 and default_workload and not args.no_live_pmc and not args.always_gather:
        live = live_pmc(args)
        if live:
            key = "mirror" if elem_bytes == 2 else "f32"
            traffic, traffic_src = live[key]["hbm_bytes_per_launch"], live[key]["source"]
            if f32_leg is not None:
                f32_leg["traffic"], f32_leg["traffic_source"] = live["f32"]["hbm_bytes_per_launch"], live["f32"]["source"]
                f32_leg["traffic_read_write"] = [live["f32"]["read_bytes_corrected"], live["f32"]["write_bytes"]]
            traffic_rw = [live[key]["read_bytes_corrected"], live[key]["write_bytes"]]
        else:
            traffic_rw = None
    else:
        traffic_rw = None
    if world == 1 and default_workload and not args.no_other_configs and not args.always_gather:
        idx.close()  # the children need the HBM (config 5 alone is 61 GB + workspace)
        others = other_configs()
    if rank == 0:
        line = {
            "metric": "queries/sec, brute-force SIMILAR TOP-K (recall@K = 1.0 vs CPU oracle)",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{total_rows}x{args.dim} f32 {args.metric} TOP-{args.k}, nq={args.nq}/step"
                                   + (f", WHERE mask selectivity {args.mask}" if args.mask < 1.0 else ""),
                       "rows_total": total_rows, "rows_per_gpu": local_rows, "dim": args.dim, "k": args.k,
                       "nq": args.nq, "streams": n_streams,
                       "approximate_sweep": ("bf16 mirror of the corpus, f32 accumulate" if elem_bytes == 2 else "f32 corpus")
                                            + "; every candidate re-scored from the f32 corpus in the reference's order",
                       "parallelism": f"row-range shards x{world}, RCCL all-gather of top-k"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if scan_ms else None, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_read_write": traffic_rw,
                         "kernel": ("nmn::exact_scan_kernel" if args.k > 4096 else
                                    "nmn::scan_mfma_kernel" if mfma else "nmn::scan_kernel"), "avg_kernel_ms": scan_avg,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_corpus_element": elem_bytes,
                         "pricing": ("bytes the kernel is asked to read: rows*dim*2, the bf16 mirror" if elem_bytes == 2
                                     else "SURVEY §8(d): rows*dim*4, the f32 corpus"),
                         # the same kernel time priced as SURVEY §8(d) writes it (N*d*4 per query): an EFFECTIVE rate
                         "achieved_priced_as_survey_8d": achieved * 4 / elem_bytes if scan_ms else None,
                         "frac_priced_as_survey_8d": achieved * 4 / elem_bytes / HBM_PEAK_GBS if scan_ms else None,
                         "pipeline_ms_per_query_batch": float(np.mean(total_ms)) if total_ms else None,
                         # measured in this run by nmn_index_read_probe: a pure read sweep, no arithmetic
                         "measured_read_ceiling": read_ceiling,
                         "frac_of_read_ceiling": (achieved / read_ceiling) if (scan_ms and read_ceiling) else None},
            "roofline_f32_corpus": f32_leg,
            "cpu_baseline": cpu,
            "parity": parity,
            "batched": batched,
            "concurrent_callers": callers,
            "other_configs": others,
            "fill_s": t_fill,
        }
        print(json.dumps(line), flush=True)
    if world > 1 or args.always_gather:
        dist.barrier()  # every rank is done with its collectives before any of them tears the group down
        dist.destroy_process_group()


def _synth(seed, row0, n, dim):
    from neumann_amd import synth_rows
    return synth_rows(seed, row0, n, dim)


if __name__ == "__main__":
    main()
