#!/bin/bash
# round 3, final GPU call: full-size parity + whole gpu suite, default bench (live PMC, all legs), kernel traces of the headline
# loop (2 streams, 8-bit mirror) and of a 64-query batch, timelines, PMC of the batched 8-bit sweep
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite_pytest.log 2>&1
echo "gpu suite rc=$?" >> $O/gpu_suite_pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
# the very command the timed loop is (2 streams), traced: span / steps must reproduce ms_per_step
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --steps 50 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity > $O/bench_traced_2streams.json 2> $O/bench_traced.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/trace_timeline.py $DB --steps 50 --warmup 5 --kernel scan_i8_kernel --title "python bench.py --steps 50 --warmup 5 --rebuilds 1 (2 streams, 8-bit mirror): the timed loop" > $O/timeline_i8.txt 2>&1
python $R/tools/prof_summary.py $DB "bench.py --steps 50 --warmup 5 --rebuilds 1 (headline loop on the 8-bit mirror, then the f32 and bf16 legs)" > $O/kernel_trace.txt 2>&1
rm -rf $O/trace
# a 64-query batch loop, traced
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace64 -o t -- python $R/bench.py --nq 64 --steps 30 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc > $O/bench_batched64_traced.json 2> $O/bench_batched64.err
DB=$(find $O/trace64 -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "bench.py --nq 64 --steps 30 (matrix-core sweep over the 8-bit mirror)" > $O/kernel_trace_batched64.txt 2>&1
rm -rf $O/trace64
# PMC: HBM bytes of the batched 8-bit sweep (separate passes, kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/bench.py --nq 64 --steps 6 --warmup 2 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity > /dev/null 2> $O/pmc_$c.err
  DB=$(find $O/pmc_$c -name "*.db" | head -1)
  python - "$DB" $c >> $O/pmc_mfma_i8_10Mx768.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = sys.argv[2]
rows = list(db.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)))
agg = {}
for name, v in rows:
    k = "scan_mfma_kernel (I8)" if "scan_mfma_kernel" in name else None
    if k: agg.setdefault(k, []).append(v)
for k, v in agg.items():
    big = [x for x in v if x * 4 >= max(v)]   # the main sweeps (the sampling pass reads 1/32 of the mirror)
    small = [x for x in v if x * 4 < max(v)]
    scale = 1024 * (2 if c == "FETCH_SIZE" else 1)   # KiB; gfx950: FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads
    nb = max(1, len(small))   # one sampling pass per query batch; the main sweep of a batch is TWO launches (bound refinement)
    print(f"{c}: {k}: per batch of 64 queries: main sweep {sum(big)/nb*scale/1e9:.4f} GB in {len(big)//nb} launches; sampling pass {(sum(small)/nb)*scale/1e9:.4f} GB  ({nb} batches; algorithmic: 7.68 GB per sweep of the 8-bit mirror)")
PY
  rm -rf $O/pmc_$c
done
tail -4 $O/gpu_suite_pytest.log; head -12 $O/timeline_i8.txt; cat $O/pmc_mfma_i8_10Mx768.txt
# the same headline loop with ONE stream (steps serialised): here the trace's average duration of the sweep kernel is directly
# comparable with roofline.avg_kernel_ms of the bench line (which times the kernel with HIP events, one step on the device at a time)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace1 -o t -- python $R/bench.py --streams 1 --steps 50 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs > $O/bench_traced_1stream.json 2> $O/bench_traced1.err
DB=$(find $O/trace1 -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "bench.py --streams 1 --steps 50 --warmup 5 --rebuilds 1 --no-mirror-legs (one step on the device at a time)" > $O/kernel_trace_1stream.txt 2>&1
rm -rf $O/trace1
head -6 $O/kernel_trace_1stream.txt
