O=gpurun_out/r05p; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o p -- python $R/bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_1stream.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_a -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (headline loop on the f32 corpus, then the 8-bit mirror leg)" > $O/kernel_trace_f32_headline_1stream.txt
( cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_2streams.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_b -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (two streams: the default pipelining)" > $O/kernel_trace_f32_headline_2streams.txt
python tools/trace_timeline.py $(find /tmp/prof_b -name "*.db" | head -1) --steps 20 --warmup 5 --kernel "scan_kernel<" > $O/timeline_f32_headline_2streams.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o p -- python $R/tools/mfma_loop.py --mirror 0 --reps 20 > $R/$O/loop_f32_b64.txt 2>/dev/null )
DB=$(find /tmp/prof_c -name "*.db" | head -1)
python tools/prof_summary.py $DB "python tools/mfma_loop.py --mirror 0 --reps 20  (10M x 768, 64 queries per call over the f32 rows, one stream)" > $O/kernel_trace_batched64_f32.txt
python tools/trace_gantt.py $DB --kernel scan_mfma_kernel --skip 30 --steps 3 >> $O/kernel_trace_batched64_f32.txt
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/prof_d; ( cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_d -o p -- python $R/tools/mfma_loop.py --mirror 0 --reps 6 > /dev/null 2>&1 ); python - $(find /tmp/prof_d -name "*.db" | head -1) $c <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = sys.argv[2]
rows = [v for n, v in db.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)) if "scan_mfma_kernel" in n]
# per batch: sampling pass + two main launches; 4 warm + 6 timed calls
per_batch = sum(rows) / 10.0
scale = 1024 * (2 if c == "FETCH_SIZE" else 1)
print(f"{c}: scan_mfma_kernel launches {len(rows)}, per 64-query batch (sampling pass + both main launches) {per_batch * scale / 1e9:.3f} GB" + (" (FETCH_SIZE*1024*2: gfx950 correction)" if c == "FETCH_SIZE" else " (WRITE_SIZE*1024)"))
PY
done > $O/pmc_traffic_batched64_f32.txt
tail -c 300 $O/bench_default.err; cat $O/kernel_trace_f32_headline_1stream.txt | head -12; cat $O/timeline_f32_headline_2streams.txt | tail -8; cat $O/kernel_trace_batched64_f32.txt | head -8; cat $O/pmc_traffic_batched64_f32.txt
