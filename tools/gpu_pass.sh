O=gpurun_out/r05z; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o p -- python $R/bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_1stream.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_a -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (ONE stream: the headline loop on the f32 corpus, then the 8-bit mirror leg)" > $O/kernel_trace_f32_headline_1stream.txt
( cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_2streams.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_b -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (two streams: the default pipelining)" > $O/kernel_trace_f32_headline_2streams.txt
python tools/trace_timeline.py $(find /tmp/prof_b -name "*.db" | head -1) --steps 20 --warmup 5 --kernel "scan_kernel<" > $O/timeline_f32_headline_2streams.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_c && rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o p -- python $R/tools/mfma_loop.py --mirror 0 --reps 20 > $R/$O/loop_f32_b64.txt 2>/dev/null )
DB=$(find /tmp/prof_c -name "*.db" | head -1)
python tools/prof_summary.py $DB "python tools/mfma_loop.py --mirror 0 --reps 20  (10M x 768, 64 queries per call over the f32 rows, one stream)" > $O/kernel_trace_batched64_f32.txt
python tools/trace_gantt.py $DB --kernel scan_mfma_kernel --skip 30 --steps 3 >> $O/kernel_trace_batched64_f32.txt
for i in 1 2; do python tools/mfma_loop.py --mirror 1 --reps 12 --realloc 2 --nq 64 --tag i8_nq64; python tools/mfma_loop.py --mirror 1 --reps 12 --realloc 2 --nq 128 --tag i8_nq128; python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --nq 64 --tag f32_nq64; python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --nq 128 --tag f32_nq128; done > $O/mfma_medians_over_rebuilds.txt 2>&1
cat $O/gpu_suite.txt $O/smoke.txt; tail -4 $O/bench_default.err
