O=gpurun_out/r05zz; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && rm -rf /tmp/prof_a && rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o p -- python $R/bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_1stream.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_a -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (ONE stream: the headline loop on the f32 corpus, then the 8-bit mirror leg)" > $O/kernel_trace_f32_headline_1stream.txt
( cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p -- python $R/bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > $R/$O/bench_2streams.json 2>/dev/null )
python tools/prof_summary.py $(find /tmp/prof_b -name "*.db" | head -1) "python bench.py --steps 20 --warmup 5 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (two streams: the default pipelining)" > $O/kernel_trace_f32_headline_2streams.txt
python tools/trace_timeline.py $(find /tmp/prof_b -name "*.db" | head -1) --steps 20 --warmup 5 --kernel "scan_ring_kernel" > $O/timeline_f32_headline_2streams.txt 2>&1
python tools/pmc_sq.py --kernel scan_ring_kernel --title "the headline sweep: one f32 query, 10M x 768 (scan_ring_kernel)" -- python $R/tools/search_child.py --rows 10000000 --dim 768 --mirror 0 --api device --reps 6 > $O/pmc_sq_ring.txt 2>&1
cat $O/gpu_suite.txt $O/smoke.txt; tail -4 $O/bench_default.err; head -8 $O/kernel_trace_f32_headline_1stream.txt | cut -c1-200; tail -6 $O/timeline_f32_headline_2streams.txt; tail -5 $O/pmc_sq_ring.txt
