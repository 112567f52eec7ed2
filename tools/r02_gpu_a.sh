#!/bin/bash
# round 2, GPU call A: full-size parity tests, whole gpu suite, default bench (f32 leg + live PMC), 2-stream kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
( time timeout 1100 python -m pytest tests/test_gpu_fullsize.py -x -q ) > $O/fullsize.log 2>&1
echo "fullsize rc=$?" >> $O/fullsize.log
( time timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py ) > $O/gpu_suite.log 2>&1
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity > $O/bench_traced.json 2> $O/bench_traced.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/trace_timeline.py $DB --steps 50 --warmup 5 --title "python bench.py --steps 50 --warmup 5 (2 streams, bf16 mirror), then the f32 leg" > $O/timeline_mirror.txt 2>&1
# f32 leg: its 5 warm-up + 50 timed sweeps follow the mirror loop's 5 + 50 + 30 (timing) sweeps; the f32 template is a different kernel name
python $R/tools/trace_timeline.py $DB --steps 50 --warmup 5 --kernel "true, false>" --min-us 1000 --title "f32 leg (nmn_index_set_mirror(0)) of the same run" > $O/timeline_f32.txt 2>&1
python $R/tools/prof_summary.py $DB "bench.py --steps 50 --warmup 5 --streams 2 (default), mirror loop + f32 leg" > $O/kernel_trace.txt 2>&1
rm -rf $O/trace
tail -3 $O/fullsize.log $O/gpu_suite.log; cat $O/timeline_mirror.txt | head -12
