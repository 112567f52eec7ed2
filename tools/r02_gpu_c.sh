#!/bin/bash
# round 2, GPU call C (final tree): whole gpu suite, default bench (f32 leg + live PMC + other configs), 2-stream kernel
# trace + timelines, matrix-core sweep across shapes at 64 and 128 queries, clean single-stream loops
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
[ -n "$SKIP_SUITE" ] || ( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite.log 2>&1   # SKIP_SUITE=1 / SKIP_SHAPES=1: artifacts refreshed elsewhere
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity > $O/bench_traced.json 2> $O/bench_traced.err
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/trace_timeline.py $DB --steps 50 --warmup 5 --title "python bench.py --steps 50 --warmup 5 (2 streams, bf16 mirror), then the f32 leg" > $O/timeline_mirror.txt 2>&1
python $R/tools/trace_timeline.py $DB --steps 50 --warmup 5 --kernel "true, false>" --min-us 1000 --title "f32 leg (nmn_index_set_mirror(0)) of the same run" > $O/timeline_f32.txt 2>&1
python $R/tools/prof_summary.py $DB "bench.py --steps 50 --warmup 5 --streams 2 (default), mirror loop + f32 leg" > $O/kernel_trace.txt 2>&1
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --batched 128 --steps 10 --no-cpu-baseline --no-other-configs --callers 0 --no-live-pmc --no-f32-leg > $O/bench_batched128_traced.json 2>/dev/null
DB=$(find $O/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "bench.py --batched 128 --steps 10 (nq=1 loop, then 128-query batches)" > $O/kernel_trace_batched128.txt 2>&1
rm -rf $O/trace
cd $R
[ -n "$SKIP_SHAPES" ] || { echo "# matrix-core sweep across row lengths, final tree (tools/mfma_shapes.sh; sweep_ms includes the sampling pass)"; NQ=64 bash tools/mfma_shapes.sh; NQ=128 bash tools/mfma_shapes.sh; } > $O/mfma_shapes.txt 2>&1
[ -n "$SKIP_SHAPES" ] || { python tools/mfma_loop.py --nq 64 --realloc 4; python tools/mfma_loop.py --nq 128 --realloc 4; python tools/mfma_loop.py --nq 64 --realloc 4; python tools/mfma_loop.py --nq 128 --realloc 4; } > $O/mfma_loop.txt 2>&1
timeout 300 python tools/latency_probe.py 1000:128:5 10000:128:5 10000:768:10 65536:128:5 100000:768:100 1000000:768:100 10000000:768:100 > $O/latency.txt 2>&1
timeout 300 python tools/fallback_probe.py > $O/fallback.txt 2>/dev/null
NMN_NO_GRID_SELECT=1 timeout 300 python tools/fallback_probe.py >> $O/fallback.txt 2>/dev/null
bash tools/mask_ab.sh default > $O/masks.txt 2>&1
tail -3 $O/gpu_suite.log; cat $O/smoke.log | tail -1; head -c 1500 $O/bench_default.json; echo; head -12 $O/timeline_mirror.txt; cat $O/mfma_shapes.txt $O/mfma_loop.txt $O/latency.txt $O/fallback.txt $O/masks.txt
