#!/bin/bash
# scratch: final evidence on the final tree
R=$PWD; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/suite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/suite.txt
cat gpurun_out/suite.txt | tail -3
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 300 gpurun_out/bench_default.json; echo
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof1 -o p -- python $R/bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8 > /tmp/prof1.log 2>&1
DB=$(find /tmp/prof1 -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "python bench.py --steps 20 --warmup 5 --streams 1 --rebuilds 1 --no-other-configs --no-cpu-baseline --callers 0 --no-live-pmc --batched 0 --legs i8  (ONE stream: the headline loop on the f32 corpus, then the 8-bit mirror leg)" > $R/gpurun_out/trace_1stream.txt
head -6 $R/gpurun_out/trace_1stream.txt | cut -c1-200
