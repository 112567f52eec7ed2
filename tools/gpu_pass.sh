#!/bin/bash
# scratch: final evidence on the final tree
R=$PWD; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/suite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/suite.txt
cat gpurun_out/suite.txt | tail -3
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 200 gpurun_out/bench_default.json; echo
