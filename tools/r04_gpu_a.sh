#!/bin/bash
# round 4, first GPU pass: the whole GPU suite (incl. config 4 at full size), then the default bench line
set -x
OUT=gpurun_out/r04a; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
tail -c 600 $OUT/bench.err
