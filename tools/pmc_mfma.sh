#!/bin/bash
# HBM traffic of the matrix-core sweep (final tree): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (--kernel-trace only)
# around bench.py --batched NQ at 10M x 768, summarised per kernel by tools/pmc_summary.py
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for NQ in 64 128; do
  ARGS="--batched $NQ --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --callers 0 --no-f32-leg --no-live-pmc --no-parity"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f$NQ -o f -- python $R/bench.py $ARGS > $O/f$NQ.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w$NQ -o w -- python $R/bench.py $ARGS > $O/w$NQ.log 2>&1
  F=$(find $O/f$NQ -name "*.db" | head -1); W=$(find $O/w$NQ -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $F $W "bench.py --batched $NQ (10M x 768, nq=1 loop then $NQ-query batches), final tree of round 2" $O/pmc_$NQ.json > $O/pmc_$NQ.txt 2>&1
  rm -rf $O/f$NQ $O/w$NQ
done
cat $O/pmc_64.txt $O/pmc_128.txt
