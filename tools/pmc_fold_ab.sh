#!/bin/bash
# HBM fetch bytes of the matrix-core sweep with the query blocks folded onto one XCD vs as a plain 2-D grid
# (NMN_MFMA_NO_FOLD=1): rocprofv3 --pmc FETCH_SIZE passes (--kernel-trace only), 2M x 3072, 64 queries = 2 query blocks.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_fold
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--rows ${ROWS:-2000000} --dim ${DIM:-3072} --batched ${NQ:-64} --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --callers 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fold -o f -- python $R/bench.py $ARGS > $OUT/fold.log 2>&1
NMN_MFMA_NO_FOLD=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/nofold -o f -- python $R/bench.py $ARGS > $OUT/nofold.log 2>&1
python - <<PY
import sqlite3, glob
for tag in ("fold", "nofold"):
    for db in glob.glob("$OUT/%s/**/*.db" % tag, recursive=True):
        c = sqlite3.connect(db)
        for name, n, a, mn, mx in c.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%scan_mfma%' group by kernel_name order by 3 desc"):
            print("%-7s %-90s launches=%d avg_read_GB=%.3f (FETCH_SIZE KiB x 1024 x 2) min=%.3f max=%.3f" % (tag, name[:90], n, a*2048/1e9, mn*2048/1e9, mx*2048/1e9))
PY
