#!/bin/bash
# round 2, GPU call B: upload (one-pass ingest vs three kernels, + kernel trace), small-shard latency, the all-identical
# fallback, the sharded handle on one box
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 300 python tools/ingest_bench.py > $O/ingest.txt 2>$O/ingest.err
NMN_NO_INGEST=1 timeout 300 python tools/ingest_bench.py >> $O/ingest.txt 2>>$O/ingest.err
timeout 300 python tools/ingest_bench.py >> $O/ingest.txt 2>>$O/ingest.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $O/itrace -o t -- python $R/tools/ingest_bench.py --reps 2 > /dev/null 2>$O/itrace.err
  DB=$(find $O/itrace -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB "tools/ingest_bench.py --reps 2 (10M x 768 in 4 uploads of 2.5M rows, 3 passes)" > $O/ingest_kernel_trace.txt 2>&1
  rm -rf $O/itrace )
( cd /tmp && export TMPDIR=/tmp && NMN_NO_INGEST=1 timeout 300 rocprofv3 --kernel-trace -d $O/itrace -o t -- python $R/tools/ingest_bench.py --reps 2 > /dev/null 2>>$O/itrace.err
  DB=$(find $O/itrace -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB "NMN_NO_INGEST=1 tools/ingest_bench.py --reps 2 (round 1's three kernels)" > $O/ingest_kernel_trace_3kernels.txt 2>&1
  rm -rf $O/itrace )
timeout 300 python tools/latency_probe.py 1000:128:5 10000:128:5 10000:768:10 65536:128:5 100000:768:100 1000000:768:100 > $O/latency.txt 2>&1
NMN_NO_TINY=1 timeout 300 python tools/latency_probe.py 1000:128:5 10000:128:5 10000:768:10 65536:128:5 > $O/latency_notiny.txt 2>&1
timeout 300 python tools/fallback_probe.py > $O/fallback.txt 2>$O/fallback.err
NMN_NO_GRID_SELECT=1 timeout 300 python tools/fallback_probe.py >> $O/fallback.txt 2>>$O/fallback.err
timeout 400 python tools/sharded_probe.py > $O/sharded.txt 2>$O/sharded.err
tail -n 30 $O/ingest.txt $O/latency.txt $O/latency_notiny.txt $O/fallback.txt $O/sharded.txt; tail -5 $O/*.err
grep -i "ingest\|norms\|half" $O/ingest_kernel_trace.txt $O/ingest_kernel_trace_3kernels.txt | head -20
