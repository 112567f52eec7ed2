#!/bin/bash
# per-kernel durations of one all-identical-rows search (the exact-fallback route)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/ftrace -o t -- python $R/tools/fallback_probe.py > $O/fallback_traced.txt 2>/dev/null
DB=$(find $O/ftrace -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "tools/fallback_probe.py (10M identical rows x 768; 3 metrics x 7 searches)" > $O/fallback_kernel_trace.txt 2>&1
rm -rf $O/ftrace
cat $O/fallback_kernel_trace.txt | head -40
