#!/bin/bash
# kernel-level view of the single-launch path: duration of tiny_search_kernel at the reference's published size
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in ${CASES:-1000:128:5 10000:128:5 10000:768:10 65536:128:5}; do
  timeout 200 rocprofv3 --kernel-trace -d $O/ttrace -o t -- python $R/tools/latency_probe.py $c > $O/tiny_lat_$c.txt 2>/dev/null
  DB=$(find $O/ttrace -name "*.db" | head -1)
  echo "== $c" ; python $R/tools/prof_summary.py $DB "latency_probe $c" 2>&1 | grep -i "tiny\|Name" | head -3
  rm -rf $O/ttrace
done
