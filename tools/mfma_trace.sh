#!/bin/bash
# per-kernel durations of the batched leg (rocprofv3 kernel trace): library variant $1 ("" = default), queries $2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/trace_tmp_$1_$2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ] && [ "$1" != "default" ]; then export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$1.so; fi
rocprofv3 --kernel-trace -d $O -o t -- python $R/bench.py --batched $2 --steps 10 --no-other-configs --no-cpu-baseline --callers 0 --no-parity --no-f32-leg --no-live-pmc > /dev/null 2>&1
DB=$(find $O -name "*.db" | head -1)
python - <<PY
import sqlite3, collections
db = sqlite3.connect("$DB")
rows = list(db.execute("select name, end-start from kernels where name like '%scan_mfma%'"))
d = collections.defaultdict(list)
for n, t in rows: d[n[:60]].append(t/1e3)
for n, v in d.items():
    v.sort()
    big = [x for x in v if x > 0.5*v[-1]]
    small = [x for x in v if x <= 0.5*v[-1]]
    print("variant=%-8s nq=$2 %-56s main sweep: n=%d min=%.1f med=%.1f us | sampling pass: n=%d min=%.1f" % ("${1:-default}", n, len(big), big[0], big[len(big)//2], len(small), small[0] if small else 0))
PY
rm -rf $O
