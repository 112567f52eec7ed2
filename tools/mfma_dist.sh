#!/bin/bash
# distribution of the main sweep's duration + the bench's own batched figures: variant $1, queries $2 (env passes through)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/trace_tmp_$1_$2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$1" ] && [ "$1" != "default" ]; then export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$1.so; fi
python $R/bench.py --batched $2 --steps 12 --no-other-configs --no-cpu-baseline --callers 0 --no-f32-leg --no-live-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('variant=$1 wgs=${NMN_MFMA_WGS:-dflt} nq=$2 q/s=%.0f step_ms=%.3f sweep_ms=%.3f frac=%.3f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['roofline']['frac'], b['exact_topk_certified_3_of_batch']))"
rocprofv3 --kernel-trace -d $O -o t -- python $R/bench.py --batched $2 --steps 10 --no-other-configs --no-cpu-baseline --callers 0 --no-parity --no-f32-leg --no-live-pmc > /dev/null 2>&1
DB=$(find $O -name "*.db" | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
v = sorted(t/1e3 for (t,) in db.execute("select end-start from kernels where name like '%scan_mfma%'"))
big = [x for x in v if x > 0.5*v[-1]]
print("   main sweep us:", " ".join("%.0f" % x for x in big))
PY
rm -rf $O
