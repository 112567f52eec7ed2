#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in prof prof_nowr prof_noepi; do for nq in 64 128; do
echo "== $v nq=$nq"
NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so python bench.py --batched $nq --steps 6 --streams 1 --no-other-configs --no-cpu-baseline --callers 0 --no-parity --no-f32-leg --no-live-pmc 2>&1 | grep -o "prof wg17 wave[0-9] stages.*" | sort | uniq -c | sort -rn | head -2
done; done
