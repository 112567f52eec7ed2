#!/bin/bash
# select_kernel's score gather: 16 (default) vs 32 loads per thread in flight.  bash tools/select_vr_ab.sh default vr32
cd ${GRAFT_REPO_ROOT:-$PWD}
for shape in "--rows 1000000" "--rows 10000000" "--rows 10000000 --dim 1536 --metric euclidean --k 1000 --mask 0.1"; do for r in 1 2; do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so
  NEUMANN_GPU_LIB=$lib python bench.py $shape --steps 60 --warmup 6 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$shape variant %-8s round $r  %8.1f q/s  ms/step %.4f  certified %s' % ('$v', d['value'], d['ms_per_step'], d['parity']['exact_topk_certified']))"
done; done; done
