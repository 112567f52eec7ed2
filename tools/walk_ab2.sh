#!/bin/bash
# variants of the survivor walk: bash tools/walk_ab2.sh default pipe6 ...
cd ${GRAFT_REPO_ROOT:-$PWD}
SELS=${SELS:-"0.5 0.1 0.01"}; ROUNDS=${ROUNDS:-2}
for shape in "--dim 1536 --metric euclidean --k 1000" "--dim 768 --metric cosine --k 100"; do
for r in $(seq $ROUNDS); do for sel in $SELS; do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so
  NEUMANN_GPU_LIB=$lib python bench.py $shape --mask $sel --steps 20 --warmup 4 --rebuilds 1 --no-cpu-baseline \
      --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$shape variant %-8s sel $sel round $r  %8.1f q/s  kernel %.4f ms  frac %.3f  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified']))"
done; done; done; done
