#!/bin/bash
# masked sweeps of 128- / 384-element rows: eight lanes per row in the tile-by-tile steps (default) against the 16-lane mapping (variant wide128)
cd ${GRAFT_REPO_ROOT:-$PWD}
for dim in 128 384; do for sel in 0.9 0.5 0.25 0.1; do for v in default wide128; do
  lib=""; [ "$v" = wide128 ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_wide128.so
  NEUMANN_GPU_LIB=$lib python bench.py --rows 10000000 --dim $dim --mask $sel --steps 40 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('10M x $dim sel $sel %-8s  %8.1f q/s  kernel %.4f ms  frac %.3f  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified']))"
done; done; done
