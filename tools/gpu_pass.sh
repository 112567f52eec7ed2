#!/bin/bash
# scratch: f32 survivor walk: density threshold
R=$PWD
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --rebuilds 1 --dim 1536 --metric euclidean --k 1000 --steps 12"
for m in 0.5 0.3 0.1; do
for mode in default wd40 wd64 default wd40 wd64; do
  if [ $mode = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$mode.so; fi
  timeout 120 python bench.py $COMMON --mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mask $m %-7s %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f' % ('$mode', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
done; done
