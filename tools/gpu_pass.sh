#!/bin/bash
# scratch: exact_rows grid 512 -> 256: the active path (fallback probe) and the idle launch under a pipelined small shard
R=$PWD
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 5 --rebuilds 1 --no-parity"
for e in old new old new; do
  if [ $e = old ]; then export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_old.so; else unset NEUMANN_GPU_LIB; fi
  echo "== $e"
  python tools/fallback_probe.py 2>/dev/null | tail -1 | cut -c1-300
  python tools/fallback_probe.py --dim 1536 --rows 5000000 2>/dev/null | tail -1 | cut -c1-300
  for rows in 300000 1000000; do
  timeout 120 python bench.py $COMMON --rows $rows --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rows $rows %9.1f q/s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']))"
  done
done
unset NEUMANN_GPU_LIB
timeout 600 python -m pytest tests -m gpu -q -x -k "fallback or crowd or overflow or duplicate or tie" 2>&1 | tail -2
