#!/bin/bash
# config 5 (10M x 1536 L2 TOP-1000, nq=1) under WHERE selectivities, non-temporal corpus loads on / off (NMN_SCAN_NT=0), interleaved
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do for nt in 1 0; do
  export NMN_SCAN_NT=$nt
  for m in ${MASKS:-0.5 0.1 0.01}; do
    python bench.py --dim 1536 --metric euclidean --k 1000 --mask $m --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt=$nt mask=%s: q/s=%.1f step_ms=%.3f scan_ms=%.3f GB/s=%.0f frac=%.3f' % ('$m', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']), d['parity']['exact_topk_certified'])"
  done
done; done
