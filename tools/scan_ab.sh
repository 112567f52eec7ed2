#!/bin/bash
# headline sweep (nq = 1, 2 streams) per library variant, ROUNDS processes each, interleaved: args = variant names; EXTRA = bench flags
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-3}); do
for v in "$@"; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-f32-leg --no-live-pmc $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-9s q/s=%.1f step_ms=%.3f scan_ms=%.3f frac=%.3f certified=%s' % ('$v', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['parity']['exact_topk_certified']))"
done; done
