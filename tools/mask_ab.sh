#!/bin/bash
# config 5 (10M x 1536 L2 TOP-1000, nq=1) under WHERE-predicate selectivities per library variant ($@: names built by
# tools/build_variant.sh NAME FLAGS nmn_scan; "default" = the shipped library), then the dense 10M x 768 cosine sweep
cd ${GRAFT_REPO_ROOT:-$PWD}
for vs in "$@"; do
  v=${vs%%:*}
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  for m in ${MASKS:-0.5 0.1 0.01}; do
    python bench.py --dim 1536 --metric euclidean --k 1000 --mask $m --steps 10 --warmup 2 --no-cpu-baseline --no-f32-leg --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant=$vs mask=%s: q/s=%.1f step_ms=%.3f scan_ms=%.3f GB/s=%.0f frac=%.3f' % ('$m', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']), d['parity']['exact_topk_certified'])"
  done
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg --no-live-pmc --no-other-configs --batched 0 --callers 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant=$vs dense768: q/s=%.1f step_ms=%.3f scan_ms=%.3f GB/s=%.0f frac=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']), d['parity']['exact_topk_certified'])"
done
