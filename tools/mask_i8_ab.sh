#!/bin/bash
# masked 8-bit sweep at config 5's shape (10M x 1536 L2 TOP-1000) per selectivity and library variant.  bash tools/mask_i8_ab.sh default NAME ...
cd ${GRAFT_REPO_ROOT:-$PWD}
SELS=${SELS:-"0.5 0.1 0.05 0.01"}; ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do for sel in $SELS; do for v in "$@"; do
  lib=""; [ "$v" != default ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so
  NEUMANN_GPU_LIB=$lib python bench.py --dim 1536 --metric euclidean --k 1000 --mask $sel --steps 20 --warmup 4 --rebuilds 1 --no-cpu-baseline \
      --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('variant %-10s sel $sel round $r  %8.1f q/s  kernel %.4f ms  frac %.3f  bytes/elem %d  cands %s  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], r['bytes_per_corpus_element'], r['candidates_rescored'], d['parity']['exact_topk_certified']))"
done; done; done
