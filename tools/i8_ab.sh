#!/bin/bash
# A/B of 8-bit sweep variants (tools/build_variant.sh NAME FLAGS nmn_scan_i8): interleaved bench runs, kernel average + q/s per
# variant; "default" = the shipped library.   bash tools/i8_ab.sh default i8nopipe ...   [DIM=768 METRIC=cosine K=100 ROUNDS=3]
cd ${GRAFT_REPO_ROOT:-$PWD}
DIM=${DIM:-768}; METRIC=${METRIC:-cosine}; K=${K:-100}; ROUNDS=${ROUNDS:-3}; ROWS=${ROWS:-10000000}
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    lib=""; [ "$v" != default ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so
    NEUMANN_GPU_LIB=$lib python bench.py --rows $ROWS --dim $DIM --metric $METRIC --k $K --steps 30 --warmup 5 --rebuilds 1 --no-cpu-baseline \
      --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('variant %-10s round $r  %8.1f q/s  kernel %.4f ms  frac %.3f  of-ceiling %.3f  cands %s  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], r['frac_of_read_ceiling'], r['candidates_rescored'], d['parity']['exact_topk_certified']))"
  done
done
