#!/bin/bash
# tools/variant_ab.sh "BENCH ARGS" VARIANT...: one bench.py leg under library variants (tools/build_variant.sh NAME "-DFLAG" [SOURCE];
# "default" = the shipped library), interleaved ROUNDS (default 2) times; prints q/s, ms per step, the dominant kernel's event time and
# its fraction of the HBM peak per run.  Example: tools/build_variant.sh pipe6 "-DNMN_I8_WALK_PIPE_CH=6" nmn_scan_i8 ;
# tools/variant_ab.sh "--dim 1536 --metric euclidean --k 1000 --steps 12 --mask 0.1" default pipe6
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
ARGS=$1; shift
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --rebuilds 1"
for round in $(seq ${ROUNDS:-2}); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
    python $R/bench.py $COMMON $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('variant %-12s round $round  %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f  certified %s' % ('$v', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified'] if d['parity'] else None))"
  done
done
