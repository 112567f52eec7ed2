#!/bin/bash
# survivor walk of the masked 8-bit sweep on / off (NMN_NO_WALK=1), per selectivity, at config 5's shape and at 10M x 768 cosine TOP-100
cd ${GRAFT_REPO_ROOT:-$PWD}
SELS=${SELS:-"0.5 0.25 0.1 0.05 0.01"}; ROUNDS=${ROUNDS:-2}
run() {  # $1 = label, rest = bench args
  label=$1; shift
  for r in $(seq $ROUNDS); do for sel in $SELS; do for v in walk nowalk; do
    env=""; [ $v = nowalk ] && env="NMN_NO_WALK=1"
    env $env python bench.py "$@" --mask $sel --steps 20 --warmup 4 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 \
        --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$label %-7s sel $sel round $r  %8.1f q/s  kernel %.4f ms  frac %.3f  cands %s  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], r['candidates_rescored'], d['parity']['exact_topk_certified']))"
  done; done; done
}
run "10Mx1536 L2 k=1000" --dim 1536 --metric euclidean --k 1000
run "10Mx768 cos k=100 " --dim 768 --metric cosine --k 100
