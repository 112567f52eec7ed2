#!/bin/bash
# scratch: masked ring sweep A/B (config 5 f32) through bench.py's headline loop
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --rebuilds 1 --dim 1536 --metric euclidean --k 1000 --steps 12"
for m in 0.5 0.1 0.02 0.004; do
for mode in ring scan ring256 scan; do
  unset NMN_NO_RING_MASKED NMN_RING_MASK_WGS
  if [ $mode = scan ]; then export NMN_NO_RING_MASKED=1; fi
  if [ $mode = ring256 ]; then export NMN_RING_MASK_WGS=256; fi
  timeout 120 python bench.py $COMMON --mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mask $m %-7s %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f  certified %s' % ('$mode', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified'] if d['parity'] else None))"
done; done
