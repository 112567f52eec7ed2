#!/bin/bash
# round 4: the batched sweep's sampling pass ahead of / behind the sweep chain's wait (config 3), two rounds each
OUT=gpurun_out/r04h; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-other-configs --callers 0 --no-mirror-legs --no-live-pmc --rebuilds 2 --steps 20"
for round in 1 2; do
  for knob in default NMN_SAMPLE_AHEAD_OF_CHAIN; do
    if [ $knob = default ]; then unset NMN_SAMPLE_AHEAD_OF_CHAIN; else export NMN_SAMPLE_AHEAD_OF_CHAIN=1; fi
    for nq in 64 128; do
      $B --batched $nq 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batched']
print('nq=$nq %-26s round $round  %9.1f q/s  %.4f ms/step  sweep %.4f ms  frac %.3f  certified %s  | nq=1: %.1f q/s' % ('$knob', b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['roofline']['frac'], b['exact_topk_certified_3_of_batch'], d['value']))" | tee -a $OUT/sampling_pass_ahead_of_chain_ab.txt
    done
  done
done
