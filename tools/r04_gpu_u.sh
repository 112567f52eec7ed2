#!/bin/bash
# round 4: config 5 (10M x 1536 Euclidean TOP-1000 under a bitmap, two streams): scan waves vs the exposed tail
OUT=$PWD/gpurun_out/r04u; mkdir -p $OUT
{
for m in 0.1 0.01 0.5 1.0; do
for w in 4096 3072 2048 1536 1024; do
  NMN_SCAN_WAVES=$w python bench.py --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --no-parity --dim 1536 --metric euclidean --k 1000 --steps 40 --mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('config 5 mask $m waves $w  %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
done; done
} > $OUT/config5_scan_waves.txt 2>&1
cat $OUT/config5_scan_waves.txt
