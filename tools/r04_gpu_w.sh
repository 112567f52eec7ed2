#!/bin/bash
# round 4: streams in flight vs the exposed tail: config 5 under bitmaps (TOP-1000: a ~0.2-ms tail of single-workgroup kernels), config 2
OUT=$PWD/gpurun_out/r04w; mkdir -p $OUT
{
for m in 0.1 0.01 0.5; do
for s in 2 3 4 6; do
  python bench.py --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --no-parity --dim 1536 --metric euclidean --k 1000 --steps 60 --mask $m --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('config 5 mask $m streams $s  %9.1f q/s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']))"
done; done
for s in 2 3 4 6; do
  python bench.py --rows 1000000 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --no-parity --steps 300 --warmup 20 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('config 2 streams $s  %9.1f q/s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']))"
done
} > $OUT/streams.txt 2>&1
cat $OUT/streams.txt
