#!/bin/bash
# round 4: scan waves per shard size (nq=1, two streams): a sweep that leaves wave slots free lets the other stream's tail run under it
OUT=$PWD/gpurun_out/r04s; mkdir -p $OUT
{
for rows in 1000000 2000000 4000000 10000000; do
B="--rows $rows --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs --steps 200 --warmup 20"
for w in 4096 2048 1024 768 512 256; do
  NMN_SCAN_WAVES=$w python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rows $rows waves $w  %9.1f q/s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']))"
done; done
} > $OUT/scan_waves_by_rows.txt 2>&1
cat $OUT/scan_waves_by_rows.txt
