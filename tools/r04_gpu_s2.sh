#!/bin/bash
# the 768-wave band's edges: pipelined loop (two streams) below 256 MiB and around 2 GiB of sweep
OUT=$PWD/gpurun_out/r04s; mkdir -p $OUT
{
for rows in 100000 200000 300000 2500000 3000000 3500000; do
B="--rows $rows --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs --steps 300 --warmup 20"
for w in 4096 768 384; do
  NMN_SCAN_WAVES=$w python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rows $rows waves $w  %9.1f q/s  %.4f ms/step  kernel %.4f ms' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']))"
done; done
} > $OUT/scan_waves_band_edges.txt 2>&1
cat $OUT/scan_waves_band_edges.txt
