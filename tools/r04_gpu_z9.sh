#!/bin/bash
# concurrent callers on small shards (batches ride the matrix-core sweep, two batches in flight): workgroups of the sweep vs the exposed tail
OUT=$PWD/gpurun_out/r04z9; mkdir -p $OUT
{
for rows in 1000000 2000000; do
for wgs in 1024 768 512 384 192 128; do
  NMN_MFMA_WGS=$wgs python bench.py --rows $rows --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --no-live-pmc --no-parity --no-mirror-legs --steps 50 --warmup 5 --callers 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['concurrent_callers']
print('rows $rows mfma_wgs $wgs: 64 callers %9.1f q/s   128 callers %9.1f q/s  (differing answers %s)' % (c['value'], c['with_twice_the_threads']['value'], c['answers_differing_from_a_lone_call']))"
done; done
} > $OUT/callers_small_shards.txt 2>&1
cat $OUT/callers_small_shards.txt
