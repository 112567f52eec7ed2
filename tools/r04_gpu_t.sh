#!/bin/bash
# round 4: 768 scan waves for sweeps of 0.25 .. 2 GiB: parity subset, config 2, host-API latency, concurrent callers at 1M
OUT=$PWD/gpurun_out/r04t; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_i8_mirror.py tests/test_gpu_fullsize.py tests/test_gpu_coalesce.py -x -q -m gpu -k "not config4 and not 10M and not config3 and not config5" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
{
for sw in 768 0; do
B="--rows 1000000 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --no-live-pmc --no-parity --no-mirror-legs --steps 300 --warmup 20"
NMN_SCAN_WAVES_SMALL=$sw python bench.py $B --callers 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d.get('concurrent_callers') or {}
print('small_waves $sw: 1M x 768  %9.1f q/s  %.4f ms/step  kernel %.4f ms  certified %s | 16 callers %s q/s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], d['parity'], c.get('value')))"
NMN_SCAN_WAVES_SMALL=$sw python tools/latency_probe.py 400000:768:100 1000000:768:100 2000000:768:100 1000000:1536:100 3000000:256:10 2>&1 | grep -v amdgpu | sed "s/^/small_waves $sw: /"
done
} > $OUT/small_waves.txt 2>&1
cat $OUT/small_waves.txt
