#!/bin/bash
# round 4: select phases (instrumented variant), what the rare-path launches cost the device API at 1M rows, the trimmed ingest kernel, suite durations
OUT=gpurun_out/r04g; mkdir -p $OUT
bash tools/r04_gpu_f.sh > /dev/null 2>&1; cp gpurun_out/r04f/select_phases.txt $OUT/; grep -a "^==\|select W" $OUT/select_phases.txt | awk '/^==/{print; n=0; next} {n++; if (n>=3 && n<=5) print}' | cut -c1-260
B="python bench.py --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 5 --rebuilds 3 --steps 300"
for cfg in "--rows 1000000" "--rows 2000000" "--rows 4000000"; do
  for knob in default NMN_MEASURE_DEVICE_SHORT_CHAIN; do
    if [ $knob = default ]; then unset NMN_MEASURE_DEVICE_SHORT_CHAIN; else export NMN_MEASURE_DEVICE_SHORT_CHAIN=1; fi
    $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg %-32s %9.1f q/s  %.4f ms/step  kernel %.4f ms  certified %s' % ('$knob', d['value'], d['ms_per_step'], r['avg_kernel_ms'], d['parity']['exact_topk_certified']))" | tee -a $OUT/device_short_chain_ab.txt
  done
done
unset NMN_MEASURE_DEVICE_SHORT_CHAIN
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_fill -o fill -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
from neumann_amd import GpuFlatIndex
for d, n in ((768, 10_000_000), (1536, 5_000_000), (128, 10_000_000), (2048, 3_000_000)):
    idx = GpuFlatIndex(d, n, device=0); idx.fill_synthetic(3, n); idx.close()
" > /dev/null 2>&1)
python tools/prof_summary.py $OUT/prof_fill/fill_results.db "fill_synthetic 10M x 768, 5M x 1536, 10M x 128, 3M x 2048 (ingest_q8_kernel, packed math)" > $OUT/fill_kernels.txt 2>&1; grep ingest $OUT/fill_kernels.txt | cut -c1-170
( time timeout 1800 python -m pytest tests -x -q -m gpu --durations=25 ) > $OUT/pytest.log 2>&1; tail -34 $OUT/pytest.log
