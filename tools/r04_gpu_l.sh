#!/bin/bash
# round 4: Euclidean exact sum with prefetch: parity (Euclidean-heavy tests) + config 5 tail
OUT=$PWD/gpurun_out/r04l; mkdir -p $OUT; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_ivf.py tests/test_gpu_i8_mirror.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
B="--rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs --dim 1536 --metric euclidean --k 1000"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OUT/trace -o t -- python $R/bench.py $B --streams 1 --steps 12 --warmup 3 --mask 0.1 > /dev/null 2>&1)
python tools/trace_gantt.py $(find $OUT/trace -name "*.db" | head -1) --kernel scan_i8_kernel --skip 6 --steps 1 > $OUT/gantt_config5_mask0.1_1stream.txt 2>&1; rm -rf $OUT/trace
cat $OUT/gantt_config5_mask0.1_1stream.txt
for m in 0.1 0.01; do
  python bench.py --rebuilds 2 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --dim 1536 --metric euclidean --k 1000 --steps 30 --mask $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('config 5 mask $m  %9.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.3f  certified %s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified']))"
done
python bench.py --next-rows-child 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('filtered', d['filtered_similar_sel0.1']['ms_per_query_wall'], 'ivf', d['ivf_probe']['ms_per_query_wall'], d['ivf_probe']['ms_per_query_wall_32_per_call'])"
