O=gpurun_out/r05z2; mkdir -p $O
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --rebuilds 1 --dim 1536 --metric euclidean --k 1000"
for i in 1 2; do
for m in 0.5 0.1 0.01; do
for v in ch24 ch12; do
if [ $v = ch12 ]; then export NMN_SCAN_MASKED_CH12=1; else unset NMN_SCAN_MASKED_CH12; fi
python bench.py $COMMON --mask $m --steps $( [ $m = 0.5 ] && echo 12 || echo 30 ) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mask $m %-5s %8.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.4f  step_frac %.4f certified %s' % ('$v', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['step_priced_as_survey_8d_frac'], d['parity']['exact_topk_certified']))"
done; done; done > $O/masked_ch24_ab.txt 2>&1
unset NMN_SCAN_MASKED_CH12
timeout 600 python -m pytest tests/test_gpu_filter.py tests/test_gpu_edge_cases.py -x -q 2>&1 | grep "passed\|failed" >> $O/masked_ch24_ab.txt
cat $O/masked_ch24_ab.txt
