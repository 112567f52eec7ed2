O=gpurun_out/r05z0; mkdir -p $O
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --steps 20 --rebuilds 2"
for i in 1 2 3; do
for v in default strided; do
if [ $v = strided ]; then export NMN_SCAN_STRIDED=1; else unset NMN_SCAN_STRIDED; fi
python bench.py $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %7.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.4f  alone %.4f  certified %s draws %s' % ('$v', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['avg_kernel_ms_alone'], d['parity']['exact_topk_certified'], ['%.1f'%x for x in d['rebuilds']['queries_per_s']]))"
done
done > $O/strided_f32.txt 2>&1
unset NMN_SCAN_STRIDED
cat $O/strided_f32.txt
