O=gpurun_out/r05s; mkdir -p $O
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --no-parity --warmup 3 --steps 20 --rebuilds 2"
for i in 1 2; do
for w in 4096 2048 1536 1024 768 512; do
NMN_SCAN_WAVES=$w python bench.py $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('waves %5d  %7.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.4f  alone %.4f  draws %s' % ($w, d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['avg_kernel_ms_alone'], ['%.1f'%x for x in d['rebuilds']['queries_per_s']]))"
done
done > $O/scan_waves_f32.txt 2>&1
cat $O/scan_waves_f32.txt
