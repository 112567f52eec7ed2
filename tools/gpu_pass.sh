O=gpurun_out/r05zb; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ring.py tests/test_gpu_parity_basic.py tests/test_gpu_edge_cases.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config2 or config3 or config5" 2>&1 | grep -E "passed|failed" >> $O/tests.txt
COMMON="--no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 3 --rebuilds 2"
for i in 1 2; do
for v in ring noring; do
if [ $v = noring ]; then export NMN_NO_RING=1; else unset NMN_NO_RING; fi
for cfg in "--steps 20" "--rows 1000000 --steps 100" "--dim 1536 --metric euclidean --k 1000 --steps 12" "--rows 30000000 --dim 128 --steps 20"; do
python bench.py $COMMON $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-7s %-50s %7.1f q/s  %.4f ms/step  kernel %.4f ms  frac %.4f  alone %.4f  certified %s' % ('$v', d['config']['workload'][:50], d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['avg_kernel_ms_alone'], d['parity']['exact_topk_certified']))"
done; done; done > $O/ring_ab.txt 2>&1
unset NMN_NO_RING
cat $O/tests.txt $O/ring_ab.txt
