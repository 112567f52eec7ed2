# quick timing probe used during kernel tuning: bash tools/ab_scan.sh  (on the GPU box)
for i in 1 2; do
  python bench.py --rows 1000000 --steps 200 --warmup 20 --no-cpu-baseline --no-parity | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1M  step_ms=%.4f scan_ms=%.4f pipe_ms=%.4f q/s=%.1f'%(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['pipeline_ms_per_query_batch'], d['value']))"
done
python bench.py --steps 30 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('10M step_ms=%.4f scan_ms=%.4f pipe_ms=%.4f q/s=%.1f'%(d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['pipeline_ms_per_query_batch'], d['value']), d['parity'])"
