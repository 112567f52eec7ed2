# timing probe of the batched path: bash tools/ab_batch.sh (on the GPU box)
for rep in 1 2; do for nq in 64 16 8; do
python bench.py --nq $nq --steps 12 --warmup 2 --streams 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nq=%d 10M: q/s=%.0f step_ms=%.3f scan_ms=%.3f GB/s=%.0f frac=%.3f' % (d['config']['nq'], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))"
done; done
