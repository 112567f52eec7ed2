# config 5: 10M x 1536 f32 L2 TOP-1000 with a WHERE-predicate mask, nq=1 (on the GPU box)
for m in 1.0 0.5 0.1; do
python bench.py --dim 1536 --metric euclidean --k 1000 --mask $m --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mask=%s: q/s=%.1f step_ms=%.3f scan_ms=%.3f GB/s=%.0f frac=%.3f' % ('$m', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']), d['parity'])"
done
