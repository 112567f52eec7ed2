for rep in 1 2; do for st in 1 2 3; do for rows in 1000000 10000000; do
python bench.py --rows $rows --streams $st --steps $([ $rows = 1000000 ] && echo 300 || echo 40) --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$st rows=%-9d step_ms=%.4f scan_ms=%.4f q/s=%.1f cert=%s'%(d['config']['rows_total'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['value'], d['parity']['exact_topk_certified']))"
done; done; done
