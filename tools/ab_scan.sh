# same-box A/B of scan-kernel knobs: bash tools/ab_scan.sh (on the GPU box)
run() { env $1 python bench.py --rows $2 --steps $3 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[%-22s] rows=%-9d step_ms=%.4f scan_ms=%.4f GB/s=%.0f q/s=%.1f'%('$1', d['config']['rows_total'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['value']))"; }
for rep in 1 2; do
  for v in "X=1" "NMN_SCAN_NT=0" "NMN_SCAN_WAVES=2048" "NMN_SCAN_WAVES=3072" "NMN_SCAN_WAVES=1024"; do
    run $v 1000000 200
    run $v 10000000 30
  done
done
