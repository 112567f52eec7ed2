#!/bin/bash
# VALU sweep (NMN_MFMA_MIN_NQ=99) vs matrix-core sweep (NMN_MFMA_MIN_NQ=2) for small query batches
rows=${1:-10000000}; dim=${2:-768}
for nq in ${NQS:-1 2 3 4 5 8}; do
  for m in 99 2; do
    out=$(NMN_MFMA_MIN_NQ=$m python bench.py --rows $rows --dim $dim --nq $nq --steps 30 --warmup 3 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1)
    echo "rows=$rows dim=$dim nq=$nq mfma_min=$m $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("q/s=%.0f ms_per_step=%.3f kernel=%s avg_kernel_ms=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_kernel_ms"]))')"
  done
done
