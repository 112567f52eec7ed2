#!/bin/bash
# a few shapes of the matrix-core sweep at 64 and 128 queries (A/B of kernel changes)
for shape in "10000000 768" "10000000 128" "6000000 512" "5000000 1024" "5000000 1536" "2000000 3072"; do
  set -- $shape
  for nq in 64 128; do
  python bench.py --rows $1 --dim $2 --batched $nq --steps 10 --no-other-configs --no-cpu-baseline --callers 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('rows=$1 dim=$2 nq=$nq q/s=%.0f step_ms=%.3f sweep_ms=%.3f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['exact_topk_certified_3_of_batch']))"
  done
done
