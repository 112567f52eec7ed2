#!/bin/bash
# matrix-core sweep across row lengths: NQ queries per step (default 64), one line per shape
for shape in "10000000 128" "10000000 256" "8000000 384" "6000000 512" "5000000 640" "10000000 768" "5000000 1024" "4000000 1280" "5000000 1536" "2000000 2048" "2000000 3072" "1500000 4096"; do
  set -- $shape
  python bench.py --rows $1 --dim $2 --batched ${NQ:-64} --metric ${METRIC:-cosine} --steps 10 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('rows=$1 dim=$2 nq=${NQ:-64} q/s=%.0f step_ms=%.3f sweep_ms=%.3f GB/s=%.0f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['roofline']['achieved'], b['exact_topk_certified_3_of_batch']))"
done
