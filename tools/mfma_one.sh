#!/bin/bash
# one line of the matrix-core sweep: tools/mfma_one.sh rows dim nq [metric]
python bench.py --rows $1 --dim $2 --batched $3 --metric ${4:-cosine} --steps 10 --no-other-configs --no-cpu-baseline --callers 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['batched']; print('rows=$1 dim=$2 nq=$3 ${4:-cosine} q/s=%.0f step_ms=%.3f sweep_ms=%.3f certified=%s' % (b['value'], b['ms_per_step'], b['sweep_ms_incl_sampling_pass'], b['exact_topk_certified_3_of_batch']))"
