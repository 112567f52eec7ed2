#!/bin/bash
# A/B of the batched 8-bit sweep: env settings per variant, interleaved, NQ=64|128.   bash tools/mfma_i8_ab.sh "name:ENV=1 ENV2=x" ...
cd ${GRAFT_REPO_ROOT:-$PWD}
NQ=${NQ:-64}; ROUNDS=${ROUNDS:-3}; DIM=${DIM:-768}; ROWS=${ROWS:-10000000}; METRIC=${METRIC:-cosine}
for r in $(seq $ROUNDS); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
    env $envs python bench.py --rows $ROWS --dim $DIM --metric $METRIC --nq $NQ --steps 16 --warmup 4 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 \
      --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('variant %-14s nq $NQ round $r  %9.1f q/s  ms/step %.4f  sweep %.4f ms  frac %.3f  bytes/elem %d  cands %s  certified %s' % ('$name', d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['bytes_per_corpus_element'], r['candidates_rescored'], d['parity']['exact_topk_certified']))"
  done
done
