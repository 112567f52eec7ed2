#!/bin/bash
# 64 / 128 queries per step on the default library: bash tools/batched_ab.sh  (prints ms per step and the sweep's event time)
cd ${GRAFT_REPO_ROOT:-$PWD}
for nq in 64 128; do for r in 1 2 3; do
python bench.py --nq $nq --steps 30 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('10M x 768 nq $nq round $r  %8.1f q/s  ms/step %.4f  sweep %.4f ms  frac %.3f  certified %s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity']['exact_topk_certified']))"
done; done
