#!/bin/bash
# 128 queries per step on the 8-bit matrix-core sweep: 128 stationary queries per workgroup (default) against two folded blocks of 64 (NMN_MFMA_I8_NO_128=1)
cd ${GRAFT_REPO_ROOT:-$PWD}
for shape in "--rows 10000000 --dim 768" "--rows 10000000 --dim 256" "--rows 10000000 --dim 512"; do for e in "" "NMN_MFMA_I8_NO_128=1"; do
env $e python bench.py $shape --nq 128 --steps 20 --warmup 4 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$shape nq 128 [$e]: %8.1f q/s  ms/step %.4f  frac %.3f  bytes/elem %d  certified %s' % (d['value'], d['ms_per_step'], r['frac'], r['bytes_per_corpus_element'], d['parity']['exact_topk_certified']))"
done; done
