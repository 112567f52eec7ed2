#!/bin/bash
# rows of an odd number of 128-element halves on the 8-bit mirror (--mirror 1) against the bf16 mirror (--mirror 2): one query per step
cd ${GRAFT_REPO_ROOT:-$PWD}
for dim in 384 640 896; do for m in 1 2; do
python bench.py --rows 10000000 --dim $dim --mirror $m --steps 40 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('10M x $dim mirror $m: %8.1f q/s  kernel %.4f ms  frac %.3f  bytes/elem %d  certified %s' % (d['value'], r['avg_kernel_ms'], r['frac'], r['bytes_per_corpus_element'], d['parity']['exact_topk_certified']))"
done; done
