#!/bin/bash
# single-query 8-bit sweep per row length (10M rows, or 5M from 1536 up): q/s, kernel time, fraction of peak on its bytes
cd ${GRAFT_REPO_ROOT:-$PWD}
for dim in ${DIMS:-128 256 384 512 640 768 1024 1280 1536 2048 3072 4096}; do
rows=10000000; [ $dim -ge 2048 ] && rows=4000000; [ $dim -ge 4096 ] && rows=2000000
python bench.py --rows $rows --dim $dim --steps 40 --warmup 5 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%8d x %4d: %8.1f q/s  kernel %.4f ms  frac %.3f  bytes/elem %d  cands %s certified %s' % ($rows, $dim, d['value'], r['avg_kernel_ms'], r['frac'], r['bytes_per_corpus_element'], r['candidates_rescored'], d['parity']['exact_topk_certified']))"
done
