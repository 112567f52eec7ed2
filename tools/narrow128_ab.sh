#!/bin/bash
# rows of 128 elements on the 8-bit sweep: eight lanes per row (default) against the 16-lane mapping (variant wide128) and the bf16 mirror
cd ${GRAFT_REPO_ROOT:-$PWD}
for r in 1 2; do for v in default wide128 bf16; do
  lib=""; m=1; [ "$v" = wide128 ] && lib=$PWD/neumann_amd/lib/variants/libneumann_gpu_wide128.so; [ "$v" = bf16 ] && m=2
  NEUMANN_GPU_LIB=$lib python bench.py --rows 10000000 --dim 128 --mirror $m --steps 60 --warmup 6 --rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('10M x 128 %-8s round $r  %8.1f q/s  kernel %.4f ms  frac %.3f  bytes/elem %d  certified %s' % ('$v', d['value'], r['avg_kernel_ms'], r['frac'], r['bytes_per_corpus_element'], d['parity']['exact_topk_certified']))"
done; done
