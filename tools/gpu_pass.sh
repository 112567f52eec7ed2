#!/bin/bash
# scratch: deferred branch-free epilogue: tests + A/B
R=$PWD
timeout 900 python -m pytest tests/test_gpu_batched.py -x -q 2>&1 | tail -2
for round in 1 2 3; do
for v in old new; do
  if [ $v = new ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python tools/mfma_loop.py --nq 64 --reps 12 --realloc 2 --mirror 0 --tag $v 10000000:768 2>/dev/null
  python tools/mfma_loop.py --nq 64 --reps 12 --realloc 1 --mirror 2 --tag $v 10000000:768 2>/dev/null
done; done
