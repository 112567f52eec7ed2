#!/bin/bash
# scratch: f32-rows batched sweep: what the bf16 conversion and the epilogue cost
R=$PWD
for round in 1 2; do
for v in default nowr32 noepi32 tmax1; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python tools/mfma_loop.py --nq 64 --reps 12 --realloc 2 --mirror 0 --tag $v 10000000:768 2>/dev/null
done; done
