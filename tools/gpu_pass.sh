#!/bin/bash
# scratch: 8-bit batched sweep without the hand-placed scheduling fences
R=$PWD
for round in 1 2 3; do
for v in default nofence nofence_burst; do
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  python tools/mfma_loop.py --nq 64 --reps 12 --realloc 2 --mirror 1 --tag $v 10000000:768 2>/dev/null
done; done
