#!/bin/bash
# scratch: ring kernel, wave maximum by DPP: tests + A/B old/new
R=$PWD
timeout 600 python -m pytest tests/test_gpu_ring.py -x -q 2>&1 | tail -2
for r in 1 2 3; do for e in old new; do
  if [ $e = old ]; then export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_old.so; else unset NEUMANN_GPU_LIB; fi
  python tools/mfma_loop.py --nq 1 --reps 16 --realloc 2 --mirror 0 --tag $e 10000000:768 1000000:768 30000000:128 2>/dev/null
done; done
