#!/bin/bash
# scratch: 2-wave workgroups, two per CU, two query groups per wave (8-bit batched sweep)
R=$PWD
python -m pytest tests/test_gpu_batched.py -x -q -k "i8 or q8 or mirror" 2>&1 | tail -2
for round in 1 2; do
  unset NEUMANN_GPU_LIB NMN_MFMA_WGS
  python tools/mfma_loop.py --nq 64 --reps 16 --realloc 2 --mirror 1 --tag default 10000000:768 2>/dev/null
  export NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_w2.so
  for w in 1024 2048; do
    NMN_MFMA_WGS=$w python tools/mfma_loop.py --nq 64 --reps 16 --realloc 2 --mirror 1 --tag w2_$w 10000000:768 2>/dev/null
  done
done
