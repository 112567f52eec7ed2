#!/bin/bash
# ABAB of library variants / workgroup counts with tools/mfma_loop.py: args "variant:wgs" ..., NQ env (default 64)
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-2}); do
for vs in "$@"; do
  v=${vs%%:*}; w=${vs##*:}
  if [ $v = default ]; then unset NEUMANN_GPU_LIB; else export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_$v.so; fi
  if [ "$w" = "-" ] || [ "$w" = "$vs" ]; then unset NMN_MFMA_WGS; else export NMN_MFMA_WGS=$w; fi
  python tools/mfma_loop.py --nq ${NQ:-64} --tag $v ${SHAPES:-10000000:768} 2>/dev/null
done; done
