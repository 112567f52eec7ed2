#!/bin/bash
OUT=$PWD/gpurun_out/r04p; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
{
for v in timing timing_nomfma timing_noloads; do
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so NMN_NO_REFINE=1 python tools/i8b_timing.py $v
done
} 2>&1 | grep -v amdgpu.ids > $OUT/timing.txt
cat $OUT/timing.txt
