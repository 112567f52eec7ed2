#!/bin/bash
OUT=$PWD/gpurun_out/r04p; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
{
for v in timing; do
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so python tools/i8b_timing.py $v
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so NMN_I8B_WAVES=2048 python tools/i8b_timing.py ${v}_w2048
done
} 2>&1 | grep -v amdgpu.ids > $OUT/timing.txt
cat $OUT/timing.txt
