#!/bin/bash
OUT=$PWD/gpurun_out/r04p; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
{
for v in timing timing_nostore timing_halfvalu timing_noepi; do
NMN_NO_REFINE=1 I8B_WG_WAVES=4 NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so python tools/i8b_timing.py $v 2>&1 | grep -v "entry times\|kernel span\|first-round"
done
} 2>&1 | grep -v amdgpu.ids > $OUT/timing2.txt
cat $OUT/timing2.txt
