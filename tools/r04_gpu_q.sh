#!/bin/bash
OUT=$PWD/gpurun_out/r04q; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_i8_mirror.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
V=$R/neumann_amd/lib/variants
{
python tools/mfma_loop.py --nq 64 --reps 30 --realloc 2 --tag i8b_refine
NMN_NO_REFINE=1 python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_1launch
NMN_I8B_WAVES=2048 python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_w2048
NMN_NO_I8B=1 python tools/mfma_loop.py --nq 64 --reps 30 --tag ring
for v in nostore; do
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so python tools/mfma_loop.py --nq 64 --reps 30 --tag $v
done
I8B_WG_WAVES=4 NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_timing.so python tools/i8b_timing.py timing
python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag i8b_l2
NMN_NO_I8B=1 python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag ring_l2
} 2>&1 | grep -v amdgpu.ids > $OUT/ab3.txt
cat $OUT/ab3.txt
