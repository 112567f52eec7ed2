#!/bin/bash
# round 4: scan_i8b_kernel with the row factors asked for a tile ahead: parity, A/B against the LDS-ring kernel
OUT=$PWD/gpurun_out/r04o; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_i8_mirror.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
V=$R/neumann_amd/lib/variants
{
python tools/mfma_loop.py --nq 64 --reps 30 --realloc 2 --tag i8b
NMN_NO_I8B=1 python tools/mfma_loop.py --nq 64 --reps 30 --realloc 2 --tag ring
NMN_NO_REFINE=1 python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_1launch
NMN_I8B_WAVES=1024 python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_w1024
NMN_I8B_WAVES=2048 python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_w2048
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_noepi.so python tools/mfma_loop.py --nq 64 --reps 30 --tag noepi
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_nostore.so python tools/mfma_loop.py --nq 64 --reps 30 --tag nostore
python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag i8b_l2
NMN_NO_I8B=1 python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag ring_l2
} 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
