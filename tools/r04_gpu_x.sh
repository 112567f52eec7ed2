#!/bin/bash
# round 4: scan_i8b_kernel by query groups (NG = ceil(nq / 16)): batches of 3 .. 64 queries against the LDS-ring kernel, 10M x 768
OUT=$PWD/gpurun_out/r04x; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_batched.py -x -q -m gpu -k "queries_in_lds" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
{
for nq in 4 8 16 24 32 48 64; do
NMN_I8B=1 python tools/mfma_loop.py --nq $nq --reps 30 --tag i8b_nq$nq
python tools/mfma_loop.py --nq $nq --reps 30 --tag ring_nq$nq
done
for nq in 16 32; do
NMN_I8B=1 python tools/mfma_loop.py --nq $nq --reps 30 --metric 1 --tag i8b_l2_nq$nq
python tools/mfma_loop.py --nq $nq --reps 30 --metric 1 --tag ring_l2_nq$nq
done
} 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
