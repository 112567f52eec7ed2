#!/bin/bash
# round 4: the ladder of score-write bounds (one main launch instead of two + a second bound kernel): parity, A/B
OUT=$PWD/gpurun_out/r04v; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_batched.py tests/test_gpu_i8_mirror.py tests/test_gpu_coalesce.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not config4" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
{
for rep in 1 2; do
python tools/mfma_loop.py --nq 64 --reps 30 --realloc 2 --tag ladder
NMN_NO_LADDER=1 python tools/mfma_loop.py --nq 64 --reps 30 --realloc 2 --tag refine
done
NMN_NO_LADDER=1 NMN_NO_REFINE=1 python tools/mfma_loop.py --nq 64 --reps 30 --tag neither
python tools/mfma_loop.py --nq 128 --reps 30 --tag ladder128
NMN_NO_LADDER=1 python tools/mfma_loop.py --nq 128 --reps 30 --tag refine128
python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag ladder_l2 5000000:1536
NMN_NO_LADDER=1 python tools/mfma_loop.py --nq 64 --reps 30 --metric 1 --tag refine_l2 5000000:1536
python tools/mfma_loop.py --nq 16 --reps 30 --tag ladder16
NMN_NO_LADDER=1 python tools/mfma_loop.py --nq 16 --reps 30 --tag refine16
} 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
B="--rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-mirror-legs --nq 64 --steps 30 --warmup 5"
for v in "A=1" "NMN_NO_LADDER=1"; do
env $v python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v batched 64: %9.1f q/s  %.4f ms/step  sweep %.4f ms  frac %.3f certified %s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['frac'], d['parity'].get('exact_topk_certified')))" | tee -a $OUT/ab.txt
done
