O=gpurun_out/r05f; mkdir -p $O
for i in 1 2; do
for w in 1024 1536 2048 3072 4096; do
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --tag f32_w$w
NMN_SAMPLE_STEP=64 NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --tag f32_w${w}_s64
done
done > $O/wgs_sample_ab.txt 2>&1
for w in 1024 2048 4096; do
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --nq 128 --tag f32_nq128_w$w
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --nq 64 --metric 1 --k 1000 --tag f32_l2_1536_w$w 5000000:1536
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 0 --reps 12 --realloc 2 --nq 64 --tag f32_128d_w$w 30000000:128
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 2 --reps 12 --realloc 2 --nq 64 --tag bf16_w$w
NMN_MFMA_WGS=$w python tools/mfma_loop.py --mirror 1 --reps 12 --realloc 2 --nq 64 --tag i8_w$w
done >> $O/wgs_sample_ab.txt 2>&1
grep -v "^+\|amdgpu.ids" $O/wgs_sample_ab.txt | sort -s -k1,1
