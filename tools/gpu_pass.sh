O=gpurun_out/r05m; mkdir -p $O
R=$PWD
python tools/pmc_sq.py --kernel scan_mfma_kernel --title "8-bit mirror, 10M x 768, 64 queries (default: 8 x 16-KiB stages, one workgroup per CU)" -- python $R/tools/mfma_loop.py --mirror 1 --reps 6 > $O/pmc_mfma_i8_default.txt 2>&1
NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_occ2.so python tools/pmc_sq.py --kernel scan_mfma_kernel --title "8-bit mirror, 64 queries, variant -DNMN_MFMA_OCC=2 -DNMN_MFMA_RING_KB=64 (two workgroups per CU, 4 x 16-KiB stages each)" -- python $R/tools/mfma_loop.py --mirror 1 --reps 6 > $O/pmc_mfma_i8_occ2.txt 2>&1
python tools/pmc_sq.py --kernel scan_mfma_kernel --title "8-bit mirror, 128 queries (two query groups per wave)" -- python $R/tools/mfma_loop.py --mirror 1 --nq 128 --reps 6 > $O/pmc_mfma_i8_nq128.txt 2>&1
python tools/pmc_sq.py --kernel scan_mfma_kernel --title "bf16 mirror, 64 queries (4 x 32-KiB stages)" -- python $R/tools/mfma_loop.py --mirror 2 --reps 6 > $O/pmc_mfma_bf16.txt 2>&1
python tools/pmc_sq.py --kernel scan_mfma_kernel --title "f32 rows, 64 queries (4 x 32-KiB stages)" -- python $R/tools/mfma_loop.py --mirror 0 --reps 6 > $O/pmc_mfma_f32.txt 2>&1
for i in 1 2; do
python tools/mfma_loop.py --mirror 1 --reps 12 --realloc 2 --tag i8
NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_occ2.so python tools/mfma_loop.py --mirror 1 --reps 12 --realloc 2 --tag i8_occ2
done > $O/occ2_ab.txt 2>&1
cat $O/pmc_*.txt; grep -v amdgpu.ids $O/occ2_ab.txt
