O=gpurun_out/r05k; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_edge_cases.py tests/test_gpu_filter.py tests/test_gpu_i8_mirror.py -x -q 2>&1 | tail -3 > $O/tests.txt
for a in "--rows 1000000 --dim 768" "--rows 10000000 --dim 768" "--rows 10000000 --dim 1536 --metric 1 --k 1000 --mirror 0"; do
echo "== $a"; NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_seltrace.so python tools/search_child.py $a --api host --reps 4 2>/dev/null | grep "^select" | tail -4
done > $O/select_phases.txt
cat $O/tests.txt $O/select_phases.txt
