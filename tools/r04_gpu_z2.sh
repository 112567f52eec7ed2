#!/bin/bash
# round 4: select_kernel's row gather by 16-byte loads: parity (the suites that exercise the selection hardest), phases, latency A/B
OUT=$PWD/gpurun_out/r04z2; mkdir -p $OUT; R=$PWD
timeout 1800 python -m pytest tests/test_gpu_parity_basic.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_i8_mirror.py tests/test_gpu_batched.py tests/test_gpu_filter.py tests/test_gpu_ivf.py tests/test_gpu_engine.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
V=$R/neumann_amd/lib/variants
{
for rep in 1 2; do
python tools/latency_probe.py 1000000:768:100 10000000:768:100 2>&1 | grep -v amdgpu | sed "s/^/rows16: /"
NEUMANN_GPU_LIB=$V/libneumann_gpu_sel_dword.so python tools/latency_probe.py 1000000:768:100 10000000:768:100 2>&1 | grep -v amdgpu | sed "s/^/dword:  /"
done
} > $OUT/latency_ab.txt
cat $OUT/latency_ab.txt
bash tools/r04_gpu_y2.sh > /dev/null 2>&1; grep -v amdgpu gpurun_out/r04y/select_phases.txt | grep "^select W=745" | head -3; grep -v amdgpu gpurun_out/r04y/select_phases.txt | grep "^select W=4007" | head -3
cp gpurun_out/r04y/select_phases.txt $OUT/select_phases_rows16.txt
