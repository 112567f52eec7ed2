#!/bin/bash
# round 4: final_kernel's ordering by runs and ranks instead of a workgroup-wide bitonic sort: the whole GPU suite, latency A/B, trace of the chain
OUT=$PWD/gpurun_out/r04z3; mkdir -p $OUT; R=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
V=$R/neumann_amd/lib/variants
{
for rep in 1 2; do
python tools/latency_probe.py 1000000:768:100 10000000:768:100 1000000:768:1000 2>&1 | grep -v amdgpu | sed "s/^/runs+ranks: /"
NEUMANN_GPU_LIB=$V/libneumann_gpu_sort_bitonic.so python tools/latency_probe.py 1000000:768:100 10000000:768:100 1000000:768:1000 2>&1 | grep -v amdgpu | sed "s/^/bitonic:    /"
done
} > $OUT/latency_ab.txt
cat $OUT/latency_ab.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_host -o host -- python -c "
import sys; sys.path.insert(0,'$R')
from neumann_amd import GpuFlatIndex, synth_rows
idx = GpuFlatIndex(768, 1_000_000, device=0); idx.fill_synthetic(3, 1_000_000)
Q = synth_rows(5, 0, 8, 768)
for i in range(40): idx.search(Q[i % 8], 100, 0)
idx.close()
" > /dev/null 2>&1
DB=$(find $OUT/trace_host -name "*.db" | head -1)
python $R/tools/trace_gantt.py $DB --kernel scan_i8_kernel --skip 20 --steps 2 > $OUT/host_search_chain_gantt.txt 2>&1
rm -rf $OUT/trace_host
head -8 $OUT/host_search_chain_gantt.txt
