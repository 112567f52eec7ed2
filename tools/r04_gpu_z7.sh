#!/bin/bash
# after the fix (rank sort only for lists of <= one entry per thread): bench's next-rows legs, clustered IVF kernel times, latency, parity
OUT=$PWD/gpurun_out/r04z7; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
run() { tag=$1; shift; env "$@" python bench.py --next-rows-child 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'filtered', round(d['filtered_similar_sel0.1']['ms_per_query_wall'],4), 'ivf', round(d['ivf_probe']['ms_per_query_wall'],4), round(d['ivf_probe']['ms_per_query_wall_32_per_call'],4), round(d['ivf_probe']['ms_per_query_wall_128_per_call'],4))"; }
{
run current A=1
run bitonic_final NEUMANN_GPU_LIB=$V/libneumann_gpu_sort_bitonic.so
run current A=1
python tools/latency_probe.py 1000000:768:100 1000000:768:1000 2>&1 | grep -v amdgpu
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
