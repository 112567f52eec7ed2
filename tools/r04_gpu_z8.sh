#!/bin/bash
# predicate counters in the tail of the packed result block (one D2H per filtered search): parity, the filtered leg
OUT=$PWD/gpurun_out/r04z8; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_filter.py tests/test_gpu_engine.py tests/test_gpu_coalesce.py tests/test_gpu_edge_cases.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for i in 1 2 3; do python bench.py --next-rows-child 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('filtered', round(d['filtered_similar_sel0.1']['ms_per_query_wall'],4), d['filtered_similar_sel0.1']['exact_topk_certified'], 'ivf', round(d['ivf_probe']['ms_per_query_wall'],4), round(d['ivf_probe']['ms_per_query_wall_32_per_call'],4))"; done | tee $OUT/next_rows.txt
