#!/bin/bash
# round 4, fourth GPU pass: IVF single query on the short chain, the trimmed ingest_q8_kernel (parity + cost), bounded search_entities
set -x
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_i8_mirror.py tests/test_gpu_parity_basic.py tests/test_gpu_engine.py tests/test_gpu_persist.py tests/test_gpu_edge_cases.py tests/test_gpu_golden.py -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_fill -o fill -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
from neumann_amd import GpuFlatIndex
for d, n in ((768, 10_000_000), (1536, 5_000_000), (128, 10_000_000), (2048, 3_000_000)):
    idx = GpuFlatIndex(d, n, device=0); idx.fill_synthetic(3, n); idx.close()
" > /dev/null 2>&1)
python tools/prof_summary.py $OUT/prof_fill/fill_results.db "fill_synthetic 10M x 768, 5M x 1536, 10M x 128, 3M x 2048 (ingest_q8_kernel, interleaved chains)" > $OUT/fill_kernels.txt 2>&1; cat $OUT/fill_kernels.txt | cut -c1-170
python bench.py --next-rows-child > $OUT/next_rows.json 2> $OUT/next_rows.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04d/next_rows.json').read().strip().splitlines()[-1])
for k,v in d.items(): print(k, {a:b for a,b in v.items() if a in ('value','unit','ms_per_query_wall','seconds','ms_per_query_wall_32_per_call')} if isinstance(v,dict) else v)
P
