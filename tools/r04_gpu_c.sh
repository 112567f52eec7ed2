#!/bin/bash
# round 4, third GPU pass: the whole GPU suite on the fused ingest / one-mirror tree, the mirror decisions of config 4 traced,
# the cost of a fill (kernel trace), config 2 with the sweep's two timing events only, the default bench line
set -x
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
NMN_TRACE_MIRROR=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k config4_80M > $OUT/config4_traced.log 2>&1
grep -a "nmn\] mirror\|config 4 on one\|passed\|failed" $OUT/config4_traced.log | tail -30
python - <<'P' > $OUT/fill_cost.txt 2>&1
import time, torch
from neumann_amd import GpuFlatIndex
for rows, d in ((10_000_000, 768), (5_000_000, 1536), (10_000_000, 128)):
    for rep in range(2):
        idx = GpuFlatIndex(d, rows, device=0)
        torch.cuda.synchronize(); t=time.perf_counter(); idx.fill_synthetic(3, rows); torch.cuda.synchronize(); dt=time.perf_counter()-t
        print("fill", rows, d, rep, round(dt*1e3,2), "ms", idx.hbm_bytes())
        idx.close()
P
cat $OUT/fill_cost.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_fill -o fill -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
from neumann_amd import GpuFlatIndex
for d, n in ((768, 10_000_000), (1536, 5_000_000), (128, 10_000_000)):
    idx = GpuFlatIndex(d, n, device=0); idx.fill_synthetic(3, n); idx.close()
" > /dev/null 2>&1)
python tools/prof_summary.py $OUT/prof_fill/fill_results.db "fill_synthetic 10M x 768, 5M x 1536, 10M x 128 (fused ingest_q8_kernel)" > $OUT/fill_kernels.txt 2>&1; cat $OUT/fill_kernels.txt
B="python bench.py --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 5 --rebuilds 3 --rows 1000000 --steps 300"
$B > $OUT/c2_s2.json 2>/dev/null; $B --streams 3 > $OUT/c2_s3.json 2>/dev/null
for f in $OUT/c2_*.json; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['rebuilds']['queries_per_s'])
P
done
# the launch chain of a host-buffer search (nmn_index_search) on 1M x 768: kernel trace of 40 calls, the timeline of three of them
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/prof_host -o host -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
import numpy as np
from neumann_amd import GpuFlatIndex, synth_rows
idx = GpuFlatIndex(768, 1_000_000, device=0); idx.fill_synthetic(3, 1_000_000)
Q = synth_rows(5, 0, 8, 768)
for i in range(40): idx.search(Q[i % 8], 100, 0)
idx.close()
" > /dev/null 2>&1)
python tools/trace_gantt.py $OUT/prof_host/host_results.db --kernel scan_i8_kernel --skip 20 --steps 3 > $OUT/host_chain_gantt.txt 2>&1; cat $OUT/host_chain_gantt.txt
python tools/prof_summary.py $OUT/prof_host/host_results.db "40 x nmn_index_search(nq=1, k=100) on 1M x 768 (host-buffer API, short chain)" > $OUT/host_chain_kernels.txt 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
