#!/bin/bash
# round 4, second GPU pass: the tests the first pass did not reach (it stopped at config 4's batch: stale binary), the new drift /
# soak tests, then config 2 with 2 / 3 / 4 streams with and without the sweep chain, and the cost of a fill
set -x
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_i8_mirror.py tests/test_gpu_ivf.py tests/test_gpu_parity_basic.py tests/test_gpu_persist.py tests/test_gpu_sharded.py tests/test_gpu_sharded_handle.py "tests/test_gpu_engine.py::test_interleaved_stores_deletes_and_searches_on_the_8_bit_mirror" -m gpu -q -s --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-mirror-legs --no-live-pmc --warmup 5 --rebuilds 3 --rows 1000000 --steps 300"
for s in 2 3 4; do
  $B --streams $s > $OUT/c2_s${s}_chain.json 2>/dev/null
  NMN_NO_SWEEP_CHAIN=1 $B --streams $s > $OUT/c2_s${s}_nochain.json 2>/dev/null
done
for f in $OUT/c2_*.json; do python - "$f" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['rebuilds']['queries_per_s'])
P
done
python - <<'P' > $OUT/fill_cost.txt 2>&1
import time, torch, os
from neumann_amd import GpuFlatIndex
for rows in (10_000_000,):
    for rep in range(3):
        idx = GpuFlatIndex(768, rows, device=0)
        torch.cuda.synchronize(); t=time.perf_counter(); idx.fill_synthetic(3, rows); torch.cuda.synchronize(); dt=time.perf_counter()-t
        print("fill", rows, rep, round(dt*1e3,2), "ms", idx.hbm_bytes())
        idx.close()
P
cat $OUT/fill_cost.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_fill -o fill -- python -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
from neumann_amd import GpuFlatIndex
idx = GpuFlatIndex(768, 10_000_000, device=0); idx.fill_synthetic(3, 10_000_000); idx.close()
" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls $OUT/prof_fill/* | head; python tools/prof_summary.py $(ls $OUT/prof_fill/*/*.db $OUT/prof_fill/*.db 2>/dev/null | head -1) 2>&1 | head -20 > $OUT/fill_kernels.txt; cat $OUT/fill_kernels.txt
