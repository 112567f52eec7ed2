#!/bin/bash
# round 4, final GPU call: the whole gpu suite (incl. config 4 at its full row count), the default bench line (live PMC, all legs),
# kernel traces of the headline loop (2 streams / 1 stream, 8-bit mirror), of the f32-corpus sweep (1 stream: VERDICT r03 #2), of a
# 64-query batch (1 stream: the launches of one batch side by side), the launch chain of a host-buffer search, the cost of a fill,
# PMC of the batched 8-bit sweep.  Outputs: gpurun_out/r04z/* -> copied to profiles/r04z_*.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04z
mkdir -p $O
cd $R
( time timeout 1800 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite_pytest.log 2>&1
echo "gpu suite rc=$?" >> $O/gpu_suite_pytest.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
B="--rebuilds 1 --no-cpu-baseline --no-other-configs --batched 0 --callers 0 --no-live-pmc --no-parity --no-mirror-legs"
cd /tmp && export TMPDIR=/tmp
trace() {  # trace NAME TITLE KERNEL bench-args...
  local name=$1 title=$2 kern=$3; shift 3
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- python $R/bench.py "$@" > $O/bench_traced_$name.json 2> $O/bench_traced_$name.err
  local DB=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB "$title" > $O/kernel_trace_$name.txt 2>&1
  [ -n "$kern" ] && python $R/tools/trace_gantt.py $DB --kernel $kern --skip 20 --steps 2 > $O/gantt_$name.txt 2>&1
  rm -rf $O/trace_$name
}
trace 2streams "bench.py --steps 50 --warmup 5 $B (the headline loop: 2 streams, 8-bit mirror, sweep chain)" scan_i8_kernel --steps 50 --warmup 5 $B
trace 1stream "bench.py --streams 1 --steps 50 --warmup 5 $B (one step on the device at a time, 8-bit mirror)" scan_i8_kernel --streams 1 --steps 50 --warmup 5 $B
trace f32_1stream "bench.py --mirror 0 --streams 1 --steps 30 --warmup 5 $B (SURVEY 8(d): the sweep of the row-major f32 corpus, one step at a time)" scan_kernel --mirror 0 --streams 1 --steps 30 --warmup 5 $B
trace batched64_1stream "bench.py --nq 64 --streams 1 --steps 30 --warmup 5 $B (matrix-core sweep over the 8-bit mirror, one batch at a time)" "" --nq 64 --streams 1 --steps 30 --warmup 5 $B
# the launch chain of a host-buffer search (nmn_index_search) on 1M x 768: the short chain
timeout 300 rocprofv3 --kernel-trace -d $O/trace_host -o host -- python -c "
import sys; sys.path.insert(0,'$R')
from neumann_amd import GpuFlatIndex, synth_rows
idx = GpuFlatIndex(768, 1_000_000, device=0); idx.fill_synthetic(3, 1_000_000)
Q = synth_rows(5, 0, 8, 768)
for i in range(40): idx.search(Q[i % 8], 100, 0)
idx.close()
" > /dev/null 2>&1
DB=$(find $O/trace_host -name "*.db" | head -1)
python $R/tools/trace_gantt.py $DB --kernel scan_i8_kernel --skip 20 --steps 3 > $O/host_search_chain_gantt.txt 2>&1
python $R/tools/prof_summary.py $DB "40 x nmn_index_search(nq=1, k=100) on 1M x 768 (host-buffer API, short chain: 5 launches per search)" > $O/host_search_chain_kernels.txt 2>&1
rm -rf $O/trace_host
# what a fill costs: synth + ONE ingest kernel per shard
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fill -o fill -- python -c "
import sys; sys.path.insert(0,'$R')
from neumann_amd import GpuFlatIndex
for d, n in ((768, 10_000_000), (1536, 5_000_000), (128, 10_000_000), (2048, 3_000_000), (1000, 4_000_000), (200, 4_000_000)):
    idx = GpuFlatIndex(d, n, device=0); idx.fill_synthetic(3, n); idx.close()
" > /dev/null 2>&1
DB=$(find $O/trace_fill -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB "fill_synthetic 10M x 768, 5M x 1536, 10M x 128, 3M x 2048, 4M x 1000 (stride 1024), 4M x 200 (bf16 mirror): one ingest kernel per fill" > $O/ingest_kernels.txt 2>&1
rm -rf $O/trace_fill
# PMC: HBM bytes of the batched 8-bit sweep (separate passes, kernel trace only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python $R/bench.py --nq 64 --steps 6 --warmup 2 $B > /dev/null 2> $O/pmc_$c.err
  DB=$(find $O/pmc_$c -name "*.db" | head -1)
  python - "$DB" $c >> $O/pmc_mfma_i8_10Mx768.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = sys.argv[2]
rows = list(db.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)))
v = [val for name, val in rows if "scan_mfma_kernel" in name]
big = [x for x in v if x * 4 >= max(v)]   # the main sweeps (the sampling pass reads 1/32 of the mirror)
small = [x for x in v if x * 4 < max(v)]
scale = 1024 * (2 if c == "FETCH_SIZE" else 1)   # KiB; gfx950: FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads
nb = max(1, len(small))   # one sampling pass per query batch; the main sweep of a batch is TWO launches (bound refinement)
print(f"{c}: scan_mfma_kernel (I8): per batch of 64 queries: main sweep {sum(big)/nb*scale/1e9:.4f} GB in {len(big)//nb} launches; sampling pass {(sum(small)/nb)*scale/1e9:.4f} GB  ({nb} batches; algorithmic: 7.68 GB per sweep of the 8-bit mirror)")
PY
  rm -rf $O/pmc_$c
done
cd $R
tail -4 $O/gpu_suite_pytest.log; head -8 $O/kernel_trace_f32_1stream.txt | cut -c1-180; head -8 $O/kernel_trace_1stream.txt | cut -c1-180; cat $O/pmc_mfma_i8_10Mx768.txt
head -14 $O/kernel_trace_batched64_1stream.txt | cut -c1-180; cat $O/host_search_chain_gantt.txt | head -12
