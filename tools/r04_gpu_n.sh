#!/bin/bash
# round 4: where the time of the queries-in-LDS batched sweep goes: variant builds (no loads / no MFMA / no epilogue / k-steps loaded one by one),
# HBM traffic and SQ counters of scan_i8b_kernel at 10M x 768, 64 queries
OUT=$PWD/gpurun_out/r04n; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
{
python tools/mfma_loop.py --nq 64 --reps 30 --tag i8b_pairs
for v in single noepi noloads nomfma noloads_noepi nomfma_noepi; do
NEUMANN_GPU_LIB=$V/libneumann_gpu_i8b_$v.so NMN_NO_REFINE=1 python tools/mfma_loop.py --nq 64 --reps 20 --tag $v
done
NMN_NO_REFINE=1 python tools/mfma_loop.py --nq 64 --reps 20 --tag i8b_1launch
NMN_NO_REFINE=1 NMN_I8B_NT=1 python tools/mfma_loop.py --nq 64 --reps 20 --tag i8b_nt_1l
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    rm -rf /tmp/bpmc
    NMN_NO_REFINE=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/bpmc -o p -- python $R/tools/batch_pmc_child.py 64 > /dev/null 2> /tmp/bpmc.err
    DB=$(find /tmp/bpmc -name "*.db" | head -1)
    [ -z "$DB" ] && { echo "set [$set]: no db ($(tail -1 /tmp/bpmc.err))" >> $OUT/counters.txt; continue; }
    python - "$DB" >> $OUT/counters.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select kernel_name, counter_name, value from counters_collection"))
agg = collections.defaultdict(list)
for n, c, v in rows:
    if "scan_i8b_kernel" in n: agg[c].append(v)
for c, v in sorted(agg.items()):
    big = [x for x in v if x * 2 >= max(v)]   # the main sweeps (the sampling passes are 1/32 of them)
    print(f"{c:28s} main sweep: mean {sum(big)/len(big):.5g}  n={len(big)} of {len(v)}")
PY
done
cat $OUT/counters.txt
