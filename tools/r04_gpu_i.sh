#!/bin/bash
# round 4: SQ counters of the masked 8-bit sweep (config 5, selectivity 0.1 and 1.0): is the survivor walk issue-, latency- or occupancy-bound?
OUT=$PWD/gpurun_out/r04i; mkdir -p $OUT
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/avail_names.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  for sel in 0.1 1.0; do
    rm -rf /tmp/sqpmc
    rocprofv3 --pmc $set --kernel-trace -d /tmp/sqpmc -o p -- python $R/tools/mask_pmc_child.py $sel > /dev/null 2> /tmp/sqpmc.err
    DB=$(find /tmp/sqpmc -name "*.db" | head -1)
    [ -z "$DB" ] && { echo "set [$set] sel $sel: no db ($(tail -1 /tmp/sqpmc.err))" >> $OUT/sq_counters.txt; continue; }
    python - "$DB" $sel >> $OUT/sq_counters.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); sel = sys.argv[2]
rows = list(db.execute("select kernel_name, counter_name, value from counters_collection"))
agg = collections.defaultdict(list)
for n, c, v in rows:
    if "scan_i8_kernel" in n: agg[c].append(v)
for c, v in sorted(agg.items()):
    big = sorted(v)[len(v)//2:]   # the sweeps proper (upper half: warm)
    print(f"sel {sel}  {c:28s} median-of-upper {sorted(big)[len(big)//2]:.4g}  n={len(v)}")
PY
  done
done
cat $OUT/sq_counters.txt
