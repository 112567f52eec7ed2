#!/bin/bash
# round 4: the launches of one IVF single-query probe (2M x 768, 256 lists, nprobe 8)
OUT=$PWD/gpurun_out/r04y; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o ivf -- python $R/tools/ivf_pmc_child.py > $OUT/child.json 2> $OUT/child.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python - "$DB" > $OUT/ivf_probe_launches.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
# the last probe: walk back from the end to the previous ivf_rank kernel
idx = [i for i, r in enumerate(rows) if 'ivf_rank' in r[2]]
a = idx[-2] if len(idx) >= 2 else 0
b = idx[-1]
# a probe = from the qprep before ivf_rank[a] ... to just before the one preceding ivf_rank[b]
seg = rows[a - 3: b - 3]
t0 = seg[0][0]
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:90]}")
print("launches", len(seg), "span us", (seg[-1][1] - t0) / 1e3)
PY
rm -rf $OUT/trace
cat $OUT/ivf_probe_launches.txt
