#!/bin/bash
# round 4: the launches of one filtered SIMILAR call (10M x 768, selectivity 0.1, host buffers)
OUT=$PWD/gpurun_out/r04z5; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace -o f -- python $R/tools/filtered_trace_child.py > /dev/null 2> $OUT/child.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python - "$DB" > $OUT/filtered_similar_launches.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'pred_eval' in r[2]]
a, b = idx[-2], idx[-1]
seg = rows[a - 1: b - 1]
t0 = seg[0][0]
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:100]}")
print("launches", len(seg), "span us", (seg[-1][1] - t0) / 1e3)
PY
rm -rf $OUT/trace
cat $OUT/filtered_similar_launches.txt
