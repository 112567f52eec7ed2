#!/bin/bash
# HBM bytes of the masked 8-bit sweep (survivor walk) against the bytes of the rows its bitmap keeps: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
# separate passes, kernel trace only; 10M x 1536 Euclidean TOP-1000 (config 5) at selectivity 0.5 / 0.1 / 0.01
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export TMPDIR=/tmp
for sel in 0.5 0.1 0.01; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/mpmc
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/mpmc -o p -- python $R/tools/search_child.py --rows 10000000 --dim 1536 --metric 1 --k 1000 --mask $sel --mirror 1 --api device --reps 4 > /tmp/mpmc.json 2>/dev/null)
  DB=$(find /tmp/mpmc -name "*.db" | head -1)
  python - "$DB" /tmp/mpmc.json $c <<'PY'
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); info = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = sys.argv[3]
rows = list(db.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)))
scan = [v for n, v in rows if "scan_i8_kernel" in n]
big = [x for x in scan if x * 2 >= max(scan)]
scale = 1024 * (2 if c == "FETCH_SIZE" else 1)   # KiB; gfx950: FETCH_SIZE reports half the bytes of 16-B-per-lane reads
b = sum(big) / len(big) * scale
kept = info["kept_rows"] * info["dim"]
print(f"selectivity {info['mask']}: {c} of scan_i8_kernel (masked, survivor walk) {b/1e9:.4f} GB per sweep over {len(big)} sweeps; the kept rows' codes: {kept/1e9:.4f} GB -> {b/kept:.3f}x")
PY
done; done
