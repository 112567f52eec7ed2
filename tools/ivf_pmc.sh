#!/bin/bash
# HBM bytes of an IVF probe's list scan (rocprofv3 --pmc FETCH_SIZE, kernel trace only) against the bytes of the listed rows, with
# the list-major copy and (NMN_IVF_NO_LIST_MAJOR=1) with the bitmap over the id-ordered rows.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for mode in list_major bitmap; do
  [ $mode = bitmap ] && export NMN_IVF_NO_LIST_MAJOR=1
  rm -rf /tmp/ivfpmc_$mode
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/ivfpmc_$mode -o p -- python $R/tools/ivf_pmc_child.py > /tmp/ivfpmc_$mode.json 2>/dev/null
  DB=$(find /tmp/ivfpmc_$mode -name "*.db" | head -1)
  python - "$DB" /tmp/ivfpmc_$mode.json $mode <<'PY'
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); info = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); mode = sys.argv[3]
rows = list(db.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE'"))
scan = [v for n, v in rows if "scan_i8_kernel" in n and ", true," in n.split("<")[1][:12]]   # masked instantiations: <METRIC, true, ...
mask = [v for n, v in rows if "ivf_mask_kernel" in n or "ivf_range_mask_kernel" in n]
probes = info["probes"]
scan_b = sum(scan[-probes:]) / max(1, min(len(scan), probes)) * 1024 * 2      # KiB; gfx950: half the bytes of 16-B-per-lane streaming reads
mask_b = sum(mask[-probes:]) / max(1, min(len(mask), probes)) * 1024 * 2
listed = info["mean_listed_rows"] * info["dim"]                                # int8 codes of the listed rows
print(f"{mode:>10}: list_major_rows {info['list_major_rows']}, listed rows/probe {info['mean_listed_rows']:.0f} = {listed/1e6:.2f} MB of codes; "
      f"list-scan kernel reads {scan_b/1e6:.2f} MB/probe = {scan_b/listed:.3f}x; selection kernel reads {mask_b/1e6:.2f} MB/probe; together {(scan_b+mask_b)/listed:.3f}x")
PY
done
