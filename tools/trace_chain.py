#!/usr/bin/env python
"""The launch chain of the LAST search in a rocprofv3 --kernel-trace (rocpd sqlite) of tools/search_child.py: every kernel from the
last qprep_kernel (or pred_eval_kernel) on — start (us, relative), duration, gap to the previous kernel's end.

    python tools/trace_chain.py DB [--first qprep_kernel] [--nth-last 1]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--first", default="qprep_kernel")
ap.add_argument("--nth-last", type=int, default=1)
a = ap.parse_args()
db = sqlite3.connect(a.db)
ker = list(db.execute("select name, start, end from kernels order by start"))
starts = [i for i, k in enumerate(ker) if a.first in k[0]]
i0 = starts[-a.nth_last]
i1 = starts[-a.nth_last + 1] if a.nth_last > 1 else len(ker)
t0, prev = ker[i0][1], ker[i0][1]
tot = 0.0
for n, s, e in ker[i0:i1]:
    short = n.split("(")[0].replace("void ", "").replace("nmn::", "").replace("(anonymous namespace)::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  {short}")
    prev = e
    tot += (e - s) / 1e3
print(f"launches {i1 - i0}  span {(prev - t0) / 1e3:.1f} us  busy {tot:.1f} us")
