#!/usr/bin/env python
"""Kernel start / end times (us, relative) of a few consecutive steps from a rocprofv3 --kernel-trace rocpd database:
python tools/trace_gantt.py DB --kernel scan_i8_kernel --skip 20 --steps 3"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--kernel", default="scan_i8_kernel")
ap.add_argument("--skip", type=int, default=20)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
db = sqlite3.connect(a.db)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
ker = list(db.execute(f"select name, start, end, {q} from kernels order by start"))
sw = [i for i, k in enumerate(ker) if a.kernel in k[0] and (k[2] - k[1]) > 20e3]
i0, i1 = sw[a.skip], sw[a.skip + a.steps]
t0 = ker[i0][1]
for n, s, e, qid in ker[i0:i1]:
    short = n.split("(")[0].split("::")[-1][:28]
    print(f"q{qid!s:>4} {short:28s} start {(s - t0) / 1e3:9.1f}  end {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}")
