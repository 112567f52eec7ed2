#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) --kernel-trace output as a per-kernel table.

    python tools/prof_summary.py gpurun_out/prof/x_results.db [title] > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = list(db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
        "from kernels group by name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary: {title}")
    print(f"# (full = the launches of at least half the template's longest: a template also serves launches that return at once — the f32 retry")
    print(f"#  behind a mirror sweep, the sampling pass of a batch — which the plain average mixes in)")
    print(f"# {'kernel':<100} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_us':>12} {'pct':>6} {'full':>5} {'full_avg_us':>12}")
    for name, n, avg, mn, mx, tot in rows:
        full = [r[0] for r in db.execute("select end-start from kernels where name=? and (end-start)*2 >= ?", (name, mx))]
        print(f"{name[:100]:<102} {n:>6} {avg / 1e3:>10.1f} {mn / 1e3:>10.1f} {mx / 1e3:>10.1f} {tot / 1e3:>12.1f} "
              f"{100.0 * tot / total:>6.2f} {len(full):>5} {sum(full) / max(len(full), 1) / 1e3:>12.1f}")


if __name__ == "__main__":
    main()
