#!/usr/bin/env python
"""Reproduce bench.py's ms_per_step from a rocprofv3 --kernel-trace (rocpd sqlite) of the SAME command, overlap included.

    rocprofv3 --kernel-trace -d OUT -o t -- python bench.py --steps K --warmup W [--streams 2] ...
    python tools/trace_timeline.py OUT/**/t_results.db --steps K --warmup W [--kernel scan_kernel] > profiles/rNN_timeline.txt

bench.py pipelines its steps over two HIP streams, so consecutive sweeps overlap at head and tail and the per-kernel
averages of `--stats` add up to MORE than the wall time of the loop.  This script takes the dispatches of the dominant
kernel in start order, drops the W warm-up sweeps, takes the next K (the timed region: bench.py fences before and after
it) and reports for that window
  * span          first timed sweep's start -> end of the last kernel of the window        (= K * ms_per_step)
  * busy union    time during which at least one kernel of the window was running
  * sum           sum of the kernel durations (what --stats averages)  ->  overlap = sum - union
so that  span / K  is directly comparable with the JSON line's ms_per_step, and  sum(dominant) / K  with
roofline.avg_kernel_ms (which is measured with the steps serialised by the event reads).
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--kernel", default="scan_kernel", help="substring of the dominant kernel's name")
    ap.add_argument("--min-us", type=float, default=100.0,
                    help="dispatches of the dominant kernel shorter than this are not sweeps (the f32 retry launches "
                         "of a mirror pass return at once)")
    ap.add_argument("--skip", type=int, default=0, help="sweeps to skip before the warm-up (earlier legs of the run)")
    ap.add_argument("--title", default="")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    ker = list(db.execute("select name, start, end from kernels order by start"))
    sweeps = [(s, e) for n, s, e in ker if a.kernel in n and (e - s) / 1e3 >= a.min_us]
    first = a.skip + a.warmup
    if len(sweeps) < first + a.steps:
        raise SystemExit(f"only {len(sweeps)} sweeps of '{a.kernel}' in the trace, need {first + a.steps}")
    win = sweeps[first:first + a.steps]
    t0 = win[0][0]
    t_next = sweeps[first + a.steps][0] if len(sweeps) > first + a.steps else None
    # every kernel that starts inside the window (the tail of the last step: select / rescore / final)
    inside = [(n, s, e) for n, s, e in ker if s >= t0 and (t_next is None or s < t_next)]
    # the timed region ends with a device synchronise: kernels of the NEXT leg start only after it, so cut at the first
    # gap that follows the last timed sweep's own pipeline (5 kernels later at most)
    last_end = win[-1][1]
    tail = [x for x in inside if x[1] >= win[-1][0]]
    for n, s, e in tail:
        if s - last_end > 200e3:  # > 200 us of idle device: the host has fenced
            break
        last_end = max(last_end, e)
    inside = [x for x in inside if x[2] <= last_end]
    span = last_end - t0
    iv = sorted((s, e) for _, s, e in inside)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    total = sum(e - s for _, s, e in inside)
    dom = sum(e - s for s, e in win)
    pair_overlap = sum(max(0, min(win[i][1], win[i + 1][1]) - win[i + 1][0]) for i in range(len(win) - 1))
    K = a.steps
    print(f"# timeline of the timed region, from the rocprofv3 kernel trace: {a.title or a.db}")
    print(f"# dominant kernel '{a.kernel}': {len(sweeps)} sweeps in the trace, window = sweeps [{first}, {first + K})")
    print(f"span_ms_total            {span / 1e6:10.3f}    span / steps = {span / 1e6 / K:.4f} ms   <- compare: ms_per_step")
    print(f"busy_union_ms            {union / 1e6:10.3f}    device idle inside the window: {(span - union) / 1e6:.3f} ms")
    print(f"sum_all_kernels_ms       {total / 1e6:10.3f}    overlapped (sum - union): {(total - union) / 1e6:.3f} ms")
    print(f"sum_dominant_ms          {dom / 1e6:10.3f}    avg per sweep = {dom / 1e6 / K:.4f} ms (overlapping sweeps share HBM: "
          f"longer than a sweep alone)")
    print(f"sweep_pair_overlap_ms    {pair_overlap / 1e6:10.3f}    consecutive sweeps running at the same time, per step "
          f"{pair_overlap / 1e6 / max(K - 1, 1):.4f} ms")
    print(f"queries_per_s_from_trace {K / (span / 1e9):10.1f}    (one query per step)")
    names = {}
    for n, s, e in inside:
        d = names.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += e - s
    print("# kernels inside the window")
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:96]:<98} {c:>5} launches {t / 1e3 / c:>10.1f} us avg {t / 1e6:>10.3f} ms total")


if __name__ == "__main__":
    main()
