#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (rocpd sqlite) per kernel and derive HBM traffic per launch.

    python tools/pmc_summary.py FETCH.db WRITE.db "title" out.json > out.txt

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read, so read bytes of the scan
kernel = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is taken as is (calibrated here on synth_fill_kernel, which
writes exactly rows*dim*4 bytes).
"""
import json
import sqlite3
import sys


def table(path, counter):
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "where counter_name=? group by kernel_name order by 3 desc")
    return list(db.execute(q, (counter,)))


def main():
    fetch_db, write_db, title, out_json = sys.argv[1:5]
    f = table(fetch_db, "FETCH_SIZE")
    w = table(write_db, "WRITE_SIZE")
    print(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only): {title}")
    print(f"# {'kernel':<80} {'counter':<11} {'launches':>8} {'avg_KiB':>14} {'min_KiB':>14} {'max_KiB':>14}")
    for name, rows in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
        for k, n, a, mn, mx in rows:
            print(f"{k[:80]:<82} {name:<11} {n:>8} {a:>14.1f} {mn:>14.1f} {mx:>14.1f}")
    # the sweeps proper: nmn::scan_kernel<...> (NOT exact_scan_kernel, whose name contains the same substring and which
    # reads the f32 corpus: that mismatch once printed "read 30.825 GB" under a 15.40 GB table) and scan_mfma_kernel
    def pick(rows, needle):
        return [r for r in rows if needle in r[0] and "exact_scan" not in r[0]]
    fill_w = next((r for r in w if "synth_fill" in r[0]), None)
    out = {"title": title, "kernels": []}
    for needle in ("::scan_kernel<", "scan_mfma_kernel<"):
        for rf in pick(f, needle):
            rw = next((r for r in pick(w, needle) if r[0] == rf[0]), None)
            # a template that serves both real sweeps and launches that return at once (the f32 retry of a mirror pass, the
            # sampling pass of a batch): the sweeps are the launches at the maximum
            rd = rf[4] * 1024 * 2
            wr = (rw[4] if rw else 0.0) * 1024
            out["kernels"].append({"kernel": rf[0], "launches": rf[1], "fetch_size_kib_max": rf[4], "fetch_size_kib_avg": rf[2],
                                   "write_size_kib_max": rw[4] if rw else None,
                                   "read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
                                   "hbm_bytes_per_launch": rd + wr,
                                   "correction": "FETCH_SIZE*1024*2 (gfx950 half-count of 16 B/lane streams) + WRITE_SIZE*1024; "
                                                 "max over the launches of the template (= the full sweeps)"})
            print(f"# {rf[0][:70]}: HBM traffic of the LARGEST launch of this template: read {rd / 1e9:.3f} GB (corrected) + write {wr / 1e9:.3f} GB")
            # (a query batch of more than 64 queries is a sampling pass + TWO launches of the main sweep, of up to 64 one + one:
            #  the per-batch figure is the sum over the template's launches divided by the batches)
            print(f"#   sum over its {rf[1]} launches: read {rf[2] * rf[1] * 1024 * 2 / 1e9:.3f} GB + write {(rw[2] * rw[1] if rw else 0.0) * 1024 / 1e9:.3f} GB")
    if fill_w:
        out["write_calibration_synth_fill_bytes"] = fill_w[2] * 1024
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
