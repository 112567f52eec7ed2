#!/usr/bin/env python
"""SQ / GRBM counters of ONE kernel template, per launch, from separate rocprofv3 --pmc passes (--kernel-trace only) around a child
command; the largest launches of the template (the full sweeps) are averaged.

    python tools/pmc_sq.py --kernel scan_mfma_kernel --title "..." -- python tools/mfma_loop.py --mirror 1 --reps 6

Passes (8 SQ slots + 2 GRBM per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  time    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE
  insts   SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
  active  SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

PASSES = {
    "time": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE",
    "insts": "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT",
    "active": "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--title", default="")
    ap.add_argument("--passes", default="time,insts,active")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    env = dict(os.environ, TMPDIR="/tmp")
    vals, dur = {}, None
    for name in a.passes.split(","):
        tmp = tempfile.mkdtemp(prefix="nmn_sq_", dir="/tmp")
        try:
            r = subprocess.run([prof, "--pmc"] + PASSES[name].split() + ["--kernel-trace", "-d", tmp, "-o", "p", "--"] + cmd,
                               capture_output=True, text=True, timeout=600, cwd="/tmp", env=env)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                print(f"# pass {name}: rocprofv3 failed (rc {r.returncode}) {r.stderr[-300:]}")
                continue
            db = sqlite3.connect(dbs[0])
            rows = list(db.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name like ?", (f"%{a.kernel}%",)))
            per = {}
            for did, cn, v in rows:
                per.setdefault(did, {})
                per[did][cn] = per[did].get(cn, 0.0) + v
            if not per:
                print(f"# pass {name}: no dispatch of {a.kernel}")
                continue
            ref = PASSES[name].split()[0]
            big = max(d.get(ref, 0.0) for d in per.values())
            sel = [d for d in per.values() if d.get(ref, 0.0) * 2 >= big]  # the full sweeps (sampling passes / short launches dropped)
            for cn in PASSES[name].split():
                xs = [d[cn] for d in sel if cn in d]
                if xs:
                    vals[cn] = (sum(xs) / len(xs), len(xs))
            if dur is None:
                k = list(db.execute("select end - start from kernels where name like ? order by 1 desc", (f"%{a.kernel}%",)))
                if k:
                    top = [x[0] for x in k if x[0] * 2 >= k[0][0]]
                    dur = sum(top) / len(top) / 1e3
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print(f"# SQ counters per launch of *{a.kernel}* (largest launches averaged; separate --pmc passes, --kernel-trace only): {a.title}")
    print(f"# command: {' '.join(cmd)}")
    if dur:
        print(f"kernel duration under the counters (us, largest launches): {dur:.1f}")
    for cn, (v, n) in vals.items():
        print(f"{cn:<28} {v:>18.0f}   (n={n})")
    g = lambda c: vals.get(c, (0.0, 0))[0]
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        print("# shares of SQ_WAVE_CYCLES: parked (s_waitcnt / barrier) WAIT_ANY %.3f | issue stall WAIT_INST_ANY %.3f (of which LDS issue %.3f) | issuing ACTIVE_INST_ANY %.3f"
              % (g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_ANY") / wc))
        if g("SQ_ACTIVE_INST_VALU"):
            print("# issuing by type / WAVE_CYCLES: VALU(+MFMA) %.3f  LDS %.3f  VMEM %.3f  scalar %.3f  misc (barrier, waitcnt, nop) %.3f"
                  % tuple(g(c) / wc for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC")))
    if g("SQ_WAVES"):
        w = g("SQ_WAVES")
        print("# per wave: VALU %.0f  MFMA %.0f  LDS %.0f  VMEM rd %.0f wr %.0f  SALU %.0f instructions; MFMA busy cycles %.0f; LDS bank-conflict cycles %.0f; wave quad-cycles %.0f"
              % (g("SQ_INSTS_VALU") / w, g("SQ_INSTS_MFMA") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w,
                 g("SQ_INSTS_SALU") / w, g("SQ_VALU_MFMA_BUSY_CYCLES") / w, g("SQ_LDS_BANK_CONFLICT") / w, wc / w))
    if g("GRBM_GUI_ACTIVE") and dur:
        # (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs)
        print("# effective clock: GRBM_GUI_ACTIVE / 8 XCDs / duration = %.2f GHz" % (g("GRBM_GUI_ACTIVE") / 8.0 / (dur * 1e3)))


if __name__ == "__main__":
    main()
