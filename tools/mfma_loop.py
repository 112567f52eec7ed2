#!/usr/bin/env python3
"""Clean A/B timing of the batched sweep: one stream, nothing else on the device, N searches of NQ queries back to back;
reports the event-timed scan portion (sampling pass + main sweep) of every call.  Variants: NEUMANN_GPU_LIB=<variant .so>,
knobs: NMN_MFMA_WGS etc.  Several (rows, dim) pairs may follow.

    python tools/mfma_loop.py [--nq 64] [--reps 40] [--metric 0] [rows:dim ...]      default 10000000:768
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=64)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--tag", default="")
    ap.add_argument("--mirror", type=int, default=1, help="nmn_index_set_mirror before the rows arrive: 1 default, 2 bf16, 0 the f32 rows")
    ap.add_argument("--realloc", type=int, default=1, help="build the index this many times in the same process (placement A/B)")
    ap.add_argument("shapes", nargs="*", default=["10000000:768"])
    a = ap.parse_args()
    for sh in a.shapes:
        rows, dim = (int(x) for x in sh.split(":"))
        for _round in range(a.realloc):
          with GpuFlatIndex(dim, rows) as idx:
              idx.set_mirror(a.mirror)
              idx.fill_synthetic(3, rows)
              idx.set_timing(True)
              Q = synth_rows(4, 0, a.nq * 4, dim)
              for i in range(4):
                  idx.search(Q[i * a.nq:(i + 1) * a.nq], a.k, a.metric)
              t = []
              for i in range(a.reps):
                  j = i % 4
                  _, _, _, st = idx.search(Q[j * a.nq:(j + 1) * a.nq], a.k, a.metric, with_stats=True)
                  t.append(st.scan_ms)
              t = np.sort(np.array(t))
              eb = st.bytes_scanned // max(1, st.rows_scanned * dim)  # bytes per element the sweep read (1 / 2 / 4)
              gb = rows * dim * eb / 1e6  # GB per ms -> TB/s below
              print(f"{a.tag or os.environ.get('NEUMANN_GPU_LIB', 'default').split('_')[-1]:>10} wgs={os.environ.get('NMN_MFMA_WGS', '-'):>5} "
                    f"{rows}x{dim} nq={a.nq} {eb} B/elem: scan_ms min {t[0]:.3f} p25 {t[len(t) // 4]:.3f} med {np.median(t):.3f} p90 {t[int(len(t) * 0.9)]:.3f}"
                    f"  | med -> {gb / np.median(t):.0f} GB/s = {gb / np.median(t) / 8000:.3f} of 8 TB/s (sampling pass included)")


if __name__ == "__main__":
    main()
