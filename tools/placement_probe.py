#!/usr/bin/env python3
"""Which allocation's placement moves the matrix-core sweep's time?  One index (mirror fixed), several HIP streams: every
stream gets a workspace of its own (scores, tile maxima, ...), so a spread ACROSS streams is the workspace's placement, a
spread across index builds (tools/mfma_loop.py --realloc) the mirror's.

    python tools/placement_probe.py [--nq 128] [--streams 4] [--builds 2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=128)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--builds", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    Q = torch.from_numpy(synth_rows(4, 0, a.nq, a.dim)).to(dev)
    for b in range(a.builds):
        with GpuFlatIndex(a.dim, a.rows) as idx:
            idx.fill_synthetic(3, a.rows)
            idx.set_timing(True)
            ld = idx.row_stride if hasattr(idx, "row_stride") else a.dim
            for si in range(a.streams):
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    for _ in range(3):
                        idx.search_device(Q, 100, 0, stream=st)
                    st.synchronize()
                    t = []
                    for _ in range(24):
                        idx.search_device(Q, 100, 0, stream=st)
                        st.synchronize()
                        t.append(idx.last_stats(st).scan_ms)
                t = np.sort(np.array(t))
                print(f"build {b} stream {si}: scan_ms min {t[0]:.3f} med {np.median(t):.3f} p90 {t[int(len(t) * 0.9)]:.3f}")


if __name__ == "__main__":
    main()
