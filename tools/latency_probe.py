"""Host-buffer API latency (PCIe-inclusive: query H2D + results D2H + stream sync) at several shard sizes.

  python tools/latency_probe.py [rows:dim:k ...]      default: 10000:128:5 100000:768:100 1000000:768:100 10000000:768:100
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402

cases = sys.argv[1:] or ["10000:128:5", "100000:768:100", "1000000:768:100", "10000000:768:100"]
for c in cases:
    rows, dim, k = (int(x) for x in c.split(":"))
    with GpuFlatIndex(dim, rows) as idx:
        idx.fill_synthetic(3, rows)
        Q = synth_rows(4, 0, 64, dim)
        for i in range(8):
            idx.search(Q[i], k, 0)
        lat = []
        for i in range(64):
            t0 = time.perf_counter()
            idx.search(Q[i], k, 0)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        print(f"rows={rows} dim={dim} k={k}: nmn_index_search median {np.median(lat):.3f} ms  p10 {np.percentile(lat, 10):.3f}  "
              f"p90 {np.percentile(lat, 90):.3f}  ({1e3 / np.median(lat):.0f} q/s single caller)")
