"""Host-buffer API latency (PCIe-inclusive: query H2D + results D2H + 2 stream syncs) vs device API."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from neumann_amd import GpuFlatIndex, synth_rows
for rows in (1_000_000, 10_000_000):
    with GpuFlatIndex(768, rows) as idx:
        idx.fill_synthetic(3, rows)
        Q = synth_rows(4, 0, 32, 768)
        for i in range(5):
            idx.search(Q[i], 100, 0)
        t0 = time.perf_counter()
        for i in range(32):
            idx.search(Q[i], 100, 0)
        dt = (time.perf_counter() - t0) / 32
        print(f"rows={rows}: host-buffer nmn_index_search latency {dt*1e3:.3f} ms/query ({1/dt:.1f} q/s)")
