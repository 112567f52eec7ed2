#!/usr/bin/env python
"""Per-wave cycle counts of scan_i8b_kernel (variant build -DNMN_I8B_TIMING): k-loop / epilogue / total per tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from neumann_amd import GpuFlatIndex, synth_rows
from neumann_amd import _capi
rows, dim, nq = 10_000_000, 768, 64
with GpuFlatIndex(dim, rows) as idx:
    idx.fill_synthetic(3, rows)
    Q = synth_rows(4, 0, nq * 2, dim)
    for i in range(4):
        idx.search(Q[(i % 2) * nq:(i % 2 + 1) * nq], 100, 0)
    lib = ctypes.CDLL(os.environ["NEUMANN_GPU_LIB"])
    buf = np.zeros(4096 * 4, dtype=np.uint64)
    rc = lib.nmn_i8b_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
    d = buf.reshape(4096, 4).astype(np.float64)
    d = d[d[:, 3] > 0]
    tiles = d[:, 3]
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} rc={rc} waves {len(d)} tiles/wave {tiles.mean():.1f}: per tile cycles (s_memtime ticks): k-loop {np.mean(d[:,0]/tiles):.0f}  epilogue {np.mean(d[:,1]/tiles):.0f}  "
          f"all {np.mean(d[:,2]/tiles):.0f};  per wave total {d[:,2].mean():.0f} (min {d[:,2].min():.0f} max {d[:,2].max():.0f})")
