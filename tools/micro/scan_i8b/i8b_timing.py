#!/usr/bin/env python
"""Per-wave cycle counts of scan_i8b_kernel (variant build -DNMN_I8B_TIMING): k-loop / epilogue / total per tile, and the
timeline of the workgroups (entry, loop start, end) of the last main sweep."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from neumann_amd import GpuFlatIndex, synth_rows
rows, dim, nq = 10_000_000, 768, 64
W = int(os.environ.get("I8B_WG_WAVES", "4"))
with GpuFlatIndex(dim, rows) as idx:
    idx.fill_synthetic(3, rows)
    Q = synth_rows(4, 0, nq * 2, dim)
    idx.set_timing(True)
    ms = []
    for i in range(8):
        _, _, _, st = idx.search(Q[(i % 2) * nq:(i % 2 + 1) * nq], 100, 0, with_stats=True)
        ms.append(st.scan_ms)
    print("scan_ms (sampling pass + bound kernels + main sweep) of the last calls:", [round(x, 3) for x in ms[-4:]])
    lib = ctypes.CDLL(os.environ["NEUMANN_GPU_LIB"])
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    rc = lib.nmn_i8b_debug_read(buf.ctypes.data_as(ctypes.c_void_p))
    d = buf.reshape(4096, 8).astype(np.float64)
    d = d[d[:, 3] > 0]
    tiles = d[:, 3]
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{tag} rc={rc} waves {len(d)} tiles/wave {tiles.mean():.1f}: per tile cycles (s_memtime ticks): k-loop {np.mean(d[:,0]/tiles):.0f}  epilogue {np.mean(d[:,1]/tiles):.0f}  "
          f"all {np.mean(d[:,2]/tiles):.0f};  per wave total {d[:,2].mean():.0f} (min {d[:,2].min():.0f} max {d[:,2].max():.0f})")
    t0 = d[:, 4].min()
    ent, beg, end = d[:, 4] - t0, d[:, 5] - t0, d[:, 6] - t0
    print(f"{tag} kernel span (first entry -> last end) {end.max():.0f} ticks; entry->loop start mean {np.mean(beg-ent):.0f} max {np.max(beg-ent):.0f}")
    nwg = len(d) // W
    wg_ent = ent[:nwg * W].reshape(nwg, W).min(axis=1); wg_end = end[:nwg * W].reshape(nwg, W).max(axis=1)
    wg_dur = wg_end - wg_ent
    wave_dur = (end - ent)[:nwg * W].reshape(nwg, W)
    print(f"{tag} workgroups {nwg}: duration mean {wg_dur.mean():.0f} min {wg_dur.min():.0f} max {wg_dur.max():.0f}; waves of a workgroup: mean duration by slot {np.round(wave_dur.mean(axis=0)).astype(int).tolist()}")
    print(f"{tag} workgroup entry times: quartiles {np.percentile(wg_ent, [0, 25, 50, 75, 100]).astype(int).tolist()}; end times: {np.percentile(wg_end, [0, 25, 50, 75, 100]).astype(int).tolist()}")
    first = wg_ent < np.percentile(wg_ent, 45)
    print(f"{tag} first-round workgroups ({first.sum()}): duration mean {wg_dur[first].mean():.0f}; later ({(~first).sum()}): {wg_dur[~first].mean():.0f}; slowest wave / mean wave per workgroup: {np.mean(wave_dur.max(axis=1) / wave_dur.mean(axis=1)):.3f}")
