#!/bin/bash
# select_kernel's phases (variant -DNMN_SELECT_TRACE; 100 MHz ticks = 10 ns) on 1M x 768 (768 waves), 10M x 768 (4096 waves), and an IVF-like masked probe
OUT=$PWD/gpurun_out/r04y; mkdir -p $OUT; R=$PWD
NEUMANN_GPU_LIB=$R/neumann_amd/lib/variants/libneumann_gpu_seltrace.so python - > $OUT/select_phases.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from neumann_amd import GpuFlatIndex, synth_rows
for rows in (1_000_000, 10_000_000):
    with GpuFlatIndex(768, rows) as idx:
        idx.fill_synthetic(3, rows)
        Q = synth_rows(4, 0, 8, 768)
        print(f"== {rows} x 768 k=100", flush=True)
        for i in range(6):
            idx.search(Q[i], 100, 0)
        keep = np.zeros(rows, dtype=bool)
        for s in range(8):
            a = (s * 2 + 1) * rows // 17
            keep[a:a + rows // 70] = True
        words = np.packbits(keep, bitorder="little"); words = np.pad(words, (0, (-len(words)) % 8)).view(np.uint64)
        print(f"== {rows} x 768 k=100 under a bitmap of 8 runs ({keep.sum()} rows)", flush=True)
        for i in range(4):
            idx.search(Q[i], 100, 1, mask=words)
PY
grep -v amdgpu $OUT/select_phases.txt | tail -30
