#!/bin/bash
# round 4: where select_kernel's time goes (variant build with -DNMN_SELECT_TRACE: 10-ns ticks per phase, query 0)
OUT=gpurun_out/r04f; mkdir -p $OUT
export NEUMANN_GPU_LIB=$PWD/neumann_amd/lib/variants/libneumann_gpu_seltrace.so
python - > $OUT/select_phases.txt 2>&1 <<'P'
import numpy as np, sys
from neumann_amd import GpuFlatIndex, synth_rows
def run(rows, d, k, metric, sel, tag):
    idx = GpuFlatIndex(d, rows, device=0); idx.fill_synthetic(3, rows)
    Q = synth_rows(5, 0, 6, d)
    mask = None
    if sel < 1.0:
        keep = np.random.default_rng(1).random(rows) < sel
        words = (rows + 63) // 64
        pad = np.zeros(words * 64, bool); pad[:rows] = keep
        mask = np.packbits(pad.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
    print("==", tag, flush=True)
    for i in range(6):
        idx.search(Q[i], k, metric, mask=mask)
    idx.close()
run(1_000_000, 768, 100, 0, 1.0, "1M x 768 cosine TOP-100")
run(10_000_000, 768, 100, 0, 1.0, "10M x 768 cosine TOP-100")
run(10_000_000, 1536, 1000, 1, 1.0, "10M x 1536 L2 TOP-1000")
run(10_000_000, 1536, 1000, 1, 0.1, "10M x 1536 L2 TOP-1000 mask 0.1")
run(10_000_000, 1536, 1000, 1, 0.01, "10M x 1536 L2 TOP-1000 mask 0.01")
run(2_000_000, 768, 100, 0, 0.004, "2M x 768 cosine TOP-100 mask 0.004 (an IVF probe's share)")
P
cat $OUT/select_phases.txt
