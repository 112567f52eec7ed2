#!/usr/bin/env python
"""What tools/r04_gpu_n.sh profiles: 10M x 768 cosine TOP-100, batches of 64 queries (the matrix-core sweep over the 8-bit mirror), 6 batches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402

rows, dim, nq = 10_000_000, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 64
with GpuFlatIndex(dim, rows) as idx:
    idx.fill_synthetic(3, rows)
    Q = synth_rows(4, 0, nq * 2, dim)
    for i in range(6):
        idx.search(Q[(i % 2) * nq:(i % 2 + 1) * nq], 100, 0)
