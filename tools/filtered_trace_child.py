#!/usr/bin/env python
"""What tools/r04_gpu_z5.sh traces: filtered SIMILAR at selectivity 0.1 through nmn_index_search_pred (10M x 768, WHERE bucket = 3 of 10), 12 searches."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex, synth_rows  # noqa: E402
from neumann_amd import columns as g  # noqa: E402

n, d, k = 10_000_000, 768, 100
with GpuFlatIndex(d, n, device=0) as idx, g.GpuColumns(n) as cols:
    idx.fill_synthetic(3, n)
    col = cols.add_column()
    bucket = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(10)
    cols.write(col, 0, np.full(n, g.CELL_INT, np.uint8), bucket)
    cols.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    prog = [(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, col, 3, 0)]
    Q = synth_rows(7, 0, 8, d)
    for i in range(12):
        idx.search_pred(cols, prog, [], Q[i % 8], k, 0)
