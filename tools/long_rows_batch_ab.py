#!/usr/bin/env python
"""64-query batches over long rows: the 8-bit matrix-core sweep (default) against the bf16 one (NMN_NO_I8_MFMA=1): wall per batch"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from neumann_amd import GpuFlatIndex  # noqa: E402

for n, d in ((3_000_000, 2048), (2_000_000, 3072)):
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(13, n)
        Q = np.random.default_rng(2).standard_normal((64, d)).astype(np.float32)
        for metric in (0, 1):
            idx.search(Q, 100, metric)
            t0 = time.perf_counter()
            for _ in range(5):
                r = idx.search(Q, 100, metric, with_stats=True)
            ms = (time.perf_counter() - t0) / 5 * 1e3
            st = r[3]
            print(f"{'bf16' if os.environ.get('NMN_NO_I8_MFMA') else '8-bit'} {n} x {d} metric {metric}: {ms:.3f} ms per 64-query batch, {64e3 / ms:.0f} q/s, "
                  f"bytes/elem {st.bytes_scanned // (st.rows_scanned * d)}, candidates {st.candidates_rescored}, fallbacks {st.fallback_queries}")
