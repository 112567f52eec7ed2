#!/usr/bin/env python
"""A 64-query batch under one bitmap / under a bitmap per caller on the 8-bit mirror vs the bf16 mirror (NMN_NO_I8_MASKED_MFMA=1
in the environment selects the latter): 10M x 768 cosine TOP-100, host API, wall time per batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from neumann_amd import GpuFlatIndex  # noqa: E402

n, d, k = 10_000_000, 768, 100
with GpuFlatIndex(d, n) as idx:
    idx.fill_synthetic(11, n)
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((64, d)).astype(np.float32)
    keep = rng.random(n) < 0.5
    words = np.packbits(keep, bitorder="little").view(np.uint64) if n % 64 == 0 else None
    mask = words
    idx.search(Q, k, 0)
    for name, m in (("no bitmap", None), ("one bitmap, selectivity 0.5", mask)):
        r = idx.search(Q, k, 0, mask=m, with_stats=True)
        t0 = time.perf_counter()
        for _ in range(5):
            r = idx.search(Q, k, 0, mask=m, with_stats=True)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        st = r[3]
        print(f"{'bf16 (NMN_NO_I8_MASKED_MFMA)' if os.environ.get('NMN_NO_I8_MASKED_MFMA') else '8-bit'}: {name}: {ms:.3f} ms per 64-query batch, "
              f"{64e3 / ms:.0f} q/s, bytes per element {st.bytes_scanned // max(1, st.rows_scanned * d)}, candidates {st.candidates_rescored}, fallbacks {st.fallback_queries}")
