#!/usr/bin/env python
"""What tools/mask_pmc.sh profiles: 10M x 1536 Euclidean TOP-1000 under a bitmap of the selectivity in argv[1], 6 searches."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd import GpuFlatIndex  # noqa: E402

sel = float(sys.argv[1])
n, d, k = 10_000_000, 1536, 1000
dev = torch.device("cuda", 0)
with GpuFlatIndex(d, n, device=0) as idx:
    idx.fill_synthetic(0x5EED0005, n)
    keep = np.random.default_rng(5).random(n) < sel
    words = np.packbits(keep, bitorder="little")
    words = np.pad(words, (0, (-len(words)) % 8)).view(np.uint64)
    mask_t = torch.from_numpy(words.view(np.int64)).to(dev)
    q = torch.randn(4, d, device=dev)
    for i in range(6):
        idx.search_device(q[i % 4:i % 4 + 1], k, 1, mask_t=mask_t)
    torch.cuda.synchronize()
    print(json.dumps({"sel": sel, "kept_rows": int(keep.sum()), "dim": d, "searches": 6}))
