#!/usr/bin/env python
"""What tools/ivf_pmc.sh profiles: an IVF-Flat index (2M x 768, 256 lists), 24 probes at nprobe 8.  Prints the mean number of
rows in the probed lists (from the trained centroids and the list sizes, computed on the host) so that the PMC bytes of the
list-scan kernel can be set against the bytes of the listed rows."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neumann_amd.flat_index import synth_rows  # noqa: E402
from neumann_amd.ivf import GpuIvfFlat  # noqa: E402

n, d, C, nprobe, k = 2_000_000, 768, 256, 8, 100
ivf = GpuIvfFlat.build(synth_rows(0x1F6, 0, 200_000, d), C, nprobe=nprobe, max_iterations=3, seed=42, init_method="kmeans++", capacity_rows=n)
with ivf:
    for r0 in range(200_000, n, 300_000):
        ivf.add(synth_rows(0x1F6, r0, min(300_000, n - r0), d))
    cents, sizes = ivf.centroids(), ivf.cluster_sizes()
    Q = synth_rows(0x1F7, 0, 24, d)
    listed = []
    for q in Q:
        d2 = ((cents - q[None, :]) ** 2).sum(axis=1)
        listed.append(int(sizes[np.argsort(d2, kind="stable")[:nprobe]].sum()))
    ivf.search(Q[0], k)      # builds the mirrors of whichever copy the probes read
    for q in Q:
        ivf.search(q, k)
    print(json.dumps({"rows": n, "dim": d, "lists": C, "nprobe": nprobe, "probes": len(Q) + 1, "list_major_rows": ivf.list_major_rows,
                      "mean_listed_rows": float(np.mean(listed))}))
