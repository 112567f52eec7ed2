#!/usr/bin/env python
"""Where a step of the single-query loop goes when the shard is small enough for the host to matter (1M x 768): time spent in the
Python call, in the C entry point (ctypes), and the wall time per step with 1 / 2 streams, outputs preallocated or not."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from neumann_amd import GpuFlatIndex, _capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 768, 100
dev = torch.device("cuda:0")
with GpuFlatIndex(d, rows) as idx:
    idx.fill_synthetic(7, rows)
    q = torch.randn(4, d, device=dev)
    outs = [(torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev),
             torch.empty((1,), dtype=torch.int32, device=dev)) for _ in range(4)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    for ns in (1, 2, 4):
        for pre in (False, True):
            for _ in range(8):
                for i in range(ns):
                    with torch.cuda.stream(streams[i]):
                        idx.search_device(q[i:i + 1], k, 0)
            torch.cuda.synchronize()
            n = 400
            host = 0.0
            t0 = time.perf_counter()
            for i in range(n):
                s = streams[i % ns]
                with torch.cuda.stream(s):
                    a = time.perf_counter()
                    idx.search_device(q[i % 4:i % 4 + 1], k, 0, out=outs[i % 4] if pre else None)
                    host += time.perf_counter() - a
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"rows {rows} streams {ns} prealloc {pre}: wall/step {t_all / n * 1e6:7.1f} us  issue loop/step {t_issue / n * 1e6:7.1f} us  inside search_device {host / n * 1e6:7.1f} us")
    # the C entry point alone (no torch allocations, no Python wrapper): raw ctypes call
    lib = idx._lib
    qp = C.c_void_p(q.data_ptr())
    r, s_, c = outs[0]
    sp = C.c_void_p(streams[0].cuda_stream)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        lib.nmn_index_search_device(idx._h, qp, 1, k, 0, None, C.c_void_p(r.data_ptr()), C.c_void_p(s_.data_ptr()), C.c_void_p(c.data_ptr()), sp)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"raw ctypes, 1 stream: wall/step {t_all / n * 1e6:7.1f} us  issue/step {t_issue / n * 1e6:7.1f} us")
