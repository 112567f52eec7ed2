#!/usr/bin/env python3
"""Upload cost of a shard (the mirror rebuild / f1 batched uploads): device-to-device upload of f32 rows into an index,
timed with events on the upload stream.  An upload is a copy into the shard (read 4 B + write 4 B per element) and then
the derivation of what the sweeps need — magnitudes in the reference's summation order, 1/|v|, the bf16 mirror and its
rounding-error bound — which round 1 did in three kernels (three reads of the rows) and the ingest kernel does in one.

    python tools/ingest_bench.py [--rows 10000000] [--dim 768] [--chunk 2500000]
    NMN_NO_INGEST=1 python tools/ingest_bench.py      # round 1's three kernels, for the A/B
Kernel-level: rocprofv3 --kernel-trace --stats -- python tools/ingest_bench.py  (ingest_kernel row; its read rate is
rows * dim * 4 / duration)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neumann_amd import GpuFlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--chunk", type=int, default=2_500_000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    src = torch.randn(a.chunk, a.dim, device=dev, dtype=torch.float32)
    idx = GpuFlatIndex(a.dim, a.rows, device=0)
    stream = torch.cuda.current_stream()
    best = None
    for rep in range(a.reps + 1):  # first pass allocates the mirror
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for row0 in range(0, a.rows, a.chunk):
            n = min(a.chunk, a.rows - row0)
            idx.upload_device(src[:n], row0, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if rep and (best is None or ms < best):
            best = ms
    elems = a.rows * a.dim
    # a search right after: the mirror the upload built must serve it (certified elsewhere; here just that it runs)
    q = torch.randn(a.dim, device=dev)
    idx.search(q.cpu().numpy(), 10)
    print(json.dumps({
        "workload": f"upload_device of {a.rows} x {a.dim} f32 in chunks of {a.chunk}",
        "path": "three kernels (NMN_NO_INGEST)" if os.environ.get("NMN_NO_INGEST") else "one-pass ingest kernel",
        "ms_total": round(best, 3),
        "rows_per_s": round(a.rows / (best / 1e3)),
        "GBps_of_f32_rows_uploaded": round(elems * 4 / (best / 1e3) / 1e9, 1),
        "note": "wall time of copy + derivation on the stream; per-kernel durations in the rocprofv3 trace",
    }))


if __name__ == "__main__":
    main()
