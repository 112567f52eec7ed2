#!/usr/bin/env python3
"""nmn_sharded_* on one GPU box: S logical shards of one 10M x 768 corpus on device 0 (peer-copy gather), and one shard
through the RCCL all-gather — what the handle costs on top of the shard-local searches (gather + merge), and that the
merged answer is the unsharded one.

    python tools/sharded_probe.py [--rows 10000000] [--dim 768] [--k 100]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from neumann_amd import GpuFlatIndex, GpuShardedIndex, synth_rows  # noqa: E402
from neumann_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    a = ap.parse_args()
    Q = synth_rows(4, 0, 64, a.dim)
    with GpuFlatIndex(a.dim, a.rows) as flat:
        flat.fill_synthetic(3, a.rows)
        ref = [flat.search(Q[i], a.k, 0) for i in range(4)]
        ref64 = flat.search(Q, a.k, 0)
    for shards, gather, label in ((1, _capi.GATHER_RCCL, "rccl"), (2, _capi.GATHER_PEER, "peer"), (4, _capi.GATHER_PEER, "peer"),
                                  (8, _capi.GATHER_PEER, "peer")):
        with GpuShardedIndex(a.dim, a.rows, shards, devices=[0] * shards, gather=gather) as s:
            s.fill_synthetic(3, a.rows)
            s.set_timing(True)
            for i in range(4):
                r, sc, c = s.search(Q[i], a.k, 0)
                assert np.array_equal(r, ref[i][0]) and np.array_equal(sc.view(np.uint32), ref[i][1].view(np.uint32))
            r, sc, c = s.search(Q, a.k, 0)
            assert np.array_equal(r, ref64[0]) and np.array_equal(sc.view(np.uint32), ref64[1].view(np.uint32))
            lat, g = [], []
            for i in range(32):
                t0 = time.perf_counter()
                s.search(Q[i], a.k, 0)
                lat.append((time.perf_counter() - t0) * 1e3)
                g.append(s.last_gather_ms())
            t0 = time.perf_counter()
            for _ in range(5):
                s.search(Q, a.k, 0)
            b64 = (time.perf_counter() - t0) / 5 * 1e3
            # concurrent single-query callers of the handle (Python threads: ctypes releases the GIL for the call)
            import threading
            conc = {}
            for T in (1, 16, 64):
                stop = time.perf_counter() + 1.5
                done = [0] * T

                def work(t):
                    i = t
                    while time.perf_counter() < stop:
                        s.search(Q[i % 64], a.k, 0)
                        i += 1
                        done[t] += 1
                b0, c0 = s.coalesce_stats()
                t0 = time.perf_counter()
                th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                el = time.perf_counter() - t0
                b1, c1 = s.coalesce_stats()
                conc[f"threads_{T}"] = {"qps": round(sum(done) / el), "merged_batches": b1 - b0, "calls_in_them": c1 - c0}
            print(json.dumps({"shards_on_device_0": shards, "gather": label, "concurrent_callers": conc, "rows": a.rows, "dim": a.dim, "k": a.k,
                              "nq1_ms_median": round(float(np.median(lat)), 3), "gather_plus_merge_ms_median": round(float(np.median(g)), 4),
                              "nq64_ms": round(b64, 3), "identical_to_unsharded": True}))


if __name__ == "__main__":
    main()
