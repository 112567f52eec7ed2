"""Build libneumann_gpu.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m neumann_amd.build [--force]

Outputs neumann_amd/lib/libneumann_gpu.so in-tree (git-ignored; travels to the GPU box with the
snapshot).  nmn_exact.hip is compiled with -ffp-contract=off: it restates the reference's unfused
f32 arithmetic (tensor_store/src/hnsw.rs:168-229) and must never be contracted into FMAs.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libneumann_gpu.so")
ARCH = "gfx950"

SOURCES = {
    "nmn_scan.hip": [],
    "nmn_scan_ring.hip": [],
    "nmn_scan_mfma.hip": [],
    "nmn_scan_mfma_f32.hip": [],
    "nmn_scan_mfma_i8x.hip": [],
    "nmn_scan_i8.hip": [],
    "nmn_select.hip": [],
    "nmn_exact.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "nmn_synth.hip": ["-ffp-contract=off"],
    "nmn_ingest.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "nmn_sortk.hip": [],
    "nmn_columns.hip": [],
    "nmn_ivf.hip": ["-ffp-contract=off"],
    "nmn_kmeans.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "nmn_api.hip": [],
    "nmn_sharded.hip": [],
    "nmn_persist.hip": [],
    "nmn_engine.cpp": ["-ffp-contract=off"],
}
HEADERS = ["nmn_internal.h", "nmn_index.h", "nmn_scan_mfma_kernel.h", "nmn_persist.h", "nmn_select_dev.h", os.path.join("..", "..", "include", "neumann_gpu.h"),
           os.path.join("..", "..", "include", "neumann_engine.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _newer([sp, os.path.abspath(__file__)] + hdrs, obj):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall",
                   "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")] + extra + ["-c", sp, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn.strip():
                    print(warn)
    if force or jobs or _newer(objs, LIB):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
