"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/neumann_gpu.h declares,
its host-only entry points work, and without a GPU the compute entry points fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from neumann_amd import _capi
from oracle import oracle_c as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nmn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    names = _declared("neumann_gpu.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/neumann_gpu.h but not exported"
    assert set(names) == set(_capi.SIGNATURES), set(names) ^ set(_capi.SIGNATURES)


def test_rust_ffi_file_is_current_and_complete():
    """integration/rust/ffi.rs (the `mod ffi` of INTEGRATION.md §2) is generated from the header: it must be what the
    generator emits today and name every exported function exactly once; the safe wrappers only call what it declares."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ffi = open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read()
    fns = re.findall(r"pub fn (nmn_[a-z0-9_]+)\(", ffi)
    assert sorted(fns) == _declared("neumann_gpu.h")
    assert len(fns) == len(set(fns))
    wrappers = open(os.path.join(ROOT, "integration", "rust", "gpu_index.rs")).read()
    used = set(re.findall(r"ffi::(nmn_[a-z0-9_]+)\s*\(", wrappers))
    assert used and used <= set(fns), used - set(fns)
    for const in re.findall(r"ffi::(NMN_[A-Z0-9_]+)", wrappers):
        assert re.search(rf"pub const {const}:", ffi), const
    # struct layouts: same field order as the C structs ctypes uses
    for cname, ctype in (("nmn_index_desc", _capi.IndexDesc), ("nmn_sharded_desc", _capi.ShardedDesc),
                         ("nmn_search_stats", _capi.SearchStats)):
        body = re.search(rf"pub struct {cname} \{{(.*?)\}}", ffi, flags=re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == [f[0] for f in ctype._fields_], cname


def test_engine_header_symbols_exported():
    path = os.path.join(ROOT, "include", "neumann_engine.h")
    if not os.path.exists(path):
        pytest.skip("engine header not present")
    lib = _capi.load()
    for n in _declared("neumann_engine.h"):
        assert hasattr(lib, n), n


def test_status_strings_mirror_vector_error_display():  # vector_engine/src/lib.rs:151-183
    lib = _capi.load()
    assert lib.nmn_status_str(_capi.ERR_EMPTY_VECTOR) == b"Empty vector provided"
    assert lib.nmn_status_str(_capi.ERR_INVALID_TOP_K) == b"Invalid top_k value (must be > 0)"
    assert lib.nmn_status_str(_capi.OK) == b"ok"
    assert b"no CPU fallback" in lib.nmn_status_str(_capi.ERR_NO_DEVICE)


def test_host_merge_matches_oracle():
    from neumann_amd import merge_topk_host
    rng = np.random.default_rng(1)
    L, nq, k = 5, 3, 16
    rows = np.full((L, nq, k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    scores = np.full((L, nq, k), -np.inf, dtype=np.float32)
    counts = rng.integers(0, k + 1, size=(L, nq)).astype(np.uint32)
    nxt = 0
    for l in range(L):
        for q in range(nq):
            c = counts[l, q]
            s = np.sort(rng.choice([0.5, 0.25, 0.75, 1.0, -1.0, 0.0], size=c).astype(np.float32))[::-1]
            r = np.arange(nxt, nxt + c, dtype=np.uint64)   # unique rows, ascending within equal scores
            nxt += c
            scores[l, q, :c], rows[l, q, :c] = s, r
    got = merge_topk_host(rows, scores, counts, k)
    exp = oc.merge_topk(rows, scores, counts, k)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


def test_host_synth_matches_oracle_twin():
    from neumann_amd import synth_rows
    a = synth_rows(0x5EED0003, 12345, 50, 33)
    assert np.array_equal(a, oc.synth(0x5EED0003, 12345, 50, 33))
    assert abs(float(a.mean())) < 0.2 and 0.8 < float(a.std()) < 1.2


def test_no_device_fails_loudly(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    lib = _capi.load()
    h = C.c_void_p()
    desc = _capi.IndexDesc(dim=8, flags=0, capacity_rows=10, row_base=0, device=-1, cand_cap=0)
    assert lib.nmn_index_create(C.byref(desc), C.byref(h)) == _capi.ERR_NO_DEVICE
    from neumann_amd import GpuFlatIndex, NeumannGpuError
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex(8, 10)
    assert e.value.status == _capi.ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """neumann_amd/ (Python and csrc) must not import, link or call the oracle: it is test infrastructure."""
    pkg = os.path.join(ROOT, "neumann_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle/|nmn_oracle|orc_[a-z]", re.M)
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert not pat.search(txt), f"{os.path.join(base, f)} references the oracle"
