"""ctypes front end of oracle/nmn_oracle.c (test infrastructure; see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
COSINE, EUCLIDEAN, DOT, SPARSE_COS64 = 0, 1, 2, 3

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile both oracle builds with the Makefile next to this file."""
    if force or not all(os.path.exists(os.path.join(_DIR, n))
                        for n in ("libnmn_oracle.so", "libnmn_oracle_native.so")):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))


def _bind(lib):
    lib.orc_dot8.restype = C.c_float
    lib.orc_dot8.argtypes = [_f32p, _f32p, C.c_uint64]
    lib.orc_sumsq8.restype = C.c_float
    lib.orc_sumsq8.argtypes = [_f32p, C.c_uint64]
    lib.orc_magnitude.restype = C.c_float
    lib.orc_magnitude.argtypes = [_f32p, C.c_uint64]
    lib.orc_euclidean_seq.restype = C.c_float
    lib.orc_euclidean_seq.argtypes = [_f32p, _f32p, C.c_uint64]
    lib.orc_cosine.restype = C.c_float
    lib.orc_cosine.argtypes = [_f32p, _f32p, C.c_uint64, C.c_float]
    lib.orc_score.restype = C.c_float
    lib.orc_score.argtypes = [_f32p, _f32p, C.c_uint64, C.c_float, C.c_int]
    lib.orc_sparse_cos64.restype = C.c_float
    lib.orc_sparse_cos64.argtypes = [_f32p, _f32p, C.c_uint64]
    lib.orc_compute_similarity.restype = C.c_float
    lib.orc_compute_similarity.argtypes = [_f32p, _f32p, C.c_uint64]
    lib.orc_scores_all.restype = None
    lib.orc_scores_all.argtypes = [_f32p, C.c_uint64, C.c_uint32, _f32p, C.c_int, _u64p, _f32p, C.c_int]
    for name in ("orc_search", "orc_search_partial"):
        fn = getattr(lib, name)
        fn.restype = C.c_int64
        fn.argtypes = [_f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, C.c_int, _u64p, C.c_uint64,
                       _u64p, _f32p, C.c_int]
    lib.orc_merge_topk.restype = None
    lib.orc_merge_topk.argtypes = [_u64p, _f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _f32p, _u32p]
    lib.orc_synth_value.restype = C.c_float
    lib.orc_synth_value.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    lib.orc_synth_fill.restype = None
    lib.orc_synth_fill.argtypes = [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    lib.orc_synth_fill_mt.restype = None
    lib.orc_synth_fill_mt.argtypes = [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
    return lib


_libs = {}


def lib(native=False):
    key = "native" if native else "plain"
    if key not in _libs:
        build()
        name = "libnmn_oracle_native.so" if native else "libnmn_oracle.so"
        _libs[key] = _bind(C.CDLL(os.path.join(_DIR, name)))
    return _libs[key]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


def dot8(a, b):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_dot8(_p(a, _f32p), _p(b, _f32p), a.size))


def sumsq8(v):
    v = _f32(v)
    return np.float32(lib().orc_sumsq8(_p(v, _f32p), v.size))


def magnitude(v):
    v = _f32(v)
    return np.float32(lib().orc_magnitude(_p(v, _f32p), v.size))


def euclidean_seq(a, b):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_euclidean_seq(_p(a, _f32p), _p(b, _f32p), a.size))


def sparse_cos64(a, b):
    """tensor_blob artifact similarity: SparseVector::cosine_similarity of the two from_dense'd vectors."""
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_sparse_cos64(_p(a, _f32p), _p(b, _f32p), a.size))


def compute_similarity(a, b):
    a, b = _f32(a), _f32(b)
    return np.float32(lib().orc_compute_similarity(_p(a, _f32p), _p(b, _f32p), a.size))


def score(q, v, metric):
    q, v = _f32(q), _f32(v)
    return np.float32(lib().orc_score(_p(q, _f32p), _p(v, _f32p), q.size, magnitude(q), metric))


def scores_all(corpus, q, metric, mask=None, nthreads=1, native=False):
    corpus, q = _f32(corpus), _f32(q)
    n, d = corpus.shape
    out = np.empty(n, dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint64)
    lib(native).orc_scores_all(_p(corpus, _f32p), n, d, _p(q, _f32p), metric,
                               None if m is None else _p(m, _u64p), _p(out, _f32p), nthreads)
    return out


def search(corpus, q, k, metric=COSINE, mask=None, row_base=0, nthreads=1, partial=False, native=False):
    """search_similar_with_metric over a flat corpus -> (rows u64[cnt], scores f32[cnt])."""
    corpus, q = _f32(corpus), _f32(q)
    n, d = corpus.shape if corpus.ndim == 2 else (0, q.size)
    rows = np.empty(max(k, 1), dtype=np.uint64)
    sc = np.empty(max(k, 1), dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint64)
    fn = lib(native).orc_search_partial if partial else lib(native).orc_search
    cnt = fn(_p(corpus, _f32p), n, d, _p(q, _f32p), k, metric, None if m is None else _p(m, _u64p),
             row_base, _p(rows, _u64p), _p(sc, _f32p), nthreads)
    if cnt < 0:
        raise ValueError({-3: "EmptyVector", -4: "InvalidTopK"}.get(cnt, f"oracle error {cnt}"))
    return rows[:cnt].copy(), sc[:cnt].copy()


def merge_topk(rows, scores, counts, k):
    """rows/scores [lists][nq][k], counts [lists][nq] -> merged ([nq][k], [nq][k], [nq])."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    scores = _f32(scores)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nl, nq = counts.shape
    o_r = np.empty((nq, k), dtype=np.uint64)
    o_s = np.empty((nq, k), dtype=np.float32)
    o_c = np.empty(nq, dtype=np.uint32)
    lib().orc_merge_topk(_p(rows, _u64p), _p(scores, _f32p), _p(counts, _u32p), nl, nq, k,
                         _p(o_r, _u64p), _p(o_s, _f32p), _p(o_c, _u32p))
    return o_r, o_s, o_c


def synth(seed, row0, n, dim, nthreads=1):
    out = np.empty((n, dim), dtype=np.float32)
    if nthreads > 1:
        lib(native=True).orc_synth_fill_mt(_p(out, _f32p), seed, row0, n, dim, nthreads)
    else:
        lib().orc_synth_fill(_p(out, _f32p), seed, row0, n, dim)
    return out


def mask_from_bool(keep):
    """bool[n] -> LSB-first u64 words (relational_engine bitmap layout)."""
    keep = np.asarray(keep, dtype=bool)
    n = keep.size
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=bool)
    padded[:n] = keep
    return np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
