"""ctypes binding of libneumann_gpu.so (include/neumann_gpu.h).

This is the same C ABI the Rust `vector_engine::ffi` module binds; Python only
plays the role of the host language here because the image has no Rust toolchain.  Loading fails
loudly when the library has not been built: there is no Python/CPU fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEUMANN_GPU_LIB: an alternative build of the same library (A/B runs of kernel variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("NEUMANN_GPU_LIB") or os.path.join(_HERE, "lib", "libneumann_gpu.so")

OK = 0
ERR_NOT_FOUND = -1
ERR_DIMENSION_MISMATCH = -2
ERR_EMPTY_VECTOR = -3
ERR_INVALID_TOP_K = -4
ERR_STORAGE = -5
ERR_CONFIGURATION = -6
ERR_COLLECTION_EXISTS = -7
ERR_COLLECTION_NOT_FOUND = -8
ERR_SEARCH_TIMEOUT = -9
ERR_IO = -10
ERR_SERIALIZATION = -11
ERR_INVALID_ARGUMENT = -20
ERR_NO_DEVICE = -21
ERR_OUT_OF_MEMORY = -22
ERR_TOP_K_TOO_LARGE = -23
ERR_CAPACITY = -24
ERR_BUFFER_TOO_SMALL = -25

MAX_TOP_K = 4096
MAX_QUERIES = 1024

METRIC_COSINE, METRIC_EUCLIDEAN, METRIC_DOT_PRODUCT = 0, 1, 2

f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class IndexDesc(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("flags", C.c_uint32), ("capacity_rows", C.c_uint64),
                ("row_base", C.c_uint64), ("device", C.c_int32), ("cand_cap", C.c_uint32)]


class ShardedDesc(C.Structure):
    """nmn_sharded_desc: one corpus row-range sharded over several devices of ONE process."""
    _fields_ = [("dim", C.c_uint32), ("flags", C.c_uint32), ("capacity_rows", C.c_uint64), ("row_base", C.c_uint64),
                ("n_shards", C.c_uint32), ("gather", C.c_uint32), ("devices", C.POINTER(C.c_int32)),
                ("cand_cap", C.c_uint32), ("layout", C.c_uint32)]


GATHER_AUTO, GATHER_RCCL, GATHER_PEER = 0, 1, 2


class SearchStats(C.Structure):
    _fields_ = [("rows_scanned", C.c_uint64), ("bytes_scanned", C.c_uint64),
                ("candidates_rescored", C.c_uint32), ("fallback_queries", C.c_uint32),
                ("scan_ms", C.c_float), ("total_ms", C.c_float),
                ("sweep_kind", C.c_uint32), ("sweep_launches", C.c_uint32)]

    @property
    def sweep(self):
        """Name of `sweep_kind` as the library spells it (nmn_sweep_kind_str): ring_f32, valu_f32, valu_bf16, valu_i8, mfma_f32, ..."""
        return load().nmn_sweep_kind_str(self.sweep_kind).decode()


SWEEP_NONE, SWEEP_RING_F32, SWEEP_VALU_F32, SWEEP_VALU_BF16, SWEEP_VALU_I8, SWEEP_MFMA_F32, SWEEP_MFMA_BF16, SWEEP_MFMA_I8, SWEEP_EXACT = range(9)


class KMeansOptions(C.Structure):
    """nmn_kmeans_options (KMeansConfig, tensor_store/src/delta_vector.rs:691-711)."""
    _fields_ = [("max_iterations", C.c_uint64), ("convergence_threshold", C.c_float), ("seed", C.c_uint64),
                ("init_method", C.c_int32)]


class PredOp(C.Structure):
    """nmn_pred_op: one step of a WHERE-predicate program (include/neumann_gpu.h, NMN_PRED_*)."""
    _fields_ = [("op", C.c_uint32), ("cmp", C.c_uint32), ("vkind", C.c_uint32), ("column", C.c_uint32),
                ("a", C.c_uint64), ("b", C.c_uint64)]


CELL_ABSENT, CELL_NULL, CELL_BOOL, CELL_INT, CELL_FLOAT, CELL_STRING = range(6)
PRED_TRUE, PRED_FALSE, PRED_AND, PRED_OR, PRED_EXISTS, PRED_CMP, PRED_IN, PRED_STRSET = range(8)
CMP_EQ, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE = range(6)

# name -> (restype, argtypes); one entry per declaration in include/neumann_gpu.h
SIGNATURES = {
    "nmn_device_count": (C.c_int32, [i32p]),
    "nmn_status_str": (C.c_char_p, [C.c_int32]),
    "nmn_last_error": (C.c_char_p, []),
    "nmn_version": (C.c_char_p, []),
    "nmn_sweep_kind_str": (C.c_char_p, [C.c_uint32]),
    "nmn_index_create": (C.c_int32, [C.POINTER(IndexDesc), C.POINTER(vp)]),
    "nmn_index_destroy": (C.c_int32, [vp]),
    "nmn_index_upload": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64]),
    "nmn_index_upload_device": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, vp]),
    "nmn_index_set_rows": (C.c_int32, [vp, C.c_uint64]),
    "nmn_index_rows": (C.c_uint64, [vp]),
    "nmn_index_dim": (C.c_uint32, [vp]),
    "nmn_index_row_stride": (C.c_uint32, [vp]),
    "nmn_index_row_base": (C.c_uint64, [vp]),
    "nmn_index_corpus_device": (vp, [vp, u32p]),
    "nmn_index_norms_device": (vp, [vp]),
    "nmn_index_search": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, vp, vp, vp, vp,
                                     C.POINTER(SearchStats)]),
    "nmn_index_search_device": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, vp, vp, vp, vp, vp]),
    "nmn_index_last_stats": (C.c_int32, [vp, vp, C.POINTER(SearchStats)]),
    "nmn_index_set_timing": (C.c_int32, [vp, C.c_int32]),
    "nmn_index_set_mirror": (C.c_int32, [vp, C.c_int32]),
    "nmn_index_scan_history": (C.c_int32, [vp, vp, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "nmn_index_hbm_bytes": (C.c_int32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nmn_index_retry_declined": (C.c_int32, [vp]),
    "nmn_index_score_rows": (C.c_int32, [vp, vp, C.c_uint32, C.c_int32, vp, C.c_uint32, vp]),
    "nmn_index_count_exact": (C.c_int32, [vp, vp, C.c_int32, vp, C.c_float, u64p, u64p]),
    "nmn_index_read_probe": (C.c_int32, [vp, C.c_uint32, C.POINTER(C.c_double)]),
    "nmn_index_coalesce_stats": (C.c_int32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nmn_index_callers_probe": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_double, C.POINTER(C.c_double),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nmn_merge_topk_host": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "nmn_merge_topk_device": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp]),
    "nmn_merge_topk_device_strided": (C.c_int32, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  vp, vp, vp, vp]),
    "nmn_synth_value": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint32]),
    "nmn_synth_fill_host": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "nmn_index_fill_synthetic": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64]),
    "nmn_index_set_row": (C.c_int32, [vp, C.c_uint64, vp]),
    "nmn_columns_create": (C.c_int32, [C.c_int32, C.c_uint64, C.POINTER(vp)]),
    "nmn_columns_destroy": (C.c_int32, [vp]),
    "nmn_columns_add": (C.c_int32, [vp, u32p]),
    "nmn_columns_count": (C.c_uint32, [vp]),
    "nmn_columns_write": (C.c_int32, [vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]),
    "nmn_columns_clear_row": (C.c_int32, [vp, C.c_uint64]),
    "nmn_columns_write_valid": (C.c_int32, [vp, C.c_uint64, C.c_uint64, vp]),
    "nmn_columns_eval": (C.c_int32, [vp, C.POINTER(PredOp), C.c_uint32, vp, C.c_uint64, C.c_uint64, u64p]),
    "nmn_columns_eval_acquire": (C.c_int32, [vp, C.POINTER(PredOp), C.c_uint32, vp, C.c_uint64, C.c_uint64, u64p,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    "nmn_columns_eval_release": (C.c_int32, [vp, C.c_uint32]),
    "nmn_index_search_pred": (C.c_int32, [vp, vp, C.POINTER(PredOp), C.c_uint32, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32,
                                          C.c_int32, vp, vp, vp, u64p, C.POINTER(SearchStats)]),
    "nmn_columns_mask_device": (vp, [vp]),
    "nmn_columns_valid_device": (vp, [vp]),
    "nmn_columns_read_mask": (C.c_int32, [vp, vp, C.c_uint64]),
    "nmn_ivf_create": (C.c_int32, [C.POINTER(IndexDesc), vp, C.c_uint32, C.POINTER(vp)]),
    "nmn_ivf_destroy": (C.c_int32, [vp]),
    "nmn_ivf_build": (C.c_int32, [C.POINTER(IndexDesc), vp, C.c_uint64, C.c_uint32, C.POINTER(KMeansOptions), C.POINTER(vp)]),
    "nmn_ivf_centroids": (C.c_int32, [vp, vp, C.c_uint64]),
    "nmn_ivf_add": (C.c_int32, [vp, vp, C.c_uint64, vp]),
    "nmn_ivf_len": (C.c_uint64, [vp]),
    "nmn_ivf_list_major_rows": (C.c_uint64, [vp]),
    "nmn_ivf_clusters": (C.c_uint32, [vp]),
    "nmn_ivf_cluster_sizes": (C.c_int32, [vp, vp]),
    "nmn_ivf_search": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, C.POINTER(SearchStats)]),
    "nmn_ivf_vectors": (vp, [vp]),
    "nmn_index_save": (C.c_int32, [vp, C.c_char_p]),
    "nmn_index_load": (C.c_int32, [C.c_char_p, C.POINTER(IndexDesc), C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_ivf_save": (C.c_int32, [vp, C.c_char_p]),
    "nmn_ivf_load": (C.c_int32, [C.c_char_p, C.POINTER(IndexDesc), C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_sharded_create": (C.c_int32, [C.POINTER(ShardedDesc), C.POINTER(vp)]),
    "nmn_sharded_destroy": (C.c_int32, [vp]),
    "nmn_sharded_upload": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64]),
    "nmn_sharded_fill_synthetic": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64]),
    "nmn_sharded_search": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, vp, vp, vp, vp, C.POINTER(SearchStats)]),
    "nmn_sharded_shards": (C.c_uint32, [vp]),
    "nmn_sharded_rows": (C.c_uint64, [vp]),
    "nmn_sharded_shard": (vp, [vp, C.c_uint32]),
    "nmn_sharded_device": (C.c_int32, [vp, C.c_uint32]),
    "nmn_sharded_gather_mode": (C.c_uint32, [vp]),
    "nmn_sharded_layout": (C.c_uint32, [vp]),
    "nmn_sharded_global_row": (C.c_uint64, [vp, C.c_uint32, C.c_uint64]),
    "nmn_sharded_rccl_ranks": (C.c_uint32, [vp]),
    "nmn_sharded_set_timing": (C.c_int32, [vp, C.c_int32]),
    "nmn_sharded_set_mirror": (C.c_int32, [vp, C.c_int32]),
    "nmn_sharded_last_gather_ms": (C.c_int32, [vp, C.POINTER(C.c_float)]),
    "nmn_sharded_coalesce_stats": (C.c_int32, [vp, vp, vp]),
    "nmn_index_search_dmask": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, vp, vp, vp, vp,
                                           C.POINTER(SearchStats)]),
    "nmn_index_search_dmask_hint": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_int32, vp, C.c_uint64, vp, vp, vp,
                                                C.POINTER(SearchStats)]),
}

_lib = None


class NeumannGpuError(RuntimeError):
    """A non-zero nmn_status.  `.status` is the code, the message mirrors VectorError's Display."""

    def __init__(self, status, detail=""):
        self.status = status
        msg = load().nmn_status_str(status).decode()
        if detail:
            msg = f"{msg}: {detail}"
        super().__init__(msg)


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the SONAME libneumann_gpu.so needs).  If ours resolved to /opt/rocm first, a later
    `import torch` would bring up a second runtime in the same process (torch then sees no GPU, and torch
    streams / tensors would belong to a different runtime than our kernels).  Loading torch's copy first
    makes the dynamic loader satisfy our NEEDED entry with it.  Without torch installed this is a no-op
    and the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared library (once).  Raises if it was not built — no fallback."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m neumann_amd.build` "
                "(neumann_amd has no CPU fallback for the SIMILAR TOP-K path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status != OK:
        detail = load().nmn_last_error().decode(errors="replace")
        raise NeumannGpuError(status, detail)
