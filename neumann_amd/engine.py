"""Python face of the host-side `VectorEngine` mirror (include/neumann_engine.h).

Method names, argument order, error types and messages follow the reference's public Rust API
(vector_engine/src/lib.rs) so that the parity tests read like the reference's own tests:

    engine = VectorEngine()
    engine.store_embedding("a", [1.0, 0.0, 0.0])
    results = engine.search_similar([1.0, 0.0, 0.0], 3)     # -> [SearchResult(key, score), ...]

All logic lives in the C++ library; every search runs on the GPU (no CPU fallback).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from .flat_index import DistanceMetric

vp = C.c_void_p


class _EngineConfig(C.Structure):
    _fields_ = [("default_dimension", C.c_uint64), ("sparse_threshold", C.c_float),
                ("parallel_threshold", C.c_uint64), ("default_metric", C.c_int32),
                ("max_dimension", C.c_uint64), ("max_keys_per_scan", C.c_uint64),
                ("search_timeout_ms", C.c_int64), ("device", C.c_int32), ("cand_cap", C.c_uint32),
                ("max_index_file_bytes", C.c_int64), ("max_index_entries", C.c_int64),
                ("n_devices", C.c_uint32), ("devices", C.c_int32 * 16)]


class _Value(C.Structure):
    _fields_ = [("kind", C.c_int32), ("b", C.c_int32), ("i", C.c_int64), ("f", C.c_double), ("s", C.c_char_p)]


class _MetaField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("value", _Value)]


class _FilteredConfig(C.Structure):
    _fields_ = [("strategy", C.c_int32), ("selectivity_threshold", C.c_float), ("oversample_factor", C.c_uint64)]


ENGINE_SIGNATURES = {
    "nmn_engine_config_default": (None, [C.POINTER(_EngineConfig)]),
    "nmn_filtered_config_default": (None, [C.POINTER(_FilteredConfig)]),
    "nmn_engine_create": (C.c_int32, [C.POINTER(_EngineConfig), C.POINTER(vp)]),
    "nmn_engine_destroy": (None, [vp]),
    "nmn_engine_last_error": (C.c_char_p, []),
    "nmn_engine_store_embedding": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64]),
    "nmn_engine_store_embedding_with_metadata": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64,
                                                             C.POINTER(_MetaField), C.c_uint32]),
    "nmn_engine_get_embedding": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nmn_engine_delete_embedding": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_exists": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_count": (C.c_uint64, [vp]),
    "nmn_engine_list_keys": (vp, [vp]),
    "nmn_engine_list_keys_paginated": (vp, [vp, C.c_uint64, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "nmn_engine_clear": (C.c_int32, [vp, C.POINTER(C.c_uint64)]),
    "nmn_engine_batch_store": (C.c_int32, [vp, C.POINTER(C.c_char_p), vp, C.c_uint64, C.c_uint64]),
    "nmn_engine_search_similar": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_engine_search_probe": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]),
    "nmn_engine_get_metadata": (vp, [vp, C.c_char_p, C.POINTER(C.c_int32)]),
    "nmn_metalist_len": (C.c_uint64, [vp]),
    "nmn_metalist_name": (C.c_char_p, [vp, C.c_uint64]),
    "nmn_metalist_value": (C.c_int32, [vp, C.c_uint64, C.POINTER(_Value)]),
    "nmn_metalist_free": (None, [vp]),
    "nmn_engine_update_metadata": (C.c_int32, [vp, C.c_char_p, C.POINTER(_MetaField), C.c_uint32]),
    "nmn_engine_remove_metadata_field": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_has_metadata_field": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_get_metadata_field": (C.c_int32, [vp, C.c_char_p, C.c_char_p, C.POINTER(_Value), C.POINTER(C.c_int32)]),
    "nmn_engine_estimate_filter_selectivity": (C.c_int32, [vp, vp, C.POINTER(C.c_float)]),
    "nmn_engine_list_keys_matching": (vp, [vp, vp]),
    "nmn_engine_batch_delete": (C.c_int32, [vp, C.POINTER(C.c_char_p), C.c_uint64, C.POINTER(C.c_uint64)]),
    "nmn_engine_dimension": (C.c_uint64, [vp]),
    "nmn_engine_exists_in_collection": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_list_collection_keys": (vp, [vp, C.c_char_p]),
    "nmn_engine_search_similar_paginated": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32,
                                                        C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "nmn_engine_search_entities_paginated": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32,
                                                         C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "nmn_engine_blob_set_embedding": (C.c_int32, [vp, C.c_char_p, C.c_char_p, vp, C.c_uint64]),
    "nmn_engine_blob_remove": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_blob_search_by_embedding": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_engine_blob_similar": (C.c_int32, [vp, C.c_char_p, C.c_uint64, C.POINTER(vp)]),
    "nmn_results_aux": (C.c_char_p, [vp, C.c_uint64]),
    "nmn_ivf_options_default": (None, [vp]),
    "nmn_engine_build_ivf_index": (C.c_int32, [vp, vp, C.POINTER(vp)]),
    "nmn_engine_ivf_free": (None, [vp]),
    "nmn_engine_ivf_len": (C.c_uint64, [vp]),
    "nmn_engine_ivf_clusters": (C.c_uint32, [vp]),
    "nmn_engine_ivf_nprobe": (C.c_uint64, [vp]),
    "nmn_engine_ivf_key": (C.c_char_p, [vp, C.c_uint64]),
    "nmn_engine_ivf_centroids": (C.c_int32, [vp, vp, C.c_uint64]),
    "nmn_engine_ivf_cluster_sizes": (C.c_int32, [vp, vp]),
    "nmn_engine_search_with_ivf": (C.c_int32, [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_engine_set_entity_embedding": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64]),
    "nmn_engine_get_entity_embedding": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nmn_engine_entity_has_embedding": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_remove_entity_embedding": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_scan_entities_with_embeddings": (vp, [vp]),
    "nmn_engine_count_entities_with_embeddings": (C.c_uint64, [vp]),
    "nmn_engine_search_entities": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_engine_search_similar_with_metric": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(vp)]),
    "nmn_engine_search_similar_filtered": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, vp,
                                                       C.POINTER(_FilteredConfig), C.POINTER(vp)]),
    "nmn_engine_compute_similarity": (C.c_int32, [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_float)]),
    "nmn_engine_create_collection": (C.c_int32, [vp, C.c_char_p, C.c_uint64, C.c_int32]),
    "nmn_engine_delete_collection": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_collection_exists": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_collection_count": (C.c_uint64, [vp, C.c_char_p]),
    "nmn_engine_list_collections": (vp, [vp]),
    "nmn_engine_store_in_collection": (C.c_int32, [vp, C.c_char_p, C.c_char_p, vp, C.c_uint64,
                                                   C.POINTER(_MetaField), C.c_uint32]),
    "nmn_engine_get_from_collection": (C.c_int32, [vp, C.c_char_p, C.c_char_p, vp, C.c_uint64,
                                                   C.POINTER(C.c_uint64)]),
    "nmn_engine_delete_from_collection": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_search_in_collection": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "nmn_engine_search_filtered_in_collection": (C.c_int32, [vp, C.c_char_p, vp, C.c_uint64, C.c_uint64, vp,
                                                             C.POINTER(_FilteredConfig), C.POINTER(vp)]),
    "nmn_results_len": (C.c_uint64, [vp]),
    "nmn_results_key": (C.c_char_p, [vp, C.c_uint64]),
    "nmn_results_score": (C.c_float, [vp, C.c_uint64]),
    "nmn_results_free": (None, [vp]),
    "nmn_strlist_len": (C.c_uint64, [vp]),
    "nmn_strlist_get": (C.c_char_p, [vp, C.c_uint64]),
    "nmn_strlist_free": (None, [vp]),
    "nmn_filter_cmp": (vp, [C.c_int32, C.c_char_p, C.POINTER(_Value)]),
    "nmn_filter_and": (vp, [vp, vp]),
    "nmn_filter_or": (vp, [vp, vp]),
    "nmn_filter_true": (vp, []),
    "nmn_filter_exists": (vp, [C.c_char_p]),
    "nmn_filter_contains": (vp, [C.c_char_p, C.c_char_p]),
    "nmn_filter_starts_with": (vp, [C.c_char_p, C.c_char_p]),
    "nmn_filter_in": (vp, [C.c_char_p, C.POINTER(_Value), C.c_uint32]),
    "nmn_filter_free": (None, [vp]),
    "nmn_engine_count_matching": (C.c_uint64, [vp, vp]),
    "nmn_engine_mirror_builds": (C.c_uint64, [vp]),
    "nmn_engine_mirror_shard_rows": (C.c_uint32, [vp, C.c_uint64, vp, C.c_uint32]),
    "nmn_engine_mirror_hbm_bytes": (C.c_uint32, [vp, C.c_uint64, vp]),
    "nmn_engine_device_filter_evals": (C.c_uint64, [vp]),
    "nmn_engine_column_builds": (C.c_uint64, [vp]),
    "nmn_engine_mirror_cached": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_save_index": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_load_index": (C.c_int32, [vp, C.c_char_p, C.c_char_p, C.c_uint64]),
    "nmn_engine_save_index_binary": (C.c_int32, [vp, C.c_char_p, C.c_char_p]),
    "nmn_engine_load_index_binary": (C.c_int32, [vp, C.c_char_p, C.c_char_p, C.c_uint64]),
    "nmn_engine_save_all_indices": (vp, [vp, C.c_char_p, C.POINTER(C.c_int32)]),
    "nmn_engine_load_all_indices": (vp, [vp, C.c_char_p, C.POINTER(C.c_int32)]),
    "nmn_engine_ivf_save": (C.c_int32, [vp, C.c_char_p]),
    "nmn_engine_ivf_load": (C.c_int32, [vp, C.c_char_p, C.POINTER(vp)]),
}

_bound = False


def _lib():
    global _bound
    lib = _capi.load()
    if not _bound:
        for name, (res, args) in ENGINE_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return lib


class VectorError(Exception):
    """vector_engine::VectorError (lib.rs:101-149).  `.kind` is the variant name, str() the Display text."""
    KINDS = {
        _capi.ERR_NOT_FOUND: "NotFound", _capi.ERR_DIMENSION_MISMATCH: "DimensionMismatch",
        _capi.ERR_EMPTY_VECTOR: "EmptyVector", _capi.ERR_INVALID_TOP_K: "InvalidTopK",
        _capi.ERR_STORAGE: "StorageError", _capi.ERR_CONFIGURATION: "ConfigurationError",
        _capi.ERR_COLLECTION_EXISTS: "CollectionExists", _capi.ERR_COLLECTION_NOT_FOUND: "CollectionNotFound",
        _capi.ERR_SEARCH_TIMEOUT: "SearchTimeout", _capi.ERR_IO: "IoError", _capi.ERR_SERIALIZATION: "SerializationError",
    }

    def __init__(self, status, message):
        self.status = status
        self.kind = self.KINDS.get(status, "StorageError")
        super().__init__(message)


def _check(status):
    if status != _capi.OK:
        raise VectorError(status, _lib().nmn_engine_last_error().decode(errors="replace"))


@dataclass
class SearchResult:
    """vector_engine::SearchResult (lib.rs:252-266)."""
    key: str
    score: float


@dataclass
class Pagination:
    """vector_engine::Pagination (lib.rs:1052-1085); limit None = no limit."""
    skip: int = 0
    limit: int = None
    count_total: bool = False

    def with_total(self):
        return Pagination(self.skip, self.limit, True)


@dataclass
class PagedResult:
    """vector_engine::PagedResult (lib.rs:1087-1112)."""
    items: list
    total_count: int
    has_more: bool


def _py_value(v):
    return {0: None, 1: bool(v.b), 2: int(v.i), 3: float(v.f)}.get(v.kind, v.s.decode() if v.s is not None else "")


@dataclass
class SimilarArtifact:
    """tensor_blob::SimilarArtifact (tensor_blob/src/metadata.rs:155-162)."""
    id: str
    filename: str
    similarity: float


@dataclass
class VectorEngineConfig:
    """vector_engine::VectorEngineConfig (lib.rs:626-664); None = the reference's Option::None."""
    default_dimension: int = None
    sparse_threshold: float = 0.5
    parallel_threshold: int = 5000
    default_metric: DistanceMetric = DistanceMetric.Cosine
    max_dimension: int = None
    max_keys_per_scan: int = None
    search_timeout: float = None  # seconds
    device: int = -1
    cand_cap: int = 0
    max_index_file_bytes: int = 100 * 1024 * 1024   # lib.rs:660; None = no limit
    max_index_entries: int = 1_000_000              # lib.rs:661; None = no limit
    devices: tuple = ()   # GPU ordinals; two or more = every collection is one index sharded over them (nmn_sharded_*)


@dataclass
class VectorCollectionConfig:
    """vector_engine::VectorCollectionConfig (lib.rs:455-499)."""
    dimension: int = None
    distance_metric: DistanceMetric = DistanceMetric.Cosine

    def with_dimension(self, dim):
        return VectorCollectionConfig(dim, self.distance_metric)

    def with_metric(self, metric):
        return VectorCollectionConfig(self.dimension, metric)


class FilterStrategy:
    Auto, PreFilter, PostFilter = 0, 1, 2


@dataclass
class FilteredSearchConfig:
    """vector_engine::FilteredSearchConfig (lib.rs:399-449)."""
    strategy: int = FilterStrategy.Auto
    selectivity_threshold: float = 0.1
    oversample_factor: int = 3

    @staticmethod
    def pre_filter():
        return FilteredSearchConfig(FilterStrategy.PreFilter)

    @staticmethod
    def post_filter():
        return FilteredSearchConfig(FilterStrategy.PostFilter)


def _value(v):
    out = _Value()
    if v is None:
        out.kind = 0
    elif isinstance(v, bool):
        out.kind, out.b = 1, int(v)
    elif isinstance(v, (int, np.integer)):
        out.kind, out.i = 2, int(v)
    elif isinstance(v, (float, np.floating)):
        out.kind, out.f = 3, float(v)
    elif isinstance(v, str):
        out.kind, out.s = 4, v.encode()
    else:
        raise TypeError(f"unsupported metadata/filter value {v!r}")
    return out


class _IvfOptions(C.Structure):
    _fields_ = [("num_clusters", C.c_uint64), ("nprobe", C.c_uint64), ("max_iterations", C.c_uint64),
                ("convergence_threshold", C.c_float), ("seed", C.c_uint64), ("init_method", C.c_int32)]


@dataclass
class IVFBuildOptions:
    """vector_engine::IVFBuildOptions with IVFConfig::flat + KMeansConfig (lib.rs:941-1000, ivf.rs:61-147,
    delta_vector.rs:691-711).  nprobe None = default_nprobe(num_clusters)."""
    num_clusters: int = 100
    nprobe: int = None
    max_iterations: int = 100
    convergence_threshold: float = 1e-4
    seed: int = 42
    init_method: str = "kmeans++"   # or "random"

    @staticmethod
    def flat(num_clusters):
        return IVFBuildOptions(num_clusters=num_clusters)


class IVFIndex:
    """The (IVFIndex, key_mapping) pair build_ivf_index returns; vectors and lists live on the GPU."""

    def __init__(self, handle):
        self._h = handle
        n = int(_lib().nmn_engine_ivf_len(handle))
        self.keys = [_lib().nmn_engine_ivf_key(handle, i).decode() for i in range(n)]

    def __len__(self):
        return int(_lib().nmn_engine_ivf_len(self._h))

    @property
    def num_clusters(self):
        return int(_lib().nmn_engine_ivf_clusters(self._h))

    @property
    def nprobe(self):
        return int(_lib().nmn_engine_ivf_nprobe(self._h))

    def is_trained(self):
        return self.num_clusters > 0

    def centroids(self, dim):
        out = np.empty((self.num_clusters, dim), dtype=np.float32)
        if out.size:
            _check(_lib().nmn_engine_ivf_centroids(self._h, C.c_void_p(out.ctypes.data), out.size))
        return out

    def cluster_sizes(self):
        out = np.zeros(self.num_clusters, dtype=np.uint64)
        if out.size:
            _check(_lib().nmn_engine_ivf_cluster_sizes(self._h, C.c_void_p(out.ctypes.data)))
        return out

    def close(self):
        if self._h:
            _lib().nmn_engine_ivf_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilterCondition:
    """vector_engine::FilterCondition (lib.rs:296-340), built from the same constructors."""

    def __init__(self, build):
        self._build = build  # () -> owned nmn_filter*

    @staticmethod
    def _cmp(op, field, value):
        def build():
            v = _value(value)
            return _lib().nmn_filter_cmp(op, field.encode(), C.byref(v))
        return FilterCondition(build)

    Eq = staticmethod(lambda f, v: FilterCondition._cmp(0, f, v))
    Ne = staticmethod(lambda f, v: FilterCondition._cmp(1, f, v))
    Lt = staticmethod(lambda f, v: FilterCondition._cmp(2, f, v))
    Le = staticmethod(lambda f, v: FilterCondition._cmp(3, f, v))
    Gt = staticmethod(lambda f, v: FilterCondition._cmp(4, f, v))
    Ge = staticmethod(lambda f, v: FilterCondition._cmp(5, f, v))
    TRUE = None  # set below

    @staticmethod
    def Exists(field):
        return FilterCondition(lambda: _lib().nmn_filter_exists(field.encode()))

    @staticmethod
    def Contains(field, s):
        return FilterCondition(lambda: _lib().nmn_filter_contains(field.encode(), s.encode()))

    @staticmethod
    def StartsWith(field, s):
        return FilterCondition(lambda: _lib().nmn_filter_starts_with(field.encode(), s.encode()))

    @staticmethod
    def In(field, values):
        def build():
            arr = (_Value * max(len(values), 1))(*[_value(v) for v in values])
            return _lib().nmn_filter_in(field.encode(), arr, len(values))
        return FilterCondition(build)

    def and_(self, other):
        return FilterCondition(lambda: _lib().nmn_filter_and(self._build(), other._build()))

    def or_(self, other):
        return FilterCondition(lambda: _lib().nmn_filter_or(self._build(), other._build()))


FilterCondition.TRUE = FilterCondition(lambda: _lib().nmn_filter_true())


class _OwnedFilter:
    def __init__(self, cond):
        self.h = cond._build()
        if not self.h:
            raise ValueError("invalid filter")

    def __enter__(self):
        return self.h

    def __exit__(self, *exc):
        _lib().nmn_filter_free(self.h)


def _vec(v):
    a = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    return a, C.c_void_p(a.ctypes.data), a.size


def _meta_array(metadata):
    if not metadata:
        return None, 0, None
    keep = []
    arr = (_MetaField * len(metadata))()
    for i, (k, v) in enumerate(metadata.items()):
        kb = k.encode()
        keep.append(kb)
        arr[i].name = kb
        arr[i].value = _value(v)
    return arr, len(metadata), keep


class VectorEngine:
    """Host-side mirror of vector_engine::VectorEngine for the SIMILAR TOP-K path."""

    def __init__(self, config=None):
        lib = _lib()
        cfg = _EngineConfig()
        lib.nmn_engine_config_default(C.byref(cfg))
        if config is not None:
            cfg.default_dimension = config.default_dimension or 0
            cfg.sparse_threshold = config.sparse_threshold
            cfg.parallel_threshold = config.parallel_threshold
            cfg.default_metric = int(config.default_metric)
            cfg.max_dimension = config.max_dimension or 0
            if config.max_keys_per_scan is not None and int(config.max_keys_per_scan) <= 0:  # lib.rs:728-733
                raise VectorError(_capi.ERR_CONFIGURATION, "Configuration error: max_keys_per_scan must be greater than 0")
            cfg.max_keys_per_scan = config.max_keys_per_scan or 0
            cfg.search_timeout_ms = -1 if config.search_timeout is None else int(config.search_timeout * 1000)
            cfg.device = config.device
            cfg.cand_cap = config.cand_cap
            cfg.max_index_file_bytes = -1 if config.max_index_file_bytes is None else int(config.max_index_file_bytes)
            cfg.max_index_entries = -1 if config.max_index_entries is None else int(config.max_index_entries)
            devs = tuple(config.devices or ())
            if len(devs) > 16:
                raise VectorError(_capi.ERR_CONFIGURATION, "Configuration error: n_devices exceeds NMN_ENGINE_MAX_DEVICES")
            cfg.n_devices = len(devs)
            for i, d in enumerate(devs):
                cfg.devices[i] = int(d)
        self._h = vp()
        _check(lib.nmn_engine_create(C.byref(cfg), C.byref(self._h)))

    @classmethod
    def with_config(cls, config):
        return cls(config)

    def close(self):
        if getattr(self, "_h", None):
            _lib().nmn_engine_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- results helpers ------------------------------------------------------------------------
    @staticmethod
    def _take_results(h):
        lib = _lib()
        try:
            n = lib.nmn_results_len(h)
            return [SearchResult(lib.nmn_results_key(h, i).decode(), float(np.float32(lib.nmn_results_score(h, i))))
                    for i in range(n)]
        finally:
            lib.nmn_results_free(h)

    @staticmethod
    def _take_list(h):
        lib = _lib()
        try:
            return [lib.nmn_strlist_get(h, i).decode() for i in range(lib.nmn_strlist_len(h))]
        finally:
            lib.nmn_strlist_free(h)

    # -- default collection ---------------------------------------------------------------------
    def store_embedding(self, key, vector):
        a, p, n = _vec(vector)
        _check(_lib().nmn_engine_store_embedding(self._h, key.encode(), p, n))

    def store_embedding_with_metadata(self, key, vector, metadata):
        a, p, n = _vec(vector)
        arr, m, keep = _meta_array(metadata)
        _check(_lib().nmn_engine_store_embedding_with_metadata(self._h, key.encode(), p, n, arr, m))

    def batch_store_embeddings(self, keys, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        ks = (C.c_char_p * len(keys))(*[k.encode() for k in keys])
        _check(_lib().nmn_engine_batch_store(self._h, ks, C.c_void_p(rows.ctypes.data), rows.shape[0], rows.shape[1]))

    def get_embedding(self, key):
        dim = C.c_uint64()
        _check(_lib().nmn_engine_get_embedding(self._h, key.encode(), None, 0, C.byref(dim)))
        out = np.empty(dim.value, dtype=np.float32)
        _check(_lib().nmn_engine_get_embedding(self._h, key.encode(), C.c_void_p(out.ctypes.data), out.size, C.byref(dim)))
        return out

    def delete_embedding(self, key):
        _check(_lib().nmn_engine_delete_embedding(self._h, key.encode()))

    def exists(self, key):
        return bool(_lib().nmn_engine_exists(self._h, key.encode()))

    def count(self):
        return int(_lib().nmn_engine_count(self._h))

    def list_keys(self):
        return self._take_list(_lib().nmn_engine_list_keys(self._h))

    def list_keys_bounded(self):
        return self.list_keys()

    def list_keys_paginated(self, pagination):
        total, more = C.c_int64(), C.c_int32()
        h = _lib().nmn_engine_list_keys_paginated(self._h, int(pagination.skip),
                                                  -1 if pagination.limit is None else int(pagination.limit),
                                                  int(pagination.count_total), C.byref(total), C.byref(more))
        return PagedResult(self._take_list(h), None if total.value < 0 else int(total.value), bool(more.value))

    def clear(self):
        n = C.c_uint64()
        _check(_lib().nmn_engine_clear(self._h, C.byref(n)))
        return n.value

    def search_similar(self, query, top_k):
        a, p, n = _vec(query)
        h = vp()
        _check(_lib().nmn_engine_search_similar(self._h, p, n, int(top_k), C.byref(h)))
        return self._take_results(h)

    def search_probe(self, queries, top_k, calls):
        """Microseconds of each of `calls` back-to-back search_similar calls made natively by one host thread (measurement aid,
        nmn_engine_search_probe): query i % len(queries) for call i."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        out = np.empty(int(calls), dtype=np.float32)
        _check(_lib().nmn_engine_search_probe(self._h, q.ctypes.data_as(vp), q.shape[0], q.shape[1], int(top_k), int(calls),
                                              out.ctypes.data_as(vp)))
        return out

    # ---- tensor_blob artifact similarity (tensor_blob/src/lib.rs:520-625) ----
    def blob_set_embedding(self, artifact_id, filename, embedding):
        a, p, n = _vec(embedding)
        _check(_lib().nmn_engine_blob_set_embedding(self._h, artifact_id.encode(), filename.encode(), p, n))

    def blob_remove(self, artifact_id):
        _check(_lib().nmn_engine_blob_remove(self._h, artifact_id.encode()))

    @staticmethod
    def _take_artifacts(h):
        lib = _lib()
        try:
            n = lib.nmn_results_len(h)
            return [SimilarArtifact(lib.nmn_results_key(h, i).decode(), lib.nmn_results_aux(h, i).decode(),
                                    float(np.float32(lib.nmn_results_score(h, i)))) for i in range(n)]
        finally:
            lib.nmn_results_free(h)

    def blob_search_by_embedding(self, embedding, k):
        a, p, n = _vec(embedding)
        h = vp()
        _check(_lib().nmn_engine_blob_search_by_embedding(self._h, p if n else None, n, int(k), C.byref(h)))
        return self._take_artifacts(h)

    def blob_similar(self, artifact_id, k):
        h = vp()
        _check(_lib().nmn_engine_blob_similar(self._h, artifact_id.encode(), int(k), C.byref(h)))
        return self._take_artifacts(h)

    # ---- IVF (lib.rs:2641-2812) ----
    def build_ivf_index(self, options=None):
        """-> (IVFIndex, key_mapping) like the reference; k-means as the reference runs it, lists on the GPU."""
        o = options or IVFBuildOptions()
        co = _IvfOptions(num_clusters=o.num_clusters, nprobe=o.nprobe or 0, max_iterations=o.max_iterations,
                         convergence_threshold=o.convergence_threshold, seed=o.seed,
                         init_method=0 if o.init_method == "random" else 1)
        h = vp()
        _check(_lib().nmn_engine_build_ivf_index(self._h, C.byref(co), C.byref(h)))
        index = IVFIndex(h)
        return index, index.keys

    # -- index persistence (lib.rs:3733-4000) -----------------------------------------------------
    DEFAULT_COLLECTION = "default"

    def save_index(self, collection, path):
        """PersistentVectorIndex as serde_json writes it (lib.rs:3794-3801)."""
        _check(_lib().nmn_engine_save_index(self._h, collection.encode(), str(path).encode()))

    def load_index(self, path):
        """-> the collection's name (lib.rs:3827-3866); max_index_file_bytes / max_index_entries apply."""
        buf = C.create_string_buffer(4096)
        _check(_lib().nmn_engine_load_index(self._h, str(path).encode(), buf, len(buf)))
        return buf.value.decode()

    def save_index_binary(self, collection, path):
        """The snapshot with the vectors as flat shard sections (the device layout), lib.rs:3811-3817."""
        _check(_lib().nmn_engine_save_index_binary(self._h, collection.encode(), str(path).encode()))

    def load_index_binary(self, path):
        buf = C.create_string_buffer(4096)
        _check(_lib().nmn_engine_load_index_binary(self._h, str(path).encode(), buf, len(buf)))
        return buf.value.decode()

    def _take_strlist(self, h, st):
        _check(st.value)
        lib = _lib()
        try:
            return [lib.nmn_strlist_get(h, i).decode() for i in range(lib.nmn_strlist_len(h))]
        finally:
            lib.nmn_strlist_free(h)

    def save_all_indices(self, directory):
        st = C.c_int32()
        h = _lib().nmn_engine_save_all_indices(self._h, str(directory).encode(), C.byref(st))
        return self._take_strlist(h, st)

    def load_all_indices(self, directory):
        st = C.c_int32()
        h = _lib().nmn_engine_load_all_indices(self._h, str(directory).encode(), C.byref(st))
        return self._take_strlist(h, st)

    def save_ivf_index(self, index, path):
        """The (IVFIndex, key_mapping) pair with trained centroids and lists: restored without k-means."""
        _check(_lib().nmn_engine_ivf_save(index._h, str(path).encode()))

    def load_ivf_index(self, path):
        h = vp()
        _check(_lib().nmn_engine_ivf_load(self._h, str(path).encode(), C.byref(h)))
        index = IVFIndex(h)
        return index, index.keys

    def build_ivf_index_default(self):
        return self.build_ivf_index(IVFBuildOptions())

    def search_with_ivf(self, index, key_mapping, query, top_k, _nprobe=0):
        a, p, n = _vec(query)
        h = vp()
        _check(_lib().nmn_engine_search_with_ivf(self._h, index._h, p, n, int(top_k), int(_nprobe), C.byref(h)))
        res = self._take_results(h)
        if key_mapping is not index.keys and list(key_mapping) != index.keys:
            # `key_mapping.get(vector_id)` (lib.rs:2735-2743) with a caller-supplied mapping
            ids = {k: i for i, k in enumerate(index.keys)}
            res = [SearchResult(key_mapping[ids[r.key]], r.score) for r in res if ids[r.key] < len(key_mapping)]
        return res

    def search_with_ivf_nprobe(self, index, key_mapping, query, top_k, nprobe):
        if nprobe <= 0:
            return []  # `nprobe.min(len)` = 0 clusters probed (ivf.rs:339-343)
        return self.search_with_ivf(index, key_mapping, query, top_k, _nprobe=nprobe)

    # ---- unified entity mode (lib.rs:3060-3237) ----
    def set_entity_embedding(self, entity_key, vector):
        a, p, n = _vec(vector)
        _check(_lib().nmn_engine_set_entity_embedding(self._h, entity_key.encode(), p, n))

    def get_entity_embedding(self, entity_key):
        dim = C.c_uint64()
        _check(_lib().nmn_engine_get_entity_embedding(self._h, entity_key.encode(), None, 0, C.byref(dim)))
        out = np.empty(dim.value, dtype=np.float32)
        _check(_lib().nmn_engine_get_entity_embedding(self._h, entity_key.encode(), C.c_void_p(out.ctypes.data),
                                                      out.size, C.byref(dim)))
        return out

    def entity_has_embedding(self, entity_key):
        return bool(_lib().nmn_engine_entity_has_embedding(self._h, entity_key.encode()))

    def remove_entity_embedding(self, entity_key):
        _check(_lib().nmn_engine_remove_entity_embedding(self._h, entity_key.encode()))

    def scan_entities_with_embeddings(self):
        return self._take_list(_lib().nmn_engine_scan_entities_with_embeddings(self._h))

    def count_entities_with_embeddings(self):
        return int(_lib().nmn_engine_count_entities_with_embeddings(self._h))

    def search_entities(self, query, top_k):
        a, p, n = _vec(query)
        h = vp()
        _check(_lib().nmn_engine_search_entities(self._h, p, n, int(top_k), C.byref(h)))
        return self._take_results(h)

    def search_similar_with_metric(self, query, top_k, metric):
        a, p, n = _vec(query)
        h = vp()
        _check(_lib().nmn_engine_search_similar_with_metric(self._h, p, n, int(top_k), int(metric), C.byref(h)))
        return self._take_results(h)

    def search_similar_filtered(self, query, top_k, filter, config=None):
        a, p, n = _vec(query)
        h = vp()
        cfg = None
        if config is not None:
            cfg = _FilteredConfig(config.strategy, config.selectivity_threshold, config.oversample_factor)
        with _OwnedFilter(filter) as f:
            _check(_lib().nmn_engine_search_similar_filtered(self._h, p, n, int(top_k), f,
                                                             None if cfg is None else C.byref(cfg), C.byref(h)))
        return self._take_results(h)

    def compute_similarity(self, a, b):
        a1, pa, na = _vec(a)
        b1, pb, nb = _vec(b)
        out = C.c_float()
        _check(_lib().nmn_engine_compute_similarity(self._h, pa, na, pb, nb, C.byref(out)))
        return float(np.float32(out.value))

    # ---- metadata CRUD (lib.rs:3311-3385) ----
    def get_metadata(self, key):
        st = C.c_int32()
        h = _lib().nmn_engine_get_metadata(self._h, key.encode(), C.byref(st))
        _check(st.value)
        try:
            out = {}
            for i in range(_lib().nmn_metalist_len(h)):
                v = _Value()
                _check(_lib().nmn_metalist_value(h, i, C.byref(v)))
                out[_lib().nmn_metalist_name(h, i).decode()] = _py_value(v)
            return out
        finally:
            _lib().nmn_metalist_free(h)

    def update_metadata(self, key, metadata):
        arr, m, keep = _meta_array(metadata)
        _check(_lib().nmn_engine_update_metadata(self._h, key.encode(), arr, m))

    def remove_metadata_field(self, key, field):
        _check(_lib().nmn_engine_remove_metadata_field(self._h, key.encode(), field.encode()))

    def has_metadata_field(self, key, field):
        return bool(_lib().nmn_engine_has_metadata_field(self._h, key.encode(), field.encode()))

    def get_metadata_field(self, key, field):
        v, present = _Value(), C.c_int32()
        _check(_lib().nmn_engine_get_metadata_field(self._h, key.encode(), field.encode(), C.byref(v), C.byref(present)))
        return _py_value(v) if present.value else None

    def estimate_filter_selectivity(self, filter):
        out = C.c_float()
        with _OwnedFilter(filter) as f:
            _check(_lib().nmn_engine_estimate_filter_selectivity(self._h, f, C.byref(out)))
        return float(out.value)

    def list_keys_matching(self, filter):
        with _OwnedFilter(filter) as f:
            return self._take_list(_lib().nmn_engine_list_keys_matching(self._h, f))

    def batch_delete_embeddings(self, keys):
        ks = (C.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
        n = C.c_uint64()
        _check(_lib().nmn_engine_batch_delete(self._h, ks, len(keys), C.byref(n)))
        return int(n.value)

    def dimension(self):
        d = int(_lib().nmn_engine_dimension(self._h))
        return d or None

    def exists_in_collection(self, collection, key):
        return bool(_lib().nmn_engine_exists_in_collection(self._h, collection.encode(), key.encode()))

    def list_collection_keys(self, collection):
        return self._take_list(_lib().nmn_engine_list_collection_keys(self._h, collection.encode()))

    def _paginated(self, fn, query, top_k, pagination):
        a, p, n = _vec(query)
        h, total, more = vp(), C.c_int64(), C.c_int32()
        _check(fn(self._h, p, n, int(top_k), int(pagination.skip), -1 if pagination.limit is None else int(pagination.limit),
                  int(pagination.count_total), C.byref(h), C.byref(total), C.byref(more)))
        return PagedResult(self._take_results(h), None if total.value < 0 else int(total.value), bool(more.value))

    def search_similar_paginated(self, query, top_k, pagination):
        return self._paginated(_lib().nmn_engine_search_similar_paginated, query, top_k, pagination)

    def search_entities_paginated(self, query, top_k, pagination):
        return self._paginated(_lib().nmn_engine_search_entities_paginated, query, top_k, pagination)

    def count_matching(self, filter):
        with _OwnedFilter(filter) as f:
            return int(_lib().nmn_engine_count_matching(self._h, f))

    # -- collections ----------------------------------------------------------------------------
    def create_collection(self, name, config=None):
        config = config or VectorCollectionConfig()
        _check(_lib().nmn_engine_create_collection(self._h, name.encode(), config.dimension or 0,
                                                   int(config.distance_metric)))

    def delete_collection(self, name):
        _check(_lib().nmn_engine_delete_collection(self._h, name.encode()))

    def collection_exists(self, name):
        return bool(_lib().nmn_engine_collection_exists(self._h, name.encode()))

    def collection_count(self, name):
        return int(_lib().nmn_engine_collection_count(self._h, name.encode()))

    def list_collections(self):
        return self._take_list(_lib().nmn_engine_list_collections(self._h))

    def store_in_collection(self, collection, key, vector):
        self.store_in_collection_with_metadata(collection, key, vector, None)

    def store_in_collection_with_metadata(self, collection, key, vector, metadata):
        a, p, n = _vec(vector)
        arr, m, keep = _meta_array(metadata)
        _check(_lib().nmn_engine_store_in_collection(self._h, collection.encode(), key.encode(), p, n, arr, m))

    def get_from_collection(self, collection, key):
        dim = C.c_uint64()
        _check(_lib().nmn_engine_get_from_collection(self._h, collection.encode(), key.encode(), None, 0, C.byref(dim)))
        out = np.empty(dim.value, dtype=np.float32)
        _check(_lib().nmn_engine_get_from_collection(self._h, collection.encode(), key.encode(),
                                                     C.c_void_p(out.ctypes.data), out.size, C.byref(dim)))
        return out

    def delete_from_collection(self, collection, key):
        _check(_lib().nmn_engine_delete_from_collection(self._h, collection.encode(), key.encode()))

    def search_in_collection(self, collection, query, top_k):
        a, p, n = _vec(query)
        h = vp()
        _check(_lib().nmn_engine_search_in_collection(self._h, collection.encode(), p, n, int(top_k), C.byref(h)))
        return self._take_results(h)

    def search_filtered_in_collection(self, collection, query, top_k, filter, config=None):
        a, p, n = _vec(query)
        h = vp()
        cfg = None
        if config is not None:
            cfg = _FilteredConfig(config.strategy, config.selectivity_threshold, config.oversample_factor)
        with _OwnedFilter(filter) as f:
            _check(_lib().nmn_engine_search_filtered_in_collection(
                self._h, collection.encode(), p, n, int(top_k), f, None if cfg is None else C.byref(cfg), C.byref(h)))
        return self._take_results(h)

    # -- mirror bookkeeping (cache protocol tests) ------------------------------------------------
    def mirror_shard_rows(self, dim):
        out = (C.c_uint64 * 64)()
        n = int(_lib().nmn_engine_mirror_shard_rows(self._h, int(dim), out, 64))
        return [int(out[i]) for i in range(min(n, 64))]

    def mirror_hbm_bytes(self, dim):
        """(f32 row bytes, mirror bytes, per-row bytes) of the default collection's GPU mirror of `dim`; None without one."""
        out = (C.c_uint64 * 3)()
        n = int(_lib().nmn_engine_mirror_hbm_bytes(self._h, int(dim), out))
        return (int(out[0]), int(out[1]), int(out[2])) if n else None

    def mirror_builds(self):
        return int(_lib().nmn_engine_mirror_builds(self._h))

    def device_filter_evals(self):
        """Pre-filter predicates evaluated by the GPU predicate kernel so far."""
        return int(_lib().nmn_engine_device_filter_evals(self._h))

    def column_builds(self):
        """Times a metadata column set was (re)built from the store."""
        return int(_lib().nmn_engine_column_builds(self._h))

    def mirror_cached(self, collection=None):
        return bool(_lib().nmn_engine_mirror_cached(self._h, None if collection is None else collection.encode()))
