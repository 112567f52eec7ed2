"""IVF-Flat probe on the GPU — host wrapper of `nmn_ivf_*` (include/neumann_gpu.h).

Mirrors `tensor_store::ivf::IVFIndex` with `IVFStorage::Flat` (tensor_store/src/ivf.rs:160-406) for
the parts on the SIMILAR path: `add` (nearest-centroid assignment) and `search` / `search_with_nprobe`.
`GpuIvfFlat.build` trains on the GPU exactly as the reference's k-means does (`nmn_ivf_build`); the plain constructor
takes centroids trained elsewhere."""
import ctypes as C

import numpy as np

from . import _capi


class GpuIvfFlat:
    def __init__(self, centroids, capacity_rows, nprobe=None, device=-1):
        self._lib = _capi.load()
        c = np.ascontiguousarray(centroids, dtype=np.float32)
        assert c.ndim == 2 and c.shape[0] >= 1
        self.n_clusters, self.dim = int(c.shape[0]), int(c.shape[1])
        # default_nprobe (ivf.rs:46-56): ceil(sqrt(num_clusters)) computed in f32
        self.nprobe = int(np.ceil(np.sqrt(np.float32(self.n_clusters)))) if nprobe is None else int(nprobe)
        desc = _capi.IndexDesc(dim=self.dim, flags=0, capacity_rows=int(capacity_rows), row_base=0, device=int(device),
                               cand_cap=0)
        h = C.c_void_p()
        _capi.check(self._lib.nmn_ivf_create(C.byref(desc), C.c_void_p(c.ctypes.data), self.n_clusters, C.byref(h)))
        self._h = h

    @classmethod
    def build(cls, rows, num_clusters, nprobe=None, max_iterations=100, convergence_threshold=1e-4, seed=42,
              init_method="kmeans++", capacity_rows=None, device=-1):
        """IVFIndex::train(rows) + add(every row) on the GPU (`nmn_ivf_build`): k-means exactly as the reference runs it."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        r = np.ascontiguousarray(rows, dtype=np.float32)
        n, self.dim = int(r.shape[0]), int(r.shape[1])
        desc = _capi.IndexDesc(dim=self.dim, flags=0, capacity_rows=int(capacity_rows or n), row_base=0, device=int(device),
                               cand_cap=0)
        opt = _capi.KMeansOptions(max_iterations=int(max_iterations), convergence_threshold=float(convergence_threshold),
                                  seed=int(seed), init_method=0 if init_method == "random" else 1)
        h = C.c_void_p()
        _capi.check(self._lib.nmn_ivf_build(C.byref(desc), C.c_void_p(r.ctypes.data), n, int(num_clusters), C.byref(opt),
                                            C.byref(h)))
        self._h = h
        self.n_clusters = int(self._lib.nmn_ivf_clusters(h))
        self.nprobe = int(np.ceil(np.sqrt(np.float32(num_clusters)))) if nprobe is None else int(nprobe)
        return self

    def save(self, path):
        """Centroids, the list of every vector and the vectors in id order -> `path` (nmn_ivf_save)."""
        _capi.check(self._lib.nmn_ivf_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path, nprobe=None, capacity_rows=0, device=-1, max_file_bytes=0, max_entries=0):
        """nmn_ivf_load: no k-means, no re-assignment — lists and centroids come back exactly as saved."""
        self = cls.__new__(cls)
        self._lib = _capi.load()
        desc = _capi.IndexDesc(dim=0, flags=0, capacity_rows=int(capacity_rows), row_base=0, device=int(device), cand_cap=0)
        h = C.c_void_p()
        _capi.check(self._lib.nmn_ivf_load(str(path).encode(), C.byref(desc), int(max_file_bytes), int(max_entries), C.byref(h)))
        self._h = h
        self.n_clusters = int(self._lib.nmn_ivf_clusters(h))
        self.dim = int(self._lib.nmn_index_dim(self._lib.nmn_ivf_vectors(h)))
        self.nprobe = int(np.ceil(np.sqrt(np.float32(self.n_clusters)))) if nprobe is None else int(nprobe)
        return self

    def centroids(self):
        out = np.empty((self.n_clusters, self.dim), dtype=np.float32)
        _capi.check(self._lib.nmn_ivf_centroids(self._h, C.c_void_p(out.ctypes.data), out.size))
        return out

    def close(self):
        if self._h:
            self._lib.nmn_ivf_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def list_major_rows(self):
        """vectors covered by the list-major copy (0: every probe goes through the bitmap over the id-ordered rows)"""
        return int(self._lib.nmn_ivf_list_major_rows(self._h))

    def __len__(self):
        return int(self._lib.nmn_ivf_len(self._h))

    def add(self, rows):
        """IVFIndex::add for each row; returns the clusters chosen (ids are len-before .. len-after - 1)."""
        r = np.ascontiguousarray(rows, dtype=np.float32)
        if r.ndim == 1:
            r = r[None, :]
        assert r.shape[1] == self.dim
        out = np.empty(r.shape[0], dtype=np.uint32)
        _capi.check(self._lib.nmn_ivf_add(self._h, C.c_void_p(r.ctypes.data), r.shape[0], C.c_void_p(out.ctypes.data)))
        return out

    def cluster_sizes(self):
        out = np.zeros(self.n_clusters, dtype=np.uint64)
        _capi.check(self._lib.nmn_ivf_cluster_sizes(self._h, C.c_void_p(out.ctypes.data)))
        return out

    def search(self, queries, k, nprobe=None):
        """-> (ids u64 [nq,k], distances f32 [nq,k], counts u32 [nq]); IVFIndex::search_with_nprobe."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        assert q.shape[1] == self.dim
        nq, k = q.shape[0], int(k)
        ids = np.empty((nq, max(k, 1)), dtype=np.uint64)
        dist = np.empty((nq, max(k, 1)), dtype=np.float32)
        counts = np.empty(nq, dtype=np.uint32)
        _capi.check(self._lib.nmn_ivf_search(self._h, C.c_void_p(q.ctypes.data), nq, k,
                                             self.nprobe if nprobe is None else int(nprobe),
                                             C.c_void_p(ids.ctypes.data), C.c_void_p(dist.ctypes.data),
                                             C.c_void_p(counts.ctypes.data), None))
        return ids, dist, counts
