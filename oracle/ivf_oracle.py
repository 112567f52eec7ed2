"""CPU restatement of the reference's IVF-Flat index and its k-means — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module (rule in oracle/nmn_oracle.c's header); the product runs the probe
and the k-means on the GPU (neumann_amd/csrc/nmn_ivf.hip: nmn_ivf_search / nmn_ivf_add / nmn_ivf_build).

Follows (paths relative to the reference root):
  tensor_store/src/ivf.rs            IVFIndex::train 222-274, add 276-316, search_with_nprobe 325-406,
                                     find_nearest_centroid 490-497, squared_euclidean 500-508,
                                     default_nprobe 46-56
  tensor_store/src/delta_vector.rs   KMeans::fit 737-777, init_random 781-800, init_kmeans_plusplus 805-853,
                                     nearest_centroid 856-863, update_centroids 867-893,
                                     euclidean_distance_sq 896-901
  vector_engine/src/lib.rs           search_with_ivf 2731-2766 (score = 1/(1+distance), 2762)

Arithmetic: every distance is a strictly sequential f32 sum of (x-y)*(x-y) (iterator `.sum()`, which folds
from -0.0); centroid means are sequential f32 sums divided by `count as f32`; the k-means++ threshold is
`(state as f32 / u64::MAX as f32) * total` in f32.  The reference's own IVF tests (ivf.rs:566-960) are
property tests (sizes add up, results sorted, self-query finds itself), so bit-level parity is pinned
only by this source-level restatement: "parity unpinned (bit level); pinned at property level" — the
same status as oracle/nmn_oracle.c.
"""
import math

import numpy as np

F = np.float32
MASK64 = (1 << 64) - 1
LCG_MUL = 6_364_136_223_846_793_005


def u64_to_f32(x):
    """Rust `u64 as f32`: round to nearest, ties to even (no double rounding through f64)."""
    x = int(x)
    if x == 0:
        return F(0.0)
    bl = x.bit_length()
    if bl <= 24:
        return F(x)
    shift = bl - 24
    top, rem = x >> shift, x & ((1 << shift) - 1)
    half = 1 << (shift - 1)
    if rem > half or (rem == half and (top & 1)):
        top += 1
    return F(math.ldexp(top, shift))  # top <= 2^24: exact in f32 after scaling


U64_MAX_F32 = u64_to_f32(MASK64)


def sq_dist_rows(A, b):
    """squared_euclidean(row, b) for every row of A: sequential over the dimension, vectorised over rows."""
    A = np.asarray(A, dtype=F)
    b = np.asarray(b, dtype=F)
    acc = np.full(A.shape[0], -0.0, dtype=F)
    for j in range(A.shape[1]):
        d = A[:, j] - b[j]
        acc = acc + d * d
    return acc


def sq_dist(a, b):
    return sq_dist_rows(np.asarray(a, dtype=F)[None, :], b)[0]


def first_min_index(d):
    """`min_by(partial_cmp .. unwrap_or(Equal))`: the fold keeps the earlier element unless the later one is
    strictly smaller; a NaN never displaces and is never displaced."""
    best = 0
    for i in range(1, len(d)):
        if d[i] < d[best]:
            best = i
    return best


def nearest_centroid(v, centroids):
    d = sq_dist_rows(centroids, v)
    return first_min_index(d) if np.isnan(d).any() else int(np.argmin(d))


class KMeansConfig:
    def __init__(self, max_iterations=100, convergence_threshold=1e-4, seed=42, init_method="kmeans++"):
        self.max_iterations = max_iterations
        self.convergence_threshold = F(convergence_threshold)
        self.seed = seed
        self.init_method = init_method  # "random" | "kmeans++"


def _lcg(state):
    return (state * LCG_MUL + 1) & MASK64


def init_random(vectors, k, seed):
    n = len(vectors)
    idx = list(range(n))
    state = seed
    for i in range(n - 1, 0, -1):
        state = _lcg(state)
        j = state % (i + 1)
        idx[i], idx[j] = idx[j], idx[i]
    return np.stack([vectors[i] for i in idx[:k]]).astype(F)


def init_kmeans_plusplus(vectors, k, seed):
    n = len(vectors)
    state = _lcg(seed)
    cents = [vectors[state % n].copy()]
    dist = np.full(n, np.finfo(F).max, dtype=F)
    for _ in range(1, k):
        dist = np.minimum(dist, sq_dist_rows(vectors, cents[-1]))
        total = F(-0.0)
        for x in dist:
            total = F(total + x)
        state = _lcg(state)
        if total == 0.0:
            idx = state % n
        else:
            threshold = F(F(u64_to_f32(state) / U64_MAX_F32) * total)
            cum = F(0.0)
            idx = 0
            for i, x in enumerate(dist):
                cum = F(cum + x)
                if cum >= threshold:
                    idx = i
                    break
        cents.append(vectors[idx].copy())
    return np.stack(cents).astype(F)


def update_centroids(vectors, assign, k):
    n, dim = vectors.shape
    sums = np.zeros((k, dim), dtype=F)
    counts = np.zeros(k, dtype=np.int64)
    for i in range(n):
        c = assign[i]
        counts[c] += 1
        sums[c] = sums[c] + vectors[i]
    out = np.zeros((k, dim), dtype=F)
    for c in range(k):
        if counts[c]:
            out[c] = sums[c] / F(counts[c])
    return out


def kmeans_fit(vectors, k, cfg):
    vectors = np.asarray(vectors, dtype=F)
    if vectors.shape[0] == 0 or k == 0:
        return np.zeros((0, vectors.shape[1] if vectors.ndim == 2 else 0), dtype=F)
    k = min(k, vectors.shape[0])
    cents = init_random(vectors, k, cfg.seed) if cfg.init_method == "random" else init_kmeans_plusplus(vectors, k, cfg.seed)
    for _ in range(cfg.max_iterations):
        D = np.stack([sq_dist_rows(vectors, c) for c in cents], axis=1)  # [n, k]
        if np.isnan(D).any():
            assign = [first_min_index(D[i]) for i in range(vectors.shape[0])]
        else:
            assign = np.argmin(D, axis=1)  # first minimal index, as min_by
        new = update_centroids(vectors, assign, k)
        movement = F(0.0)
        for old_c, new_c in zip(cents, new):
            m = np.sqrt(sq_dist(old_c, new_c))
            movement = max(movement, m)  # f32::max fold from 0.0
        cents = new
        if movement < cfg.convergence_threshold:
            break
    return cents


def default_nprobe(num_clusters):
    return int(math.ceil(float(np.sqrt(F(num_clusters)))))


class IVFFlat:
    """IVFIndex with IVFStorage::Flat."""

    def __init__(self, num_clusters=100, nprobe=None, kmeans=None):
        self.num_clusters = num_clusters
        self.nprobe = default_nprobe(num_clusters) if nprobe is None else nprobe
        self.kmeans = kmeans or KMeansConfig()
        self.centroids = None
        self.lists = []      # per cluster: list of ids, insertion order
        self.vectors = []    # id -> vector
        self.assign = []

    def train(self, vectors):
        vectors = np.asarray(vectors, dtype=F)
        if vectors.shape[0] == 0:
            return
        self.centroids = kmeans_fit(vectors, min(self.num_clusters, vectors.shape[0]), self.kmeans)
        self.lists = [[] for _ in range(len(self.centroids))]
        self.vectors, self.assign = [], []

    def add(self, v):
        v = np.asarray(v, dtype=F)
        c = nearest_centroid(v, self.centroids)
        vid = len(self.vectors)
        self.vectors.append(v)
        self.assign.append(c)
        self.lists[c].append(vid)
        return vid

    def cluster_sizes(self):
        return [len(l) for l in self.lists]

    def search(self, q, k, nprobe=None):
        """-> (ids, distances): the nprobe nearest lists, Euclidean distance ascending, stable."""
        if self.centroids is None or len(self.centroids) == 0 or k == 0:
            return [], []
        q = np.asarray(q, dtype=F)
        nprobe = self.nprobe if nprobe is None else nprobe
        cd = sq_dist_rows(self.centroids, q)
        order = sorted(range(len(cd)), key=lambda i: _sort_key(cd[i]))  # stable; NaN-free inputs assumed
        cand = []
        for c in order[:min(nprobe, len(order))]:
            ids = self.lists[c]
            if ids:
                d = np.sqrt(sq_dist_rows(np.stack([self.vectors[i] for i in ids]), q))
                cand.extend(zip(ids, d))
        cand.sort(key=lambda t: _sort_key(t[1]))  # stable: probe order, then list order
        cand = cand[:k]
        return [c[0] for c in cand], np.array([c[1] for c in cand], dtype=F)


def _sort_key(x):
    return float(x)


def ivf_score(distance):
    """search_with_ivf (lib.rs:2762): score = 1.0 / (1.0 + distance), f32."""
    return F(F(1.0) / F(F(1.0) + F(distance)))
