"""Pins oracle/ivf_oracle.py at the level the reference pins its own IVF index: the property tests of
tensor_store/src/ivf.rs:656-770 and delta_vector.rs' k-means tests, plus the integer/rounding helpers."""
import numpy as np

from oracle import ivf_oracle as io

F = np.float32


def create_test_vectors(n, dim):  # ivf.rs:569-577
    return np.array([[((i * 7 + j * 13) % 100) / 100.0 for j in range(dim)] for i in range(n)], dtype=F)


def fast(num_clusters, nprobe=None):  # fast_test_config, ivf.rs:579-596
    return io.IVFFlat(num_clusters, nprobe=nprobe, kmeans=io.KMeansConfig(2, 1.0, 42, "random"))


def test_train_add_search_properties():
    V = create_test_vectors(20, 16)
    ivf = fast(4)
    ivf.train(V)                                            # ivf_train_creates_clusters
    assert ivf.centroids.shape == (4, 16)
    for v in V:
        ivf.add(v)
    assert sum(ivf.cluster_sizes()) == 20                   # ivf_add_assigns_correct_cluster
    ivf.nprobe = 2
    ids, d = ivf.search(V[0], 5)                            # ivf_search_basic
    assert 0 < len(ids) <= 5 and all(d[i] <= d[i + 1] for i in range(len(d) - 1))
    assert ids[0] == 0 and d[0] == 0.0


def test_nprobe_effect_and_defaults():
    V = create_test_vectors(40, 16)
    ivf = fast(8)
    ivf.train(V)
    for v in V:
        ivf.add(v)
    r1, r8 = ivf.search(V[0], 5, 1), ivf.search(V[0], 5, 8)   # ivf_search_nprobe_effect
    assert r8[1][0] <= r1[1][0] + 0.001 and len(ivf.search(V[0], 5, 4)[0]) > 0
    # probing every list is the exhaustive Euclidean ranking
    full = ivf.search(V[7], 40, 8)
    exact = np.sqrt(np.array([io.sq_dist(v, V[7]) for v in V], dtype=F))
    assert sorted(full[0]) == list(range(40)) and np.array_equal(np.sort(exact), full[1])
    assert io.default_nprobe(100) == 10 and io.default_nprobe(16) == 4 and io.default_nprobe(17) == 5  # ivf.rs:46-56
    assert io.IVFFlat().nprobe == 10 and io.IVFFlat().num_clusters == 100                                # ivf.rs:71-81


def test_kmeans_is_deterministic_and_respects_k():
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.normal(0, 0.1, (50, 4)), rng.normal(5, 0.1, (50, 4))]).astype(F)
    for init in ("random", "kmeans++"):
        a = io.kmeans_fit(X, 2, io.KMeansConfig(init_method=init))
        b = io.kmeans_fit(X, 2, io.KMeansConfig(init_method=init))
        assert np.array_equal(a, b) and a.shape == (2, 4)
        means = sorted(float(c.mean()) for c in a)
        assert abs(means[0]) < 0.2 and abs(means[1] - 5) < 0.2     # two well separated blobs are found
    assert io.kmeans_fit(X[:3], 10, io.KMeansConfig()).shape == (3, 4)   # k = min(k, n), delta_vector.rs:742
    assert io.kmeans_fit(X[:0], 3, io.KMeansConfig()).shape[0] == 0


def test_rust_cast_and_fold_helpers():
    assert io.u64_to_f32(2**64 - 1) == F(2.0**64) and io.u64_to_f32(2**24 + 1) == F(2**24)   # ties to even
    assert io.u64_to_f32(2**24 + 3) == F(2**24 + 4) and io.u64_to_f32(0) == 0.0
    assert io.u64_to_f32((1 << 40) + (1 << 16)) == F(1 << 40)                                 # exact tie -> even
    assert io.u64_to_f32((1 << 40) + (1 << 16) + 1) == F((1 << 40) + (1 << 17))
    nan = float("nan")
    assert io.first_min_index([3.0, 1.0, 1.0, 2.0]) == 1          # first of equal minima
    assert io.first_min_index([nan, 1.0, 0.5]) == 0               # a NaN head is never displaced
    assert io.first_min_index([2.0, nan, 1.0]) == 2               # a NaN never displaces
    assert io.ivf_score(0.0) == 1.0 and io.ivf_score(1.0) == F(0.5)
