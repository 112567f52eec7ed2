"""IVF-Flat probe on the GPU against oracle/ivf_oracle.py, the restatement of
tensor_store/src/ivf.rs: identical cluster assignments, identical ids in identical order, bit-equal
distances; plus the reference's own IVF property tests (ivf.rs:656-770) on the GPU index."""
import numpy as np
import pytest

from oracle import ivf_oracle as io

pytestmark = pytest.mark.gpu
F = np.float32


def create_test_vectors(n, dim):  # ivf.rs:569-577
    return np.array([[((i * 7 + j * 13) % 100) / 100.0 for j in range(dim)] for i in range(n)], dtype=F)


FAST = dict(max_iterations=2, convergence_threshold=1.0, seed=42, init_method="random")  # ivf.rs:589-596


def build_pair(vectors, num_clusters, nprobe=None, kmeans=None, train=None, spare=64):
    """Oracle index trained on `train` (default: the vectors) and a GPU index created from ITS centroids."""
    from neumann_amd.ivf import GpuIvfFlat
    orc = io.IVFFlat(num_clusters, nprobe=nprobe, kmeans=io.KMeansConfig(**(kmeans or FAST)))
    orc.train(vectors if train is None else train)
    gpu = GpuIvfFlat(orc.centroids, capacity_rows=len(vectors) + spare, nprobe=orc.nprobe)
    return orc, gpu


def check_same(orc, gpu, q, k, nprobe=None):
    ids, dist, counts = gpu.search(q, k, nprobe)
    eids, ed = orc.search(q, k, nprobe)
    assert counts[0] == len(eids)
    assert ids[0, :len(eids)].tolist() == eids
    assert np.array_equal(dist[0, :len(eids)], ed)
    assert np.all(ids[0, len(eids):] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isposinf(dist[0, len(eids):]))


def test_reference_property_tests():  # ivf.rs:656-770
    V = create_test_vectors(20, 16)
    orc, gpu = build_pair(V, 4, nprobe=2)
    with gpu:
        clusters = gpu.add(V)
        for v in V:
            orc.add(v)
        assert len(gpu) == 20 and int(gpu.cluster_sizes().sum()) == 20           # ivf_add_assigns_correct_cluster
        assert clusters.tolist() == orc.assign and gpu.cluster_sizes().tolist() == orc.cluster_sizes()
        ids, dist, counts = gpu.search(V[0], 5)                                     # ivf_search_basic
        assert 0 < counts[0] <= 5 and np.all(np.diff(dist[0, :counts[0]]) >= 0)
        assert ids[0, 0] == 0 and dist[0, 0] == 0.0
        check_same(orc, gpu, V[0], 5)
    V = create_test_vectors(40, 16)                                                 # ivf_search_nprobe_effect
    orc, gpu = build_pair(V, 8)
    with gpu:
        gpu.add(V)
        for v in V:
            orc.add(v)
        r1 = gpu.search(V[0], 5, 1)
        r8 = gpu.search(V[0], 5, 8)
        assert r8[1][0, 0] <= r1[1][0, 0] + 0.001 and gpu.search(V[0], 5, 4)[2][0] > 0
        for nprobe in (1, 2, 4, 8, 100):
            check_same(orc, gpu, V[3], 5, nprobe)


@pytest.mark.parametrize("n,d,c,nprobe", [(3000, 32, 16, None), (5000, 96, 50, 7), (2000, 17, 9, 3)])
def test_ivf_matches_oracle_on_random_data(n, d, c, nprobe):
    rng = np.random.default_rng(n + d)
    V = rng.standard_normal((n, d)).astype(F)
    V[11] = V[5]                      # duplicates: same list, id order
    V[n - 1] = V[5]
    orc, gpu = build_pair(V, c, nprobe=nprobe, kmeans=dict(max_iterations=5, convergence_threshold=1e-4, seed=7,
                                                           init_method="kmeans++"), train=V[:600])
    with gpu:
        # added in several calls: ids keep counting, assignments do not depend on the batching
        got = np.concatenate([gpu.add(V[:1000]), gpu.add(V[1000:1001]), gpu.add(V[1001:])])
        for v in V:
            orc.add(v)
        assert got.tolist() == orc.assign
        assert gpu.cluster_sizes().tolist() == orc.cluster_sizes()
        Q = rng.standard_normal((4, d)).astype(F)
        for q in list(Q) + [V[5], V[123]]:
            for k in (1, 10, 200):
                check_same(orc, gpu, q, k)
            check_same(orc, gpu, q, 50, nprobe=1)
            check_same(orc, gpu, q, 5000, nprobe=c)       # every list probed, k beyond the candidate pipeline
        # multi-query call = the single-query calls
        ids, dist, counts = gpu.search(Q, 10)
        for i in range(4):
            eids, ed = orc.search(Q[i], 10)
            assert ids[i].tolist() == eids and np.array_equal(dist[i], ed)


def test_ivf_equal_distances_keep_probe_order():
    """Vectors at exactly the same distance in DIFFERENT lists come back in probe order of their lists
    (stable sort of the candidate list, ivf.rs:402), not in id order."""
    from neumann_amd.ivf import GpuIvfFlat
    cents = np.array([[10.0, 0.0], [-10.0, 0.0], [0.0, 10.0]], dtype=F)
    orc = io.IVFFlat(3, nprobe=3)
    orc.centroids = cents
    orc.lists = [[], [], []]
    V = np.array([[-9.0, 0.0],    # id 0, list 1, distance 9 from the query
                  [9.0, 0.0],     # id 1, list 0, distance 9
                  [0.0, 9.0],     # id 2, list 2, distance 9
                  [9.0, 0.0],     # id 3, list 0, distance 9 (duplicate of id 1)
                  [5.0, 0.0]], dtype=F)
    q = np.array([0.0, 0.0], dtype=F) + np.array([1e-3, 0.0], dtype=F) * 0  # origin
    q = np.array([0.5, 0.0], dtype=F)   # nearest centroid order: 0 (9.5^2), 2, 1
    V[0] = [-8.0, 0.0]                  # distance 8.5
    V[1] = [9.0, 0.0]                   # distance 8.5
    V[2] = [0.5, 8.5]                   # distance 8.5
    V[3] = [9.0, 0.0]                   # distance 8.5
    with GpuIvfFlat(cents, capacity_rows=16, nprobe=3) as gpu:
        gpu.add(V)
        for v in V:
            orc.add(v)
        eids, ed = orc.search(q, 5)
        assert eids == [4, 1, 3, 2, 0]          # list 0 first (ids 1, 3), then list 2, then list 1
        check_same(orc, gpu, q, 5)
        check_same(orc, gpu, q, 3)


# ---- engine level: build_ivf_index / search_with_ivf (vector_engine/src/lib.rs:2641-2812) ------------------
@pytest.fixture
def E():
    from neumann_amd import engine
    return engine


@pytest.mark.parametrize("init", ["random", "kmeans++"])
def test_engine_build_ivf_index_matches_oracle(E, init):
    rng = np.random.default_rng(17)
    n, d, c = 1500, 24, 12
    V = rng.standard_normal((n, d)).astype(F)
    engine = E.VectorEngine()
    engine.batch_store_embeddings([f"k{i}" for i in range(n)], V)
    opts = E.IVFBuildOptions(num_clusters=c, nprobe=4, max_iterations=6, convergence_threshold=1e-4, seed=99,
                             init_method=init)
    index, keys = engine.build_ivf_index(opts)
    assert index.is_trained() and len(index) == n and index.num_clusters == c and index.nprobe == 4
    order = [int(k[1:]) for k in keys]                      # the engine's list_keys() order feeds the training
    orc = io.IVFFlat(c, nprobe=4, kmeans=io.KMeansConfig(6, 1e-4, 99, init))
    orc.train(V[order])
    for i in order:
        orc.add(V[i])
    assert np.array_equal(index.centroids(d), orc.centroids)          # k-means restated bit for bit
    assert index.cluster_sizes().tolist() == orc.cluster_sizes()
    for t in range(5):
        q = rng.standard_normal(d).astype(F)
        for nprobe in (None, 1, c):
            res = (engine.search_with_ivf(index, keys, q, 10) if nprobe is None
                   else engine.search_with_ivf_nprobe(index, keys, q, 10, nprobe))
            eids, ed = orc.search(q, 10, nprobe)
            assert [r.key for r in res] == [keys[i] for i in eids]
            assert np.array_equal(np.array([r.score for r in res], F), np.array([io.ivf_score(x) for x in ed], F))


def test_engine_ivf_edge_cases(E):  # lib.rs:2643-2647, 2717-2722; ivf.rs:326-328
    engine = E.VectorEngine()
    index, keys = engine.build_ivf_index_default()
    assert keys == [] and not index.is_trained()
    assert engine.search_with_ivf(index, keys, [1.0, 0.0], 5) == []
    for i in range(30):
        engine.store_embedding(f"k{i}", [float(i), 1.0, 0.5])
    index, keys = engine.build_ivf_index(E.IVFBuildOptions.flat(100))   # more clusters than vectors: min(k, n)
    assert index.num_clusters == 30 and len(index) == 30 and index.nprobe == 10
    res = engine.search_with_ivf_nprobe(index, keys, [3.0, 1.0, 0.5], 3, 30)
    assert res[0].key == "k3" and res[0].score == 1.0
    with pytest.raises(E.VectorError) as e:
        engine.search_with_ivf(index, keys, [], 5)
    assert e.value.kind == "EmptyVector"
    with pytest.raises(E.VectorError) as e:
        engine.search_with_ivf(index, keys, [1.0, 2.0, 3.0], 0)
    assert e.value.kind == "InvalidTopK"
    engine.store_embedding("odd", [1.0, 2.0])
    with pytest.raises(E.VectorError) as e:
        engine.build_ivf_index_default()
    assert e.value.kind == "DimensionMismatch"


@pytest.mark.parametrize("init", ["random", "kmeans++"])
def test_gpu_kmeans_training_matches_oracle(init):
    """nmn_ivf_build: k-means on the GPU (exact centroid sweeps + sequential per-(cluster, dimension) sums) gives the
    oracle's centroids bit for bit, hence the same lists and the same search results."""
    from neumann_amd.ivf import GpuIvfFlat
    rng = np.random.default_rng(3)
    n, d, c = 4000, 40, 24
    V = (rng.standard_normal((n, d)) + 2.5 * rng.integers(0, 4, (n, 1))).astype(F)
    V[100] = V[7]
    cfg = dict(max_iterations=12, convergence_threshold=1e-4, seed=2024, init_method=init)
    orc = io.IVFFlat(c, nprobe=5, kmeans=io.KMeansConfig(**cfg))
    orc.train(V)
    for v in V:
        orc.add(v)
    with GpuIvfFlat.build(V, c, nprobe=5, **cfg) as gpu:
        assert np.array_equal(gpu.centroids(), orc.centroids)
        assert len(gpu) == n and gpu.cluster_sizes().tolist() == orc.cluster_sizes()
        for q in rng.standard_normal((5, d)).astype(F) + F(2.5):
            check_same(orc, gpu, q, 20)
            check_same(orc, gpu, q, 20, nprobe=c)
    # vectors added after training go to the nearest trained centroid (capacity above was exactly n: head room here)
    extra = rng.standard_normal((50, d)).astype(F)
    with GpuIvfFlat.build(V, c, nprobe=5, capacity_rows=n + 64, **cfg) as gpu:
        got = gpu.add(extra)
        for v in extra:
            orc.add(v)
        assert got.tolist() == orc.assign[n:]
        check_same(orc, gpu, extra[3], 10)
    # more clusters than vectors, a single vector, zero iterations
    with GpuIvfFlat.build(V[:5], 100, **cfg) as gpu:
        assert gpu.n_clusters == 5 and gpu.cluster_sizes().tolist() == [1, 1, 1, 1, 1]
    o0 = io.IVFFlat(3, kmeans=io.KMeansConfig(0, 1e-4, 9, init))
    o0.train(V[:50])
    with GpuIvfFlat.build(V[:50], 3, max_iterations=0, seed=9, init_method=init) as gpu:
        assert np.array_equal(gpu.centroids(), o0.centroids)


@pytest.mark.parametrize("d", [64, 128, 768])   # VALU list scans only / mixed-filter matrix-core sweeps from 5 / 3 probes
def test_concurrent_probes_equal_sequential_probes(d):
    """Searches of one IVF index from many threads (each with a probe slot of its own; the list scans go through the
    flat index's coalescer) return exactly what the same searches return one at a time."""
    import threading
    from neumann_amd.ivf import GpuIvfFlat
    rng = np.random.default_rng(31)
    n, nlist, k = (60_000 if d < 768 else 24_000), 48, 12
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[1000:1040] = X[999]                                  # ties across and inside lists
    ivf = GpuIvfFlat.build(X[:12_000], nlist, nprobe=6, max_iterations=3, seed=7, init_method="random", capacity_rows=n)
    with ivf:
        ivf.add(X[12_000:])
        Q = rng.standard_normal((64, d)).astype(np.float32)
        Q[5] = X[999]
        want = [ivf.search(q, k) for q in Q]
        got = [None] * 64
        errs = []
        start = threading.Barrier(16)

        def work(t):
            try:
                start.wait()
                for rep in range(3):
                    for j in range(t, 64, 16):
                        got[j] = ivf.search(Q[j], k, nprobe=6 if j % 2 else 9)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        want = [ivf.search(Q[j], k, nprobe=6 if j % 2 else 9) for j in range(64)]
        th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for j in range(64):
            for a, b in zip(want[j], got[j]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), j


@pytest.mark.parametrize("n,d,c,nprobe,young", [(9000, 48, 24, 5, 0), (12_000, 128, 40, 7, 3000)])
def test_many_queries_per_call_share_the_centroid_phase(n, d, c, nprobe, young):
    """One call with many queries (chunks of 16 share the centroid sweep, the ranking launch, the bitmap launches and one
    round trip) answers exactly what one call per query answers and what the oracle answers — across a ragged last chunk,
    growth of a probe slot that first served a single query, ties, and rows younger than the list-major copy."""
    rng = np.random.default_rng(77)
    X = (rng.standard_normal((n, d)) + 3.0 * rng.standard_normal((c, d))[rng.integers(0, c, n)]).astype(F)
    X[500:520] = X[499]
    orc, gpu = build_pair(X[: n - young], c, nprobe=nprobe, spare=young + 64)
    with gpu:
        gpu.add(X[: n - young])
        for v in X[: n - young]:
            orc.add(v)
        if young:                                        # list-major copy covers the first part only
            gpu.add(X[n - young:])
            for v in X[n - young:]:
                orc.add(v)
        Q = rng.standard_normal((41, d)).astype(F) + X[rng.integers(0, n, 41)]
        Q[3] = X[499]
        one = gpu.search(Q[0], 9)                        # a slot sized for one query exists before the batched call
        for k, npb in ((9, None), (1, 2), (300, c)):
            ids, dist, counts = gpu.search(Q, k, npb)
            assert ids.shape == (41, k)
            for j in range(41):
                a = gpu.search(Q[j], k, npb)
                assert np.array_equal(a[0][0], ids[j]) and np.array_equal(a[1][0], dist[j]) and a[2][0] == counts[j], (k, j)
                eids, ed = orc.search(Q[j], k, npb)
                assert counts[j] == len(eids) and ids[j, :len(eids)].tolist() == eids, (k, j)
                assert np.array_equal(dist[j, :len(eids)], ed)
        assert np.array_equal(one[0][0], gpu.search(Q[:17], 9)[0][0])
        # two queries: below what one pass over the vectors is worth — the list scans run one by one; 41: as one batch per part
        two = gpu.search(Q[:2], 9)
        assert np.array_equal(two[0], gpu.search(Q, 9)[0][:2]) and np.array_equal(two[1], gpu.search(Q, 9)[1][:2])


def test_concurrent_multi_query_calls_each_lead_their_own_batch():
    """Several threads, each calling search with 24 queries: every call hands its list scans to the flat index as a batch of its
    own.  While the shard is busy such a request waits in the coalescer's queue — where it must not be picked up as a rider of
    somebody else's batch (its own riders would never be served) — and answers must equal the sequential ones."""
    import threading
    from neumann_amd.ivf import GpuIvfFlat
    rng = np.random.default_rng(5)
    n, d, nlist, k = 40_000, 128, 32, 10
    X = (rng.standard_normal((n, d)) + 2.0 * rng.standard_normal((nlist, d))[rng.integers(0, nlist, n)]).astype(np.float32)
    ivf = GpuIvfFlat.build(X[:8000], nlist, nprobe=4, max_iterations=3, seed=3, init_method="random", capacity_rows=n)
    with ivf:
        ivf.add(X[8000:])
        Q = rng.standard_normal((8 * 24, d)).astype(np.float32) + X[rng.integers(0, n, 8 * 24)]
        want = [ivf.search(Q[t * 24:(t + 1) * 24], k) for t in range(8)]
        singles = [ivf.search(Q[j], k) for j in range(0, 8 * 24, 7)]
        got, errs = [None] * 8, []
        got1 = [None] * len(singles)
        start = threading.Barrier(10)

        def many(t):
            try:
                start.wait()
                for _ in range(4):
                    got[t] = ivf.search(Q[t * 24:(t + 1) * 24], k)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def single(off):
            try:
                start.wait()
                for _ in range(3):
                    for i, j in enumerate(range(0, 8 * 24, 7)):
                        if i % 2 == off:
                            got1[i] = ivf.search(Q[j], k)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=many, args=(t,)) for t in range(8)] + [threading.Thread(target=single, args=(o,)) for o in (0, 1)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=120)
        assert not any(x.is_alive() for x in th), "a search never came back"
        assert not errs, errs
        for t in range(8):
            for a, b in zip(want[t], got[t]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), t
        for i in range(len(singles)):
            for a, b in zip(singles[i], got1[i]):
                assert np.array_equal(np.asarray(a), np.asarray(b)), i


# ---- the list-major copy (round 3) ---------------------------------------------------------------------------------------
def test_list_major_copy_matches_oracle_through_adds_and_relayouts():
    """>= 4096 vectors: probes read a second copy of the vectors ordered by list (contiguous ranges) and map its rows back to
    ids; vectors added since it was laid out are scanned through the bitmap and merged; a new layout happens when they make
    up an eighth.  Ids, order (probe order of the list, then id, for equal distances) and distances must be the oracle's at
    every stage — duplicates inside and across the two parts included."""
    rng = np.random.default_rng(91)
    n0, d, c = 6000, 64, 24
    centres = rng.standard_normal((c, d)).astype(F) * F(2.0)
    def rows(m):
        return (centres[rng.integers(0, c, m)] + F(0.4) * rng.standard_normal((m, d)).astype(F)).astype(F)
    V = rows(n0)
    V[77] = V[5]
    V[4000] = V[5]
    orc, gpu = build_pair(V, c, nprobe=5, kmeans=dict(max_iterations=4, convergence_threshold=1e-4, seed=3, init_method="kmeans++"),
                          train=V[:1500], spare=5000)
    with gpu:
        assert gpu.list_major_rows == 0
        gpu.add(V)
        for v in V:
            orc.add(v)
        assert gpu.list_major_rows == n0                      # laid out by the add that crossed 4096 vectors
        Q = [V[5], V[123], rows(1)[0], centres[3]]
        for q in Q:
            for nprobe in (1, 5, c):
                check_same(orc, gpu, q, 25, nprobe)
        young = rows(300)
        young[7] = V[5]                                        # a copy of an old vector among the young ones: a tie across the parts
        gpu.add(young)
        for v in young:
            orc.add(v)
        assert gpu.list_major_rows == n0 and len(gpu) == n0 + 300   # not yet an eighth: two-part probes
        for q in Q + [young[7], young[100]]:
            for nprobe in (1, 5, c):
                check_same(orc, gpu, q, 25, nprobe)
            check_same(orc, gpu, q, 700, 5)                    # k beyond most lists
        more = rows(3900)
        gpu.add(more)
        for v in more:
            orc.add(v)
        assert gpu.list_major_rows == n0 + 4200                # 4200 young vectors (>= 4096 and >= an eighth of 6000): laid out afresh
        for q in Q + [more[3]]:
            check_same(orc, gpu, q, 25, 5)


def test_list_major_copy_after_build_and_after_load(tmp_path):
    from neumann_amd.ivf import GpuIvfFlat
    rng = np.random.default_rng(92)
    n, d, c = 9000, 128, 32
    V = (rng.standard_normal((c, d))[rng.integers(0, c, n)] * 2.0 + 0.5 * rng.standard_normal((n, d))).astype(F)
    Q = V[:3] + F(0.01)
    with GpuIvfFlat.build(V, c, max_iterations=5, seed=7) as ivf:
        assert ivf.list_major_rows == n
        want = [ivf.search(Q, 30, nprobe=p) for p in (1, 6, c)]
        ex = ivf.search(Q, 30, nprobe=c)
        path = tmp_path / "i.nmnidx"
        ivf.save(path)
    with GpuIvfFlat.load(path) as ivf:
        assert ivf.list_major_rows == n
        for p, (ids, dist, cnt) in zip((1, 6, c), want):
            i2, d2, c2 = ivf.search(Q, 30, nprobe=p)
            assert np.array_equal(i2, ids) and np.array_equal(d2.view(np.uint32), dist.view(np.uint32)) and np.array_equal(c2, cnt)
    # probing every list is the exhaustive Euclidean search: same ids as the flat index over the same rows
    from neumann_amd import GpuFlatIndex
    with GpuFlatIndex(d, n, single_launch=False) as flat:
        flat.upload(V)
        rows, scores, counts = flat.search(Q, 30, 1)
        assert np.array_equal(rows, ex[0])
