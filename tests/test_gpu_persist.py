"""Persistence of the DEVICE LAYOUT (second half): nmn_index_save/load, nmn_ivf_save/load and the engine's
save_index_binary / load_index_binary / IVF index files.  save -> destroy -> load -> answers identical to the oracle;
limits as VectorEngineConfig::max_index_file_bytes / max_index_entries (vector_engine/src/lib.rs:644-646, 3831-3856)."""
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu


def _same(idx, A, Q, k, metrics=(0, 1, 2), row_base=0):
    for m in metrics:
        rows, scores, counts = idx.search(Q, k, m)
        for qi in range(Q.shape[0]):
            er, es = oc.search(A, Q[qi], k, m, row_base=row_base)
            assert counts[qi] == er.size and np.array_equal(rows[qi, :er.size], er) and np.all(scores[qi, :er.size] == es)


@pytest.mark.parametrize("n,d", [(5000, 768), (3001, 100), (64, 7)])
def test_index_save_destroy_load_matches_oracle(tmp_path, n, d):
    from neumann_amd import GpuFlatIndex
    path = tmp_path / "shard.nmnidx"
    rng = np.random.default_rng(n + d)
    A = rng.standard_normal((n, d)).astype(np.float32)
    A[7] = 0.0  # a zero row: magnitude 0, scores 0.0
    Q = rng.standard_normal((3, d)).astype(np.float32)
    with GpuFlatIndex(d, n, row_base=1000) as idx:
        idx.upload(A)
        idx.search(Q[0], 5, 0)  # the bf16 mirror exists at save time: it is not part of the file
        idx.save(path)
    assert os.path.getsize(path) == 64 + n * d * 4 + n * 4  # header | rows, stride removed | magnitudes
    with GpuFlatIndex.load(path, capacity_rows=n + 100) as idx:
        assert (idx.rows, idx.dim, idx.row_base) == (n, d, 1000)
        _same(idx, A, Q, 50, row_base=1000)
        extra = rng.standard_normal((100, d)).astype(np.float32)  # spare capacity: appends keep working
        idx.upload(extra)
        _same(idx, np.concatenate([A, extra]), Q, 50, metrics=(0,), row_base=1000)


def test_index_load_limits_and_corruption(tmp_path):
    from neumann_amd import GpuFlatIndex, NeumannGpuError, _capi
    path = tmp_path / "s.nmnidx"
    A = np.random.default_rng(1).standard_normal((200, 32)).astype(np.float32)
    with GpuFlatIndex(32, 200) as idx:
        idx.upload(A)
        idx.save(path)
    size = os.path.getsize(path)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(path, max_file_bytes=size - 1)
    assert e.value.status == _capi.ERR_CONFIGURATION and f"index file size {size} exceeds limit {size - 1}" in str(e.value)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(path, max_entries=199)
    assert e.value.status == _capi.ERR_CONFIGURATION and "index entry count 200 exceeds limit 199" in str(e.value)
    GpuFlatIndex.load(path, max_file_bytes=size, max_entries=200).close()
    raw = bytearray(path.read_bytes())
    flipped = bytearray(raw)
    flipped[64 + 4 * 1234] ^= 0x10  # one bit of one element: its row's magnitude no longer matches the stored one
    (tmp_path / "bitflip").write_bytes(flipped)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "bitflip")
    assert e.value.status == _capi.ERR_SERIALIZATION and "corrupt" in str(e.value)
    (tmp_path / "short").write_bytes(raw[:size // 2])
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "short")
    assert e.value.status == _capi.ERR_SERIALIZATION
    # a header that announces far more than the file holds (rows x dim near 2^64) must not size an allocation
    hostile = bytearray(raw)
    hostile[24:32] = (2**61).to_bytes(8, "little")       # PersistHeader.rows
    (tmp_path / "hostile").write_bytes(hostile)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "hostile")
    assert e.value.status == _capi.ERR_SERIALIZATION
    hostile[24:32] = (10**9).to_bytes(8, "little")
    hostile[40:48] = (10**9 * 33 * 4).to_bytes(8, "little")   # ... and a payload_bytes to match
    (tmp_path / "hostile2").write_bytes(hostile)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "hostile2")
    assert e.value.status == _capi.ERR_SERIALIZATION
    (tmp_path / "junk").write_bytes(b"\xff" * 100)
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "junk")
    assert e.value.status == _capi.ERR_SERIALIZATION
    with pytest.raises(NeumannGpuError) as e:
        GpuFlatIndex.load(tmp_path / "absent")
    assert e.value.status == _capi.ERR_IO


def test_ivf_save_load_restores_lists_without_retraining(tmp_path):
    from neumann_amd import _capi
    from neumann_amd.ivf import GpuIvfFlat
    path = tmp_path / "ivf.nmnidx"
    rng = np.random.default_rng(3)
    n, d, C = 6000, 64, 24
    V = (rng.standard_normal((n, d)) + 3.0 * rng.standard_normal((C, d))[rng.integers(0, C, n)]).astype(np.float32)
    Q = V[:4] + np.float32(0.01)
    with GpuIvfFlat.build(V, C, max_iterations=8, seed=7) as ivf:
        cents, sizes = ivf.centroids(), ivf.cluster_sizes()
        want = [ivf.search(Q, 20, nprobe=p) for p in (1, 5, C)]
        ivf.save(path)
    with GpuIvfFlat.load(path, capacity_rows=n + 10) as ivf:
        assert len(ivf) == n and ivf.n_clusters == C
        assert np.array_equal(ivf.centroids().view(np.uint32), cents.view(np.uint32))
        assert np.array_equal(ivf.cluster_sizes(), sizes)
        for p, (ids, dist, cnt) in zip((1, 5, C), want):
            i2, d2, c2 = ivf.search(Q, 20, nprobe=p)
            assert np.array_equal(i2, ids) and np.array_equal(d2.view(np.uint32), dist.view(np.uint32)) and np.array_equal(c2, cnt)
        assert ivf.add(V[:3] * np.float32(1.5)).size == 3  # the restored index accepts new vectors
    # a crafted file: the embedded vector section announces no payload (payload_bytes = 0 passes the "is it in the file"
    # check with any row count) — refused as inconsistent before anything is allocated by its shape
    raw = bytearray(path.read_bytes())
    second = 64 + C * d * 4 + n * 4          # ivf header | centroids | lists | flat-section header
    assert raw[second:second + 6] == b"NMNIDX"
    raw[second + 40:second + 48] = (0).to_bytes(8, "little")   # PersistHeader::payload_bytes
    bad = tmp_path / "crafted.nmnidx"
    bad.write_bytes(bytes(raw))
    with pytest.raises(_capi.NeumannGpuError) as e:
        GpuIvfFlat.load(bad)
    assert e.value.status == _capi.ERR_SERIALIZATION


def test_engine_binary_round_trip_rebuilds_the_mirror_at_once(tmp_path):
    """save_index_binary / load_index_binary (lib.rs:3811-3817, 3868-3899; tests 8248-8265): keys, metadata, config and the
    matrix per dimension; the load builds the GPU mirror itself and checks every row's magnitude against the file."""
    from neumann_amd.engine import FilterCondition, FilteredSearchConfig, VectorCollectionConfig, VectorEngine
    path = tmp_path / "coll.bin"
    rng = np.random.default_rng(11)
    e = VectorEngine()
    e.create_collection("docs", VectorCollectionConfig().with_dimension(48))
    vecs = {f"d{i}": rng.standard_normal(48).astype(np.float32) for i in range(700)}
    for i, (k, v) in enumerate(vecs.items()):
        e.store_in_collection_with_metadata("docs", k, v, {"bucket": i % 7, "tag": f"t{i % 3}"})
    e.delete_from_collection("docs", "d13")
    vecs.pop("d13")
    q = rng.standard_normal(48).astype(np.float32)
    want = e.search_in_collection("docs", q, 25)
    e.save_index_binary("docs", path)
    e2 = VectorEngine()
    assert e2.load_index_binary(path) == "docs"
    assert e2.mirror_builds() == 1 and e2.mirror_cached("docs")  # built by the load, not by the first search
    got = e2.search_in_collection("docs", q, 25)
    assert e2.mirror_builds() == 1
    assert [(r.key, r.score) for r in got] == [(r.key, r.score) for r in want]
    assert e2.collection_count("docs") == 699 and not e2.exists_in_collection("docs", "d13")
    f = FilterCondition.Eq("bucket", 3).and_(FilterCondition.Eq("tag", "t0"))
    a = e.search_filtered_in_collection("docs", q, 10, f, FilteredSearchConfig.pre_filter())
    b = e2.search_filtered_in_collection("docs", q, 10, f, FilteredSearchConfig.pre_filter())
    assert [(r.key, r.score) for r in a] == [(r.key, r.score) for r in b] and len(a) > 0
    # default collection with mixed dimensions: one matrix section per dimension
    e.store_embedding("x3", [1.0, 2.0, 3.0])
    e.store_embedding("y3", [3.0, 2.0, 1.0])
    e.store_embedding("z5", [1.0, 0.0, 0.0, 0.0, 2.0])
    p2 = tmp_path / "default.bin"
    e.save_index_binary("default", p2)
    e3 = VectorEngine()
    assert e3.load_index_binary(p2) == "default"
    assert [r.key for r in e3.search_similar([1.0, 2.0, 3.0], 5)] == ["x3", "y3"]
    assert list(e3.get_embedding("z5")) == [1.0, 0.0, 0.0, 0.0, 2.0]
    # a flipped bit in a matrix section is caught by the magnitude check
    raw = bytearray(path.read_bytes())
    raw[len(raw) - 699 * 4 - 200] ^= 0x04
    (tmp_path / "flip.bin").write_bytes(raw)
    from neumann_amd.engine import VectorError
    with pytest.raises(VectorError) as err:
        VectorEngine().load_index_binary(tmp_path / "flip.bin")
    assert err.value.kind == "SerializationError"


def test_engine_ivf_index_file(tmp_path):
    from neumann_amd.engine import IVFBuildOptions, VectorEngine
    path = tmp_path / "ivf.bin"
    rng = np.random.default_rng(5)
    e = VectorEngine()
    for i in range(900):
        e.store_embedding(f"k{i}", rng.standard_normal(32).astype(np.float32))
    index, keys = e.build_ivf_index(IVFBuildOptions(num_clusters=12, max_iterations=5))
    q = rng.standard_normal(32).astype(np.float32)
    want = e.search_with_ivf(index, keys, q, 15)
    e.save_ivf_index(index, path)
    e2 = VectorEngine()
    index2, keys2 = e2.load_ivf_index(path)
    assert keys2 == keys and index2.num_clusters == 12 and index2.nprobe == index.nprobe
    assert np.array_equal(index2.centroids(32).view(np.uint32), index.centroids(32).view(np.uint32))
    got = e2.search_with_ivf(index2, keys2, q, 15)
    assert [(r.key, r.score) for r in got] == [(r.key, r.score) for r in want]
