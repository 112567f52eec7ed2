"""Index persistence of the host-side VectorEngine mirror — the JSON form (PersistentVectorIndex as serde_json writes it,
vector_engine/src/lib.rs:500-623, 3794-3866).  Pure host logic: stores, snapshots and restores never touch the GPU, so
these run on the CPU box; they restate the reference's own tests (lib.rs:8133-8480)."""
import json
import os

import numpy as np
import pytest

from neumann_amd.engine import (DistanceMetric, VectorCollectionConfig, VectorEngine, VectorEngineConfig, VectorError)


def test_save_and_load_index_json(tmp_path):  # lib.rs:8221-8245
    path = tmp_path / "index.json"
    e = VectorEngine()
    e.store_embedding("vec1", [1.0, 2.0, 3.0])
    e.store_embedding("vec2", [4.0, 5.0, 6.0])
    e.save_index(VectorEngine.DEFAULT_COLLECTION, path)
    e2 = VectorEngine()
    assert e2.load_index(path) == VectorEngine.DEFAULT_COLLECTION
    assert e2.count() == 2
    assert list(e2.get_embedding("vec1")) == [1.0, 2.0, 3.0]
    assert list(e2.get_embedding("vec2")) == [4.0, 5.0, 6.0]


def test_file_is_the_serde_json_layout_of_persistent_vector_index(tmp_path):
    """Field names, Option::None as null, externally tagged MetadataValue, DistanceMetric by variant name, version 1
    (lib.rs:509-546, 577-596): what serde_json::to_string_pretty(&PersistentVectorIndex) produces."""
    path = tmp_path / "i.json"
    e = VectorEngine()
    e.store_embedding_with_metadata("k", [0.1, -2.5, 3.0, 1e-20], {"name": "t\"x\n", "score": 42, "ok": True, "w": 0.5, "n": None})
    e.store_embedding("plain", [1.0])
    e.save_index("default", path)
    doc = json.loads(path.read_text())
    assert list(doc) == ["collection", "config", "vectors", "created_at", "version"]
    assert doc["collection"] == "default" and doc["version"] == 1 and isinstance(doc["created_at"], int)
    assert doc["config"] == {"dimension": None, "distance_metric": "Cosine", "auto_index": False, "auto_index_threshold": 1000}
    by_key = {v["key"]: v for v in doc["vectors"]}
    assert list(by_key["k"]) == ["key", "vector", "metadata"]
    assert np.array_equal(np.array(by_key["k"]["vector"], dtype=np.float32), np.array([0.1, -2.5, 3.0, 1e-20], dtype=np.float32))
    assert by_key["k"]["metadata"] == {"name": {"String": "t\"x\n"}, "score": {"Int": 42}, "ok": {"Bool": True},
                                       "w": {"Float": 0.5}, "n": "Null"}
    assert by_key["plain"]["metadata"] is None  # `if metadata.is_empty() { None }`
    assert "0.1," in path.read_text()  # shortest round-trip decimals (ryu), not 0.100000001


def test_reads_a_file_written_by_the_reference(tmp_path):
    """A PersistentVectorIndex exactly as the Rust side pretty-prints it (hand-written here from lib.rs:509-546)."""
    path = tmp_path / "ref.json"
    path.write_text('''{
  "collection": "mycoll",
  "config": {
    "dimension": 3,
    "distance_metric": "Euclidean",
    "auto_index": true,
    "auto_index_threshold": 500
  },
  "vectors": [
    {
      "key": "vec1",
      "vector": [
        1.0,
        2.0,
        3.5
      ],
      "metadata": {
        "name": {
          "String": "caf\\u00e9"
        },
        "score": {
          "Int": -7
        },
        "nothing": "Null"
      }
    },
    {
      "key": "vec2",
      "vector": [0.25, 1e-3, -4],
      "metadata": null
    }
  ],
  "created_at": 1700000000,
  "version": 1
}''')
    e = VectorEngine()
    assert e.load_index(path) == "mycoll"
    assert e.collection_exists("mycoll")
    assert list(e.get_from_collection("mycoll", "vec1")) == [1.0, 2.0, 3.5]
    assert np.array_equal(e.get_from_collection("mycoll", "vec2"), np.array([0.25, 1e-3, -4], dtype=np.float32))
    with pytest.raises(VectorError) as err:  # the restored collection enforces its dimension (config.dimension = 3)
        e.store_in_collection("mycoll", "bad", [1.0, 2.0])
    assert err.value.kind == "DimensionMismatch"
    # the config survives a second round trip, auto_index fields included (lib.rs:8412-8440)
    out = tmp_path / "again.json"
    e.save_index("mycoll", out)
    cfg = json.loads(out.read_text())["config"]
    assert cfg == {"dimension": 3, "distance_metric": "Euclidean", "auto_index": True, "auto_index_threshold": 500}


def test_save_and_load_index_with_metadata(tmp_path):  # lib.rs:8268-8302
    path = tmp_path / "index.json"
    e = VectorEngine()
    e.store_embedding_with_metadata("vec1", [1.0, 2.0], {"name": "test", "score": 42})
    e.save_index("default", path)
    e2 = VectorEngine()
    e2.load_index(path)
    meta = e2.get_metadata("vec1")
    assert meta["name"] == "test" and meta["score"] == 42 and isinstance(meta["score"], int)


def test_save_and_load_named_collection(tmp_path):  # lib.rs:8305-8326
    path = tmp_path / "mycoll.json"
    e = VectorEngine()
    e.create_collection("mycoll", VectorCollectionConfig().with_dimension(3))
    e.store_in_collection("mycoll", "vec1", [1.0, 2.0, 3.0])
    e.save_index("mycoll", path)
    e2 = VectorEngine()
    assert e2.load_index(path) == "mycoll"
    assert e2.collection_exists("mycoll")
    assert list(e2.get_from_collection("mycoll", "vec1")) == [1.0, 2.0, 3.0]


def test_save_all_and_load_all_indices(tmp_path):  # lib.rs:8329-8393
    e = VectorEngine()
    e.store_embedding("default_vec", [1.0, 2.0])
    e.create_collection("coll_a", VectorCollectionConfig())
    e.store_in_collection("coll_a", "vec_a", [3.0, 4.0])
    e.create_collection("coll_b", VectorCollectionConfig())
    e.store_in_collection("coll_b", "vec_b", [5.0, 6.0])
    e.create_collection("empty", VectorCollectionConfig())  # lib.rs:8396-8410: empty collections are not saved
    d = tmp_path / "all" / "nested"
    saved = e.save_all_indices(d)
    assert sorted(saved) == ["coll_a", "coll_b", "default"]
    assert sorted(os.listdir(d)) == ["coll_a.json", "coll_b.json", "default.json"]
    (d / "broken.json").write_text("not valid json")  # "Log but continue with other files"
    e2 = VectorEngine()
    loaded = e2.load_all_indices(d)
    assert sorted(loaded) == ["coll_a", "coll_b", "default"]
    assert list(e2.get_embedding("default_vec")) == [1.0, 2.0]
    assert list(e2.get_from_collection("coll_a", "vec_a")) == [3.0, 4.0]


def test_errors(tmp_path):  # lib.rs:8443-8480
    e = VectorEngine()
    with pytest.raises(VectorError) as err:
        e.load_index("/nonexistent/path/index.json")
    assert err.value.kind == "IoError" and str(err.value).startswith("IO error: ")
    bad = tmp_path / "invalid.json"
    bad.write_text("not valid json")
    with pytest.raises(VectorError) as err:
        e.load_index(bad)
    assert err.value.kind == "SerializationError" and str(err.value).startswith("Serialization error: ")
    binbad = tmp_path / "invalid.bin"
    binbad.write_bytes(bytes([0xFF, 0xFF, 0xFF]))
    with pytest.raises(VectorError) as err:
        e.load_index_binary(binbad)
    assert err.value.kind == "SerializationError"
    for frag in ('{"collection": "x"}', '{"collection":"x","config":{"dimension":null,"distance_metric":"Manhattan","auto_index":false,'
                 '"auto_index_threshold":1},"vectors":[],"created_at":0,"version":1}'):
        bad.write_text(frag)
        with pytest.raises(VectorError) as err:
            e.load_index(bad)
        assert err.value.kind == "SerializationError"


def test_limits(tmp_path):  # lib.rs:3831-3856, 6222-6245
    path = tmp_path / "i.json"
    e = VectorEngine()
    for i in range(20):
        e.store_embedding(f"k{i}", [float(i), 1.0, 2.0])
    e.save_index("default", path)
    size = os.path.getsize(path)
    small = VectorEngine(VectorEngineConfig(max_index_file_bytes=size - 1))
    with pytest.raises(VectorError) as err:
        small.load_index(path)
    assert err.value.kind == "ConfigurationError"
    assert str(err.value) == f"Configuration error: index file size {size} exceeds limit {size - 1}"
    few = VectorEngine(VectorEngineConfig(max_index_entries=19))
    with pytest.raises(VectorError) as err:
        few.load_index(path)
    assert str(err.value) == "Configuration error: index entry count 20 exceeds limit 19"
    assert few.count() == 0  # nothing was restored
    exact = VectorEngine(VectorEngineConfig(max_index_file_bytes=size, max_index_entries=20))
    exact.load_index(path)
    assert exact.count() == 20
    unlimited = VectorEngine(VectorEngineConfig(max_index_file_bytes=None, max_index_entries=None))
    unlimited.load_index(path)
    for field in ("max_index_file_bytes", "max_index_entries"):
        with pytest.raises(VectorError) as err:
            VectorEngine(VectorEngineConfig(**{field: 0}))
        assert err.value.kind == "ConfigurationError" and field in str(err.value)


def test_load_overwrites_existing_keys_and_float_round_trip(tmp_path):
    """`If the collection already exists with vectors, they will be overwritten`; every f32 survives the decimal form."""
    path = tmp_path / "i.json"
    rng = np.random.default_rng(9)
    v = rng.standard_normal(300).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 300).astype(np.float32)
    v[:4] = [np.float32(1e-45), np.float32(3.4028235e38), np.float32(-0.0), np.float32(16777217.0)]
    e = VectorEngine()
    e.store_embedding("a", v)
    e.save_index("default", path)
    e2 = VectorEngine()
    e2.store_embedding("a", np.zeros(300, np.float32))
    e2.store_embedding("other", [1.0])
    e2.load_index(path)
    assert e2.count() == 2
    assert np.array_equal(e2.get_embedding("a").view(np.uint32), v.view(np.uint32))


def test_engine_rejects_more_devices_than_it_can_hold():
    with pytest.raises(VectorError) as ei:
        VectorEngine(VectorEngineConfig(devices=(0,) * 17))
    assert ei.value.kind == "ConfigurationError"
