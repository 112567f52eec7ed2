"""`VectorEngineConfig::max_keys_per_scan` on the host-side VectorEngine mirror: every unbounded walk over the store stops
after that many keys (vector_engine/src/lib.rs:2321-2329 list_keys_bounded, 2340-2354 clear, 2945-2980 list_keys_paginated,
3224-3237 scan / count of entities; Some(0) is a ConfigurationError, lib.rs:728-733).  Pure host logic: runs on the CPU box.
The reference's own tests for these: lib.rs:6193-6207, 6279-6309, 9524-9541."""
import pytest

from neumann_amd.engine import Pagination, VectorEngine, VectorEngineConfig, VectorError


def _filled(n, **cfg):
    e = VectorEngine(VectorEngineConfig(**cfg)) if cfg else VectorEngine()
    for i in range(n):
        e.store_embedding(f"v{i}", [float(i)])
    return e


def test_config_validate_invalid_max_keys_per_scan_zero():  # lib.rs:6193-6207
    with pytest.raises(VectorError) as ex:
        VectorEngine(VectorEngineConfig(max_keys_per_scan=0))
    assert ex.value.kind == "ConfigurationError" and "max_keys_per_scan" in str(ex.value)
    VectorEngine(VectorEngineConfig(max_dimension=1024, max_keys_per_scan=1000))  # with_config_valid_succeeds, lib.rs:6268-6277


def test_list_keys_bounded_respects_limit():  # lib.rs:6279-6295
    e = _filled(10, max_keys_per_scan=3)
    keys = e.list_keys_bounded()
    assert len(keys) == 3 and len(set(keys)) == 3 and set(keys) <= {f"v{i}" for i in range(10)}
    assert len(e.list_keys()) == 3  # list_keys() IS list_keys_bounded() (lib.rs:2312-2314)
    assert e.count() == 10          # count() is not a scan of keys


def test_list_keys_bounded_no_limit_and_loose_limit():  # lib.rs:6297-6309, 9524-9541
    assert len(_filled(10).list_keys_bounded()) == 10
    assert len(_filled(10, max_keys_per_scan=100).list_keys_bounded()) == 10


def test_clear_is_bounded_and_converges():  # lib.rs:2331-2354: "call again until 0 is returned"
    e = _filled(10, max_keys_per_scan=4)
    assert e.clear() == 4 and e.count() == 6
    assert e.clear() == 4 and e.count() == 2
    assert e.clear() == 2 and e.count() == 0
    assert e.clear() == 0
    e2 = _filled(5)
    assert e2.clear() == 5 and e2.count() == 0  # clear_all_embeddings, lib.rs:9545-9560


def test_list_keys_paginated():  # lib.rs:2945-2980
    e = _filled(10)
    p = e.list_keys_paginated(Pagination(skip=2, limit=3, count_total=True))
    assert len(p.items) == 3 and p.total_count == 10 and p.has_more
    p = e.list_keys_paginated(Pagination(skip=8, limit=5, count_total=True))
    assert len(p.items) == 2 and p.total_count == 10 and not p.has_more
    p = e.list_keys_paginated(Pagination(skip=0, limit=4))       # no total: has_more = (items == limit)
    assert len(p.items) == 4 and p.total_count is None and p.has_more
    p = e.list_keys_paginated(Pagination(skip=0, limit=20))
    assert len(p.items) == 10 and not p.has_more
    p = e.list_keys_paginated(Pagination())                      # no limit: everything, has_more = (items == 0)
    assert len(p.items) == 10 and not p.has_more
    pages = [e.list_keys_paginated(Pagination(skip=s, limit=4)).items for s in (0, 4, 8)]
    assert sorted(sum(pages, [])) == sorted(f"v{i}" for i in range(10))


def test_list_keys_paginated_is_bounded_by_max_keys_per_scan():  # fetch_limit = min(skip + limit, max_scan), lib.rs:2950-2954
    e = _filled(10, max_keys_per_scan=5)
    assert len(e.list_keys_paginated(Pagination(skip=0, limit=8)).items) == 5
    assert len(e.list_keys_paginated(Pagination(skip=3, limit=8)).items) == 2
    assert e.list_keys_paginated(Pagination(skip=7, limit=2)).items == []
    p = e.list_keys_paginated(Pagination(skip=0, limit=None, count_total=True))
    assert len(p.items) == 5 and p.total_count == 10 and p.has_more  # total_count = count(), not the bounded scan


def test_scan_entities_with_embeddings_is_bounded():  # lib.rs:3224-3237
    e = VectorEngine(VectorEngineConfig(max_keys_per_scan=3))
    for i in range(7):
        e.set_entity_embedding(f"user:{i}", [1.0, float(i)])
    assert len(e.scan_entities_with_embeddings()) == 3 and e.count_entities_with_embeddings() == 3
    free = VectorEngine()
    for i in range(7):
        free.set_entity_embedding(f"user:{i}", [1.0, float(i)])
    assert len(free.scan_entities_with_embeddings()) == 7 and free.count_entities_with_embeddings() == 7
