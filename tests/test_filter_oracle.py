"""Pins oracle/filter_oracle.py against the reference's own filter tests (vector_engine/src/lib.rs
6968-7722): every assertion below restates one of those tests at the predicate level (which stored
items match), which is what the filtered searches there count."""
from oracle import filter_oracle as fo

ITEMS = {  # setup_filtered_search_engine (lib.rs:6968-7001)
    "item0": {"category": "electronics", "price": 100, "active": True},
    "item1": {"category": "clothing", "price": 50, "active": False},
    "item2": {"category": "food", "price": 25, "active": True},
}


def matching(cond, items=ITEMS):
    return sorted(k for k, m in items.items() if fo.evaluate(m, cond))


def test_comparison_operators_on_the_reference_fixture():
    assert matching(("eq", "category", "electronics")) == ["item0"]                    # lib.rs:7004-7017
    assert matching(("eq", "price", 50)) == ["item1"]                                  # 7020-7030
    assert len(matching(("gt", "price", 30))) == 2                                     # 7033-7042
    assert len(matching(("lt", "price", 60))) == 2                                     # 7045-7054
    assert len(matching(("le", "price", 50))) == 2                                     # 7057-7067
    assert len(matching(("ge", "price", 50))) == 2                                     # 7070-7080
    assert matching(("and", ("gt", "price", 30), ("lt", "price", 80))) == ["item1"]    # 7083-7095
    assert len(matching(("or", ("eq", "category", "electronics"), ("eq", "category", "food")))) == 2  # 7098-7114
    assert len(matching(("true",))) == 3                                               # 7117-7126
    assert len(matching(("in", "category", ["electronics", "food"]))) == 2             # 7277-7292
    assert len(matching(("ne", "category", "electronics"))) == 2                       # 7295-7307
    assert len(matching(("eq", "active", True))) == 2                                  # 7310-7319
    assert matching(("eq", "category", "nonexistent")) == []                           # 7322-7334
    assert matching(("eq", "missing_field", 1)) == []                                  # 7261-7274


def test_exists_contains_starts_with():
    items = {"with_tag": {"tag": "important"}, "without_tag": {}}                       # lib.rs:7129-7152
    assert matching(("exists", "tag"), items) == ["with_tag"]
    items = {"item1": {"description": "blue shirt"}, "item2": {"description": "red pants"}}  # 7155-7183
    assert matching(("contains", "description", "shirt"), items) == ["item1"]
    items = {"item": {"count": 42}}                                                      # 7186-7205
    assert matching(("contains", "count", "4"), items) == []
    items = {"item1": {"sku": "ABC123"}, "item2": {"sku": "XYZ789"}}                     # 7208-7236
    assert matching(("startswith", "sku", "ABC"), items) == ["item1"]
    items = {"item": {"count": 123}}                                                     # 7239-7258
    assert matching(("startswith", "count", "1"), items) == []


def test_typed_comparisons():
    items = {"high": {"score": 0.95}, "low": {"score": 0.5}}                            # lib.rs:7491-7519
    assert matching(("gt", "score", 0.8), items) == ["high"]
    assert matching(("gt", "value", 50), {"item": {"value": 50.5}}) == ["item"]        # 7522-7541 float field, int filter
    items = {"item": {"count": 100}}                                                    # 7544-7572 int field, float filter
    assert matching(("gt", "count", 50.5), items) == ["item"]
    assert matching(("gt", "count", 100.0), items) == []
    items = {"with_null": {"optional": None}, "without_field": {}}                      # 7575-7601
    assert matching(("eq", "optional", None), items) == ["with_null"]
    items = {"item1": {"name": "apple"}, "item2": {"name": "banana"}}                   # 7604-7644
    assert matching(("gt", "name", "app"), items) == ["item1", "item2"]
    assert matching(("le", "name", "apple"), items) == ["item1"]
    items = {"active_item": {"active": True}, "inactive_item": {"active": False}}      # 7647-7676
    assert matching(("eq", "active", False), items) == ["inactive_item"]
    assert matching(("eq", "value", 42), {"item": {"value": "text"}}) == []            # 7679-7698 incompatible types


def test_incomparable_pairs_are_false_for_every_operator():
    """`ordering.is_some_and(cmp)` (lib.rs:3644): None is false for Ne too; NaN makes partial_cmp None."""
    for op in ("eq", "ne", "lt", "le", "gt", "ge"):
        assert not fo.evaluate({"v": "text"}, (op, "v", 1))
        assert not fo.evaluate({"v": float("nan")}, (op, "v", 1.0))
        assert not fo.evaluate({"v": 1.0}, (op, "v", float("nan")))
        assert not fo.evaluate({"v": True}, (op, "v", 1))
        assert not fo.evaluate({"v": None}, (op, "v", 0))
    assert fo.compare(2**53 + 1, float(2**53)) == 0      # `*a as f64` rounds to even before comparing
    assert fo.compare(float(2**53), 2**53 + 1) == 0
    assert fo.compare(2**53 + 1, 2**53) == 1             # Int/Int stays exact
    assert fo.compare("é", "z") == 1                     # bytewise UTF-8 order, as String::cmp
