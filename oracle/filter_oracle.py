"""CPU restatement of the reference's FilterCondition evaluator — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module (see oracle/nmn_oracle.c's header for the rule); the product
evaluates predicates with the HIP kernel in neumann_amd/csrc/nmn_columns.hip.

Follows vector_engine/src/lib.rs:
  evaluate_filter                      3592-3630
  compare_field                        3633-3645   (missing field -> false; `ordering.is_some_and(cmp)`)
  compare_tensor_value_to_filter       3648-3670   (typed comparison, None for incompatible types)
  string_contains / string_starts_with 3673-3692
Pinned against the reference's own filter tests (lib.rs:6968-7722) in tests/test_filter_oracle.py.

Conditions are plain tuples so the oracle does not depend on the product's classes:
  ("true",) ("and", a, b) ("or", a, b) ("exists", field) ("contains", field, s) ("startswith", field, s)
  ("in", field, [values]) ("eq"|"ne"|"lt"|"le"|"gt"|"ge", field, value)
Metadata is a dict field -> Python value: None (ScalarValue::Null), bool, int (i64), float (f64), str.
"""
import math


def _kind(v):
    if v is None:
        return "null"
    if isinstance(v, bool):  # before int: bool is an int subclass in Python
        return "bool"
    if isinstance(v, int):
        return "int"
    if isinstance(v, float):
        return "float"
    if isinstance(v, str):
        return "str"
    raise TypeError(f"unsupported metadata value {v!r}")


def _cmp3(a, b):
    return -1 if a < b else (1 if a > b else 0)


def _as_f64(i):
    """Rust `i64 as f64`: round to nearest, ties to even — Python's float(int) does the same."""
    return float(i)


def compare(stored, flt):
    """compare_tensor_value_to_filter (lib.rs:3648-3670): -1/0/+1, or None when incomparable."""
    ks, kf = _kind(stored), _kind(flt)
    if ks == "int" and kf == "int":
        return _cmp3(stored, flt)
    if ks == "float" and kf == "float":
        return None if math.isnan(stored) or math.isnan(flt) else _cmp3(stored, flt)  # partial_cmp
    if ks == "float" and kf == "int":
        return None if math.isnan(stored) else _cmp3(stored, _as_f64(flt))
    if ks == "int" and kf == "float":
        return None if math.isnan(flt) else _cmp3(_as_f64(stored), flt)
    if ks == "str" and kf == "str":
        return _cmp3(stored.encode("utf-8"), flt.encode("utf-8"))  # String::cmp is bytewise
    if ks == "bool" and kf == "bool":
        return _cmp3(int(stored), int(flt))
    if ks == "null" and kf == "null":
        return 0
    return None


_TESTS = {
    "eq": lambda o: o == 0, "ne": lambda o: o != 0, "lt": lambda o: o < 0,
    "le": lambda o: o <= 0, "gt": lambda o: o > 0, "ge": lambda o: o >= 0,
}


def _compare_field(meta, field, value, test):
    if field not in meta:
        return False
    o = compare(meta[field], value)
    return o is not None and test(o)


def evaluate(meta, cond):
    """evaluate_filter (lib.rs:3592-3630) for one row's metadata."""
    op = cond[0]
    if op == "true":
        return True
    if op == "and":
        return evaluate(meta, cond[1]) and evaluate(meta, cond[2])
    if op == "or":
        return evaluate(meta, cond[1]) or evaluate(meta, cond[2])
    if op == "exists":
        return cond[1] in meta
    if op == "contains":
        v = meta.get(cond[1], None)
        return cond[1] in meta and isinstance(v, str) and cond[2] in v
    if op == "startswith":
        v = meta.get(cond[1], None)
        return cond[1] in meta and isinstance(v, str) and v.startswith(cond[2])
    if op == "in":
        return any(_compare_field(meta, cond[1], v, _TESTS["eq"]) for v in cond[2])
    return _compare_field(meta, cond[1], cond[2], _TESTS[op])


def mask(rows_meta, cond):
    """bool list: evaluate(meta, cond) for every row."""
    return [evaluate(m, cond) for m in rows_meta]
