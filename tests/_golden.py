"""Helpers to load the committed fixtures of tests/golden/ (see tests/golden/make_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SYNTH_FILES = ["synth_10000x128_top5.npz", "synth_4096x768_top100.npz", "synth_4096x1536_top1000.npz",
               "synth_5000x100_top64.npz"]


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def synth_cases(g):
    """Yield (metric, query_index, tag, mask_words_or_None, exp_rows, exp_scores) of a synth fixture."""
    for key in g.files:
        if not key.startswith("rows_m"):
            continue
        _, m, q, tag = key.split("_")
        mask = None if tag == "all" else g[f"mask_{tag}"]
        yield int(m[1:]), int(q[1:]), tag, mask, g[key], g["scores_" + key[len("rows_"):]]


def rebuild_corpus(g, synth_fn):
    """corpus = generator(seed) with the planted rows written over it."""
    A = synth_fn(int(g["seed"]), 0, int(g["n"]), int(g["dim"]))
    if g["planted_idx"].size:
        A[g["planted_idx"]] = g["planted"]
    return A


# ---- fixtures of the widened rows (tests/golden/make_golden_next.py) ---------------------------------------
def _dec(v):
    tag = v[0]
    if tag == "null":
        return None
    if tag == "bool":
        return bool(v[1])
    if tag == "int":
        return int(v[1])
    if tag == "float":
        return float.fromhex(v[1])
    return v[1]


def _dec_cond(c):
    op = c[0]
    if op in ("and", "or"):
        return (op, _dec_cond(c[1]), _dec_cond(c[2]))
    if op == "in":
        return (op, c[1], [_dec(v) for v in c[2]])
    if op in ("eq", "ne", "lt", "le", "gt", "ge"):
        return (op, c[1], _dec(c[2]))
    return tuple(c)


def load_filters():
    """-> (rows: list of metadata dicts, cases: list of (condition tuple, selected row indices))."""
    import json
    with open(os.path.join(GOLDEN, "filters_mixed.json")) as fh:
        doc = json.load(fh)
    rows = [{k: _dec(v) for k, v in r.items()} for r in doc["rows"]]
    return rows, [(_dec_cond(c["cond"]), c["selected"]) for c in doc["cases"]]
