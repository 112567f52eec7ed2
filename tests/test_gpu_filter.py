"""On-device WHERE-predicate evaluation: the columnar metadata + predicate kernel of
neumann_amd/csrc/nmn_columns.hip against oracle/filter_oracle.py (the restatement of the reference's
evaluate_filter, vector_engine/src/lib.rs:3592-3692), bit for bit on the selection bitmap, and the
engine's pre-filtered SIMILAR on top of it against the oracle's masked search."""
import numpy as np
import pytest

from oracle import filter_oracle as fo
from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
F = np.float32

FIELDS = ["price", "score", "category", "active", "opt", "mixed"]
WORDS = ["", "a", "ab", "abc", "electronics", "clothing", "food", "Food", "é", "zeta", "abé"]


def random_value(rng, field):
    if field == "price":
        return int(rng.integers(-5, 60))
    if field == "score":
        return [0.5, 0.95, -0.0, 0.0, float("nan"), float("inf"), -1e300, 49.5, 50.0][int(rng.integers(0, 9))]
    if field == "category":
        return WORDS[int(rng.integers(0, len(WORDS)))]
    if field == "active":
        return bool(rng.integers(0, 2))
    if field == "opt":
        return None
    # "mixed": any type in the same column
    pick = int(rng.integers(0, 6))
    return [None, True, 7, 7.0, "7", 2**53 + 1][pick]


def random_meta(rng):
    return {f: random_value(rng, f) for f in FIELDS if rng.random() < 0.7}


def random_filter_value(rng):
    pick = int(rng.integers(0, 9))
    return [None, True, False, int(rng.integers(-5, 60)), 50.0, 49.5, float("nan"),
            WORDS[int(rng.integers(0, len(WORDS)))], float(2**53)][pick]


def random_cond(rng, depth=0):
    r = rng.random()
    if depth < 4 and r < 0.35:
        return ("and" if rng.random() < 0.5 else "or", random_cond(rng, depth + 1), random_cond(rng, depth + 1))
    field = (FIELDS + ["nosuchfield"])[int(rng.integers(0, len(FIELDS) + 1))]
    r = rng.random()
    if r < 0.05:
        return ("true",)
    if r < 0.15:
        return ("exists", field)
    if r < 0.25:
        return ("contains", field, ["", "a", "b", "oo", "é"][int(rng.integers(0, 5))])
    if r < 0.35:
        return ("startswith", field, ["", "a", "ab", "F", "é"][int(rng.integers(0, 5))])
    if r < 0.5:
        return ("in", field, [random_filter_value(rng) for _ in range(int(rng.integers(0, 5)))])
    return (["eq", "ne", "lt", "le", "gt", "ge"][int(rng.integers(0, 6))], field, random_filter_value(rng))


def to_fc(E, cond):
    FC = E.FilterCondition
    op = cond[0]
    if op == "true":
        return FC.TRUE
    if op == "and":
        return to_fc(E, cond[1]).and_(to_fc(E, cond[2]))
    if op == "or":
        return to_fc(E, cond[1]).or_(to_fc(E, cond[2]))
    if op == "exists":
        return FC.Exists(cond[1])
    if op == "contains":
        return FC.Contains(cond[1], cond[2])
    if op == "startswith":
        return FC.StartsWith(cond[1], cond[2])
    if op == "in":
        return FC.In(cond[1], cond[2])
    return getattr(FC, op.capitalize())(cond[1], cond[2])


# ---- the C ABI directly: cells, programs, bitmap ---------------------------------------------------------
def encode_column(values, n):
    """values: {row: python value}.  Returns kinds u8[n], payload u64[n], dictionary list."""
    from neumann_amd import columns as gc
    kinds = np.zeros(n, np.uint8)
    payload = np.zeros(n, np.uint64)
    strings, ids = [], {}
    for r, v in values.items():
        if v is None:
            kinds[r] = gc.CELL_NULL
        elif isinstance(v, bool):
            kinds[r], payload[r] = gc.CELL_BOOL, int(v)
        elif isinstance(v, int):
            kinds[r], payload[r] = gc.CELL_INT, gc.i64_bits(v)
        elif isinstance(v, float):
            kinds[r], payload[r] = gc.CELL_FLOAT, gc.f64_bits(v)
        else:
            if v not in ids:
                ids[v] = len(strings)
                strings.append(v)
            kinds[r], payload[r] = gc.CELL_STRING, ids[v]
    return kinds, payload, strings


def compile_cond(cond, cols, consts):
    """Straightforward postfix translation of a tuple condition (test-side twin of the engine's
    compile_filter, without its operand reordering).  cols: field -> (column id, dictionary)."""
    from neumann_amd import columns as gc
    op = cond[0]
    if op == "true":
        return [(gc.PRED_TRUE, 0, 0, 0, 0, 0)]
    if op in ("and", "or"):
        return compile_cond(cond[1], cols, consts) + compile_cond(cond[2], cols, consts) + \
            [(gc.PRED_AND if op == "and" else gc.PRED_OR, 0, 0, 0, 0, 0)]
    if cond[1] not in cols:
        return [(gc.PRED_FALSE, 0, 0, 0, 0, 0)]
    cid, strings = cols[cond[1]]

    def strset(test):
        off = len(consts)
        consts.extend([0] * ((len(strings) + 63) // 64))
        for i, s in enumerate(strings):
            if test(s):
                consts[off + i // 64] |= 1 << (i % 64)
        return (gc.PRED_STRSET, 0, 0, cid, off, len(strings))

    def cell(v):
        if v is None:
            return gc.CELL_NULL, 0
        if isinstance(v, bool):
            return gc.CELL_BOOL, int(v)
        if isinstance(v, int):
            return gc.CELL_INT, gc.i64_bits(v)
        return gc.CELL_FLOAT, gc.f64_bits(v)

    if op == "exists":
        return [(gc.PRED_EXISTS, 0, 0, cid, 0, 0)]
    if op == "contains":
        return [strset(lambda s: cond[2] in s)]
    if op == "startswith":
        return [strset(lambda s: s.startswith(cond[2]))]
    if op == "in":
        out = []
        scal = [v for v in cond[2] if not isinstance(v, str)]
        strs = [v for v in cond[2] if isinstance(v, str)]
        if scal:
            off = len(consts)
            for v in scal:
                consts.extend(cell(v))
            out.append((gc.PRED_IN, 0, 0, cid, off, len(scal)))
        if strs:
            out.append(strset(lambda s: s in strs))
            if scal:
                out.append((gc.PRED_OR, 0, 0, 0, 0, 0))
        return out or [(gc.PRED_FALSE, 0, 0, 0, 0, 0)]
    cmp_code = ["eq", "ne", "lt", "le", "gt", "ge"].index(op)
    if isinstance(cond[2], str):
        tests = [lambda o: o == 0, lambda o: o != 0, lambda o: o < 0, lambda o: o <= 0, lambda o: o > 0, lambda o: o >= 0]
        b = cond[2].encode()
        return [strset(lambda s: tests[cmp_code]((s.encode() > b) - (s.encode() < b)))]
    vk, vp = cell(cond[2])
    return [(gc.PRED_CMP, cmp_code, vk, cid, vp, 0)]


def bits(words, n):
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 20011])
def test_predicate_kernel_matches_oracle_bitmap(n):
    from neumann_amd.columns import GpuColumns
    rng = np.random.default_rng(1000 + n)
    metas = [random_meta(rng) for _ in range(n)]
    valid = rng.random(n) < 0.9
    with GpuColumns(n + 100) as gc:
        cols = {}
        for f in FIELDS:
            kinds, payload, strings = encode_column({r: m[f] for r, m in enumerate(metas) if f in m}, n)
            cid = gc.add_column()
            gc.write(cid, 0, kinds, payload)
            cols[f] = (cid, strings)
        assert gc.n_columns == len(FIELDS)
        gc.write_valid(0, oc.mask_from_bool(valid))
        for _ in range(60):
            cond = random_cond(rng)
            consts = []
            prog = compile_cond(cond, cols, consts)
            cnt = gc.eval(prog, consts, n)
            got = bits(gc.read_mask(n), n)
            exp = np.array(fo.mask(metas, cond), bool) & valid
            assert np.array_equal(got, exp), cond
            assert cnt == int(exp.sum())


def test_predicate_program_validation_and_cell_updates():
    from neumann_amd import NeumannGpuError
    from neumann_amd import columns as g
    with g.GpuColumns(256) as gc:
        c0 = gc.add_column()
        gc.write_valid(0, np.full(4, 0xFFFFFFFFFFFFFFFF, np.uint64))
        T, A = (g.PRED_TRUE, 0, 0, 0, 0, 0), (g.PRED_AND, 0, 0, 0, 0, 0)
        for bad in ([A], [T, T], [T, A], [(g.PRED_EXISTS, 0, 0, 5, 0, 0)], [(99, 0, 0, 0, 0, 0)],
                    [(g.PRED_IN, 0, 0, c0, 0, 3)], [(g.PRED_STRSET, 0, 0, c0, 0, 65)], [T] * 65 + [A] * 64):
            with pytest.raises(NeumannGpuError):
                gc.eval(bad, [], 200)
        assert gc.eval([T] * 64 + [A] * 63, [], 200) == 200      # the deepest stack the kernel supports
        assert gc.eval([(g.PRED_EXISTS, 0, 0, c0, 0, 0)], [], 200) == 0
        # single-cell writes, overwrite, clear_row
        gc.write(c0, 7, [g.CELL_INT], [g.i64_bits(-3)])
        gc.write(c0, 130, [g.CELL_INT], [g.i64_bits(5)])
        lt0 = [(g.PRED_CMP, g.CMP_LT, g.CELL_INT, c0, g.i64_bits(0), 0)]
        assert gc.eval(lt0, [], 200) == 1 and np.flatnonzero(bits(gc.read_mask(200), 200)).tolist() == [7]
        gc.write(c0, 7, [g.CELL_FLOAT], [g.f64_bits(0.5)])       # the cell changes type
        assert gc.eval(lt0, [], 200) == 0
        assert gc.eval([(g.PRED_CMP, g.CMP_GT, g.CELL_INT, c0, g.i64_bits(0), 0)], [], 200) == 2
        gc.clear_row(130)
        assert gc.eval([(g.PRED_EXISTS, 0, 0, c0, 0, 0)], [], 200) == 1
        # rows beyond n_rows and invalid rows never show up
        gc.write(c0, 250, [g.CELL_INT], [1])
        assert gc.eval([(g.PRED_EXISTS, 0, 0, c0, 0, 0)], [], 200) == 1
        gc.write_valid(0, np.zeros(1, np.uint64))
        assert gc.eval([(g.PRED_EXISTS, 0, 0, c0, 0, 0)], [], 200) == 0
        with pytest.raises(NeumannGpuError):
            gc.write(c0, 250, np.zeros(10, np.uint8), np.zeros(10, np.uint64))   # beyond capacity
        with pytest.raises(NeumannGpuError):
            gc.write(c0, 0, [9], [0])                                            # bad kind


def test_int_float_widening_matches_rust_as_f64():
    """`(*a as f64).partial_cmp(b)` (lib.rs:3658-3663): the i64 side is rounded to nearest-even f64."""
    from neumann_amd import columns as g
    ints = [2**53, 2**53 + 1, 2**53 + 2, 2**53 + 3, -(2**53) - 1, 2**63 - 1, -(2**63), 2**62 + 2**9 + 1, 0, -1]
    fl = [float(2**53), float(2**53 + 2), float(2**63), -float(2**63), float(2**62 + 2**10), 0.0, -0.0, -1.0]
    with g.GpuColumns(64) as gc:
        ci, cf = gc.add_column(), gc.add_column()
        gc.write(ci, 0, [g.CELL_INT] * len(ints), [g.i64_bits(v) for v in ints])
        gc.write(cf, 0, [g.CELL_FLOAT] * len(fl), [g.f64_bits(v) for v in fl])
        gc.write_valid(0, np.full(1, 0xFFFFFFFFFFFFFFFF, np.uint64))
        ops = ["eq", "ne", "lt", "le", "gt", "ge"]
        for code, name in enumerate(ops):
            for v in fl:    # Int cells vs Float filter
                gc.eval([(g.PRED_CMP, code, g.CELL_FLOAT, ci, g.f64_bits(v), 0)], [], len(ints))
                got = bits(gc.read_mask(64), len(ints)).tolist()
                assert got == [fo.evaluate({"x": i}, (name, "x", v)) for i in ints], (name, v)
            for i in ints:  # Float cells vs Int filter
                gc.eval([(g.PRED_CMP, code, g.CELL_INT, cf, g.i64_bits(i), 0)], [], len(fl))
                got = bits(gc.read_mask(64), len(fl)).tolist()
                assert got == [fo.evaluate({"x": v}, (name, "x", i)) for v in fl], (name, i)


def test_search_with_device_bitmap_equals_host_bitmap():
    from neumann_amd import GpuFlatIndex
    from neumann_amd import columns as g
    n, d, k = 30000, 96, 40
    A = oc.synth(5, 0, n, d)
    q = oc.synth(6, 0, 1, d)[0]
    bucket = np.arange(n) % 11
    with GpuFlatIndex(d, n) as idx, g.GpuColumns(n) as gc:
        idx.upload(A)
        c = gc.add_column()
        gc.write(c, 0, np.full(n, g.CELL_INT, np.uint8), bucket.astype(np.uint64))
        gc.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
        cnt = gc.eval([(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, c, 3, 0)], [], n)
        assert cnt == int((bucket == 3).sum())
        for metric in (0, 1, 2):
            rows, scores, counts = idx.search_dmask(q, k, metric, gc.mask_device)
            er, es = oc.search(A, q, k, metric, mask=oc.mask_from_bool(bucket == 3))
            assert counts[0] == k and np.array_equal(rows[0], er) and np.all(scores[0] == es)


# ---- through the engine: search_similar_filtered with the pre-filter strategy ------------------------------
@pytest.fixture
def E():
    from neumann_amd import engine
    return engine


def check_filtered(E, engine, A, metas, live, q, k, cond, coll=None):
    keep = np.array([live[i] and fo.evaluate(metas[i], cond) for i in range(len(metas))], bool)
    cfg = E.FilteredSearchConfig.pre_filter()
    if coll is None:
        res = engine.search_similar_filtered(q, k, to_fc(E, cond), cfg)
    else:
        res = engine.search_filtered_in_collection(coll, q, k, to_fc(E, cond), cfg)
    if not keep.any():
        assert res == [], cond
        return
    er, es = oc.search(A, q, k, 0, mask=oc.mask_from_bool(keep))
    assert [r.key for r in res] == [f"k{i}" for i in er], cond
    assert np.all(np.array([r.score for r in res], F) == es), cond


def test_engine_prefilter_runs_on_device_and_matches_oracle(E):
    rng = np.random.default_rng(77)
    n, d, k = 3000, 48, 20
    A = rng.standard_normal((n, d)).astype(F)
    metas = [random_meta(rng) for _ in range(n)]
    live = [True] * n
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_embedding_with_metadata(f"k{i}", A[i], metas[i])
    q = rng.standard_normal(d).astype(F)
    assert engine.device_filter_evals() == 0 and engine.column_builds() == 0
    for t in range(40):
        check_filtered(E, engine, A, metas, live, q, k, random_cond(rng))
    assert engine.device_filter_evals() == 40 and engine.column_builds() == 1 and engine.mirror_builds() == 1

    # churn: overwrites that change the metadata (and its types), deletes, new keys with new fields —
    # the columns are patched in place, never rebuilt
    for i in rng.choice(n, 200, replace=False):
        i = int(i)
        A[i] = rng.standard_normal(d).astype(F)
        metas[i] = random_meta(rng)
        engine.store_embedding_with_metadata(f"k{i}", A[i], metas[i])
    for i in rng.choice(n, 150, replace=False):
        i = int(i)
        if live[i]:
            engine.delete_embedding(f"k{i}")
            live[i] = False
    extra = 300
    A = np.vstack([A, rng.standard_normal((extra, d)).astype(F)])
    for i in range(n, n + extra):
        m = random_meta(rng)
        if i % 3 == 0:
            m["brand_new_field"] = i % 5
        metas.append(m)
        live.append(True)
        engine.store_embedding_with_metadata(f"k{i}", A[i], m)
    # the oracle sees dead rows as absent: give it the live rows only, keeping ids via the mask
    for t in range(40):
        check_filtered(E, engine, A, metas, live, q, k, random_cond(rng))
    check_filtered(E, engine, A, metas, live, q, k, ("ge", "brand_new_field", 3))
    check_filtered(E, engine, A, metas, live, q, k, ("exists", "brand_new_field"))
    assert engine.column_builds() == 1 and engine.mirror_builds() == 1
    assert engine.device_filter_evals() == 82


def test_engine_prefilter_deeply_nested_condition(E):
    """A 300-deep right-nested And/Or chain: the compiler's operand ordering keeps the device stack shallow."""
    rng = np.random.default_rng(5)
    n, d = 500, 16
    A = rng.standard_normal((n, d)).astype(F)
    metas = [{"price": int(rng.integers(0, 1000)), "category": WORDS[int(rng.integers(0, len(WORDS)))]} for _ in range(n)]
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_embedding_with_metadata(f"k{i}", A[i], metas[i])
    cond = ("lt", "price", 5)
    for j in range(300):
        leaf = ("eq", "price", 7 * j) if j % 2 else ("ne", "category", WORDS[j % len(WORDS)])
        cond = ("or" if j % 3 else "and", leaf, cond)
    check_filtered(E, engine, A, metas, [True] * n, rng.standard_normal(d).astype(F), 10, cond)
    assert engine.device_filter_evals() == 1


def test_engine_prefilter_in_collection_and_no_metadata(E):
    rng = np.random.default_rng(9)
    n, d = 400, 8
    A = rng.standard_normal((n, d)).astype(F)
    metas = [{"g": int(i % 4)} if i % 2 else {} for i in range(n)]
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_in_collection_with_metadata("c", f"k{i}", A[i], metas[i])
    q = rng.standard_normal(d).astype(F)
    for cond in (("eq", "g", 1), ("true",), ("exists", "g"), ("eq", "nosuch", 1), ("in", "g", [1, 3, "x"])):
        check_filtered(E, engine, A, metas, [True] * n, q, 15, cond, coll="c")
    # an engine whose rows carry no metadata at all: every field predicate is false, True selects all
    plain = E.VectorEngine()
    for i in range(50):
        plain.store_embedding(f"k{i}", A[i])
    check_filtered(E, plain, A[:50], [{}] * 50, [True] * 50, q, 5, ("true",))
    check_filtered(E, plain, A[:50], [{}] * 50, [True] * 50, q, 5, ("exists", "g"))


def test_engine_concurrent_prefiltered_searches_share_sweeps(E):
    """Filtered searches of many threads run under the engine's shared lock: each predicate evaluation gets its own
    device bitmap, and the searches consuming them merge into query batches with one bitmap per query.  Every thread
    must get exactly the answer of its own filter (checked against the oracle)."""
    import threading
    rng = np.random.default_rng(91)
    n, d, k = 6000, 768, 15                      # 768: batches take the matrix-core sweep from 3 queries
    A = rng.standard_normal((n, d)).astype(F)
    metas = [random_meta(rng) for _ in range(n)]
    live = [True] * n
    engine = E.VectorEngine()
    for i in range(n):
        engine.store_embedding_with_metadata(f"k{i}", A[i], metas[i])
    conds = [random_cond(rng) for _ in range(48)]
    qs = rng.standard_normal((48, d)).astype(F)
    cfg = E.FilteredSearchConfig.pre_filter()
    engine.search_similar_filtered(qs[0], k, to_fc(E, conds[0]), cfg)     # builds mirror + columns (exclusive lock)
    fcs = [to_fc(E, c) for c in conds]
    out = [None] * 48
    errs = []
    start = threading.Barrier(16)

    def work(t):
        try:
            start.wait()
            for rep in range(3):
                for j in range(t, 48, 16):
                    out[j] = engine.search_similar_filtered(qs[j], k, fcs[j], cfg)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    assert engine.column_builds() == 1 and engine.mirror_builds() == 1
    assert engine.device_filter_evals() == 1 + 48 * 3
    for j in range(48):
        keep = np.array([fo.evaluate(metas[i], conds[j]) for i in range(n)], bool)
        if not keep.any():
            assert out[j] == []
            continue
        er, es = oc.search(A, qs[j], k, 0, mask=oc.mask_from_bool(keep))
        assert [r.key for r in out[j]] == [f"k{i}" for i in er], conds[j]
        assert np.all(np.array([r.score for r in out[j]], F) == es), conds[j]


@pytest.mark.parametrize("n,d", [(30000, 768), (30000, 96)])   # with and without the matrix-core sweep for batches
def test_search_pred_one_call_equals_eval_then_search(n, d):
    """nmn_index_search_pred: predicate + search in one call, alone and from many threads with different predicates
    (their evaluations share one launch, their searches one sweep) — always the oracle's answer for that filter."""
    import threading
    from neumann_amd import GpuFlatIndex
    from neumann_amd import columns as g
    k = 30
    A = oc.synth(15, 0, n, d)
    Q = oc.synth(16, 0, 40, d)
    bucket = np.arange(n) % 13
    price = (np.arange(n) * 7919) % 1000
    with GpuFlatIndex(d, n) as idx, g.GpuColumns(n) as gc:
        idx.upload(A)
        cb, cp = gc.add_column(), gc.add_column()
        gc.write(cb, 0, np.full(n, g.CELL_INT, np.uint8), bucket.astype(np.uint64))
        gc.write(cp, 0, np.full(n, g.CELL_INT, np.uint8), price.astype(np.uint64))
        gc.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
        progs = []
        for j in range(40):
            if j % 4 == 0:    # bucket == b
                progs.append(([(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, cb, j % 13, 0)], bucket == j % 13))
            elif j % 4 == 1:  # price < t
                t = 50 + 20 * j
                progs.append(([(g.PRED_CMP, g.CMP_LT, g.CELL_INT, cp, t, 0)], price < t))
            elif j % 4 == 2:  # bucket == b AND price >= t
                t = 300
                progs.append(([(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, cb, j % 13, 0), (g.PRED_CMP, g.CMP_GE, g.CELL_INT, cp, t, 0),
                               (g.PRED_AND, 0, 0, 0, 0, 0)], (bucket == j % 13) & (price >= t)))
            else:             # nothing passes
                progs.append(([(g.PRED_CMP, g.CMP_GT, g.CELL_INT, cp, 5000, 0)], np.zeros(n, bool)))

        def check(j, out, metric):
            rows, scores, counts, selected = out
            keep = progs[j][1]
            assert selected == int(keep.sum()), j
            if not keep.any():
                assert counts[0] == 0
                return
            er, es = oc.search(A, Q[j], k, metric, mask=oc.mask_from_bool(keep))
            c = er.size
            assert counts[0] == c and np.array_equal(rows[0, :c], er) and np.all(scores[0, :c] == es), j

        for j in (0, 1, 2, 3):                                  # alone
            for metric in (0, 1, 2):
                check(j, idx.search_pred(gc, progs[j][0], [], Q[j], k, metric), metric)
        out = [None] * 40
        errs = []
        start = threading.Barrier(20)

        def work(t):
            try:
                start.wait()
                for rep in range(3):
                    for j in range(t, 40, 20):
                        out[j] = idx.search_pred(gc, progs[j][0], [], Q[j], k, 0)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(t,)) for t in range(20)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for j in range(40):
            check(j, out[j], 0)


def test_search_pred_with_several_queries_and_few_matches():
    """One predicate, several queries in the call; fewer matching rows than k; and an empty selection."""
    from neumann_amd import GpuFlatIndex
    from neumann_amd import columns as g
    n, d, k = 20000, 768, 25
    A = oc.synth(25, 0, n, d)
    Q = oc.synth(26, 0, 5, d)
    tag = np.arange(n) % 2000
    with GpuFlatIndex(d, n) as idx, g.GpuColumns(n) as gc:
        idx.upload(A)
        c = gc.add_column()
        gc.write(c, 0, np.full(n, g.CELL_INT, np.uint8), tag.astype(np.uint64))
        gc.write_valid(0, np.full((n + 63) // 64, 0xFFFFFFFFFFFFFFFF, np.uint64))
        for value, metric in ((7, 0), (7, 1), (1999, 2), (5000, 0)):
            keep = tag == value                                   # 10 rows match (or none)
            rows, scores, counts, selected = idx.search_pred(gc, [(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, c, value, 0)], [], Q, k, metric)
            assert selected == int(keep.sum())
            for qi in range(5):
                if not keep.any():
                    assert counts[qi] == 0
                    continue
                er, es = oc.search(A, Q[qi], k, metric, mask=oc.mask_from_bool(keep))
                cnt = er.size
                assert counts[qi] == cnt == 10
                assert np.array_equal(rows[qi, :cnt], er) and np.all(scores[qi, :cnt] == es)
                assert np.all(rows[qi, cnt:] == np.uint64(0xFFFFFFFFFFFFFFFF))
