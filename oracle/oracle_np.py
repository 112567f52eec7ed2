"""numpy twin of oracle/nmn_oracle.c — an independent restatement of the same reference code
(tensor_store/src/hnsw.rs:168-229, vector_engine/src/lib.rs:2231-2266), vectorised over ROWS so
each row still sees the reference's exact operation order.  numpy's f32 multiply and add are
separate, correctly rounded ufuncs (no fused multiply-add).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used to cross-check the C oracle and to
generate the committed fixtures under tests/golden/ (tests/golden/make_golden.py).
"""
import numpy as np

COSINE, EUCLIDEAN, DOT, SPARSE_COS64 = 0, 1, 2, 3
F = np.float32


def _lanes_dot(A, q):
    """A: [n,d] f32, q: [d] f32 -> dot8 per row, reference lane order (hnsw.rs:168-193)."""
    A = np.asarray(A, dtype=F)
    q = np.asarray(q, dtype=F)
    n, d = A.shape
    chunks, rem = divmod(d, 8)
    acc = np.zeros((n, 8), dtype=F)
    for c in range(chunks):
        acc = acc + A[:, 8 * c:8 * c + 8] * q[8 * c:8 * c + 8]
    r = np.full(n, -0.0, dtype=F)
    for l in range(8):
        r = r + acc[:, l]
    for i in range(chunks * 8, chunks * 8 + rem):
        r = r + A[:, i] * q[i]
    return r


def dot8(a, b):
    return _lanes_dot(np.asarray(a, dtype=F)[None, :], b)[0]


def sumsq8_rows(A):
    """sum_of_squares per row (hnsw.rs:198-222)."""
    A = np.asarray(A, dtype=F)
    n, d = A.shape
    chunks, rem = divmod(d, 8)
    acc = np.zeros((n, 8), dtype=F)
    for c in range(chunks):
        blk = A[:, 8 * c:8 * c + 8]
        acc = acc + blk * blk
    r = np.full(n, -0.0, dtype=F)
    for l in range(8):
        r = r + acc[:, l]
    for i in range(chunks * 8, chunks * 8 + rem):
        r = r + A[:, i] * A[:, i]
    return r


def magnitude_rows(A):
    return np.sqrt(sumsq8_rows(A))


def magnitude(v):
    return magnitude_rows(np.asarray(v, dtype=F)[None, :])[0]


def euclid_rows(A, q):
    """sequential sum of squared differences, then sqrt (lib.rs:2249-2253)."""
    A = np.asarray(A, dtype=F)
    q = np.asarray(q, dtype=F)
    s = np.full(A.shape[0], -0.0, dtype=F)
    for i in range(A.shape[1]):
        diff = q[i] - A[:, i]  # (x - y), x = query (first argument of compute_score's call)
        s = s + diff * diff
    return np.sqrt(s)


def sparse_cos64_rows(A, q):
    """tensor_blob artifact similarity for every row (tensor_blob/src/lib.rs:601-603; sparse_vector.rs:419-443,
    553-559, 583-599): sequential f64 sums over the positions both vectors store (value != 0.0)."""
    A = np.asarray(A, dtype=F)
    q = np.asarray(q, dtype=F)
    n = A.shape[0]
    dot = np.zeros(n, np.float64)
    sa = np.float64(0.0)
    sb = np.zeros(n, np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for i in range(A.shape[1]):
            x, y = np.float64(q[i]), A[:, i].astype(np.float64)
            sx, sy = q[i] != 0, A[:, i] != 0
            if sx:
                sa = sa + x * x
                dot = dot + np.where(sy, x * y, 0.0)
            sb = sb + np.where(sy, y * y, 0.0)
        mag_a, mag_b = np.sqrt(sa), np.sqrt(sb)
        r = dot / (mag_a * mag_b)
    out = np.where(np.isnan(r) | np.isinf(r), 0.0, np.clip(r, -1.0, 1.0))
    out[(mag_b == 0) | (mag_a == 0)] = 0.0
    return out.astype(F)


def scores(A, q, metric):
    """compute_score for every row (lib.rs:2231-2266)."""
    A = np.asarray(A, dtype=F)
    q = np.asarray(q, dtype=F)
    if metric == SPARSE_COS64:
        return sparse_cos64_rows(A, q)
    if metric == DOT:
        return _lanes_dot(A, q)
    if metric == EUCLIDEAN:
        return F(1.0) / (F(1.0) + euclid_rows(A, q))
    qmag = magnitude(q)
    dot = _lanes_dot(A, q)
    vmag = magnitude_rows(A)
    den = qmag * vmag
    with np.errstate(divide="ignore", invalid="ignore"):
        out = dot / den
    out[(vmag == 0) | (qmag == 0)] = F(0.0)
    return out.astype(F)


def search(A, q, k, metric=COSINE, keep=None, row_base=0):
    """search_similar_with_metric: (rows, scores), score desc / row asc, NaN last."""
    A = np.asarray(A, dtype=F)
    q = np.asarray(q, dtype=F)
    if q.size == 0:
        raise ValueError("EmptyVector")
    if k == 0:
        raise ValueError("InvalidTopK")
    if magnitude(q) == 0 and metric in (COSINE, DOT):
        return np.zeros(0, np.uint64), np.zeros(0, F)
    s = scores(A, q, metric) if A.shape[0] else np.zeros(0, F)
    rows = np.arange(A.shape[0], dtype=np.uint64)
    if keep is not None:
        keep = np.asarray(keep, dtype=bool)
        s, rows = s[keep], rows[keep]
    nan = np.isnan(s)
    order = np.lexsort((rows, -np.where(nan, F(0), s).astype(np.float64), nan))
    order = order[:k]
    return rows[order] + np.uint64(row_base), s[order]
