"""NMN_METRIC_SPARSE_COSINE_F64 — tensor_blob's artifact similarity (tensor_blob/src/lib.rs:591-625;
SparseVector::cosine_similarity, sparse_vector.rs:583-599) on the GPU against the oracle: identical
rows, bit-equal f32 similarities."""
import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
F = np.float32
M = oc.SPARSE_COS64


def check(idx, A, q, k, mask=None):
    rows, scores, counts = idx.search(q, k, M, mask=mask)
    er, es = oc.search(A, q, k, M, mask=mask)
    assert counts[0] == er.size
    assert np.array_equal(rows[0, :er.size], er), (rows[0, :8], er[:8])
    assert np.array_equal(scores[0, :er.size], es)


@pytest.mark.parametrize("n,d,density", [(20000, 64, 1.0), (30000, 96, 0.3), (5000, 768, 0.6), (3000, 13, 0.5)])
def test_sparse_cos64_matches_oracle(n, d, density):
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(n + d)
    A = (rng.standard_normal((n, d)) * (rng.random((n, d)) < density)).astype(F)
    A[5] = 0.0                       # zero artifact embedding: similarity 0.0, still listed
    A[9] = A[3]
    A[11] = -A[3]
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        for t in range(3):
            q = (rng.standard_normal(d) * (rng.random(d) < max(density, 0.5))).astype(F)
            for k in (1, 10, 500):
                check(idx, A, q, k)
            check(idx, A, q, 50, mask=oc.mask_from_bool(rng.random(n) < 0.3))
        check(idx, A, A[3], 20)                                    # self and duplicate at 1.0, the negation at -1.0
        check(idx, A, np.zeros(d, F), 10)                          # zero query: every similarity is 0.0, ids ascending
        check(idx, A, rng.standard_normal(d).astype(F), 6000 if n > 6000 else n)   # large-k path
        # exact scores of arbitrary rows through the rescore kernel
        rows = rng.integers(0, n, 200).astype(np.uint64)
        q = rng.standard_normal(d).astype(F)
        got = idx.score_rows(q, rows, M)
        exp = oc.scores_all(A[rows.astype(np.int64)], q, M)
        assert np.array_equal(got.reshape(-1), exp)


def test_sparse_cos64_non_finite_and_huge_values():
    """f32 overflow / NaN in the approximate sweep must not hide rows whose f64 similarity is finite."""
    from neumann_amd import GpuFlatIndex
    rng = np.random.default_rng(2)
    n, d = 4000, 32
    A = rng.standard_normal((n, d)).astype(F)
    A[0] *= F(1e30)                  # |v|^2 overflows f32, fine in f64
    A[1, 4] = np.nan                 # NaN component: similarity 0.0 (sanitised)
    A[2, 7] = np.inf                 # Inf component: 0.0
    A[3] = A[0]
    A[4] = (A[0] / F(1e30) * F(1e-30)).astype(F)   # |v|^2 underflows f32 to 0, fine in f64: similarity 1.0 too
    q = (A[0] / F(1e30)).astype(F)   # parallel to the huge row: similarity 1.0 for rows 0 and 3
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        rows, scores, counts = idx.search(q, 5, M)
        assert sorted(rows[0, :3].tolist()) == [0, 3, 4] and np.all(scores[0, :3] > 0.999999)
        for k in (5, 100, n):
            check(idx, A, q, k)
        qn = q.copy()
        qn[3] = np.nan               # NaN query: every similarity is 0.0
        check(idx, A, qn, 10)


# ---- engine level: BlobStore::{set_embedding, search_by_embedding, similar} (tensor_blob/src/lib.rs:520-625) ----
def test_engine_blob_similarity():
    from neumann_amd import engine as E
    rng = np.random.default_rng(12)
    n, d, k = 2500, 48, 15
    A = (rng.standard_normal((n, d)) * (rng.random((n, d)) < 0.5)).astype(F)
    eng = E.VectorEngine()
    for i in range(n):
        eng.blob_set_embedding(f"art{i}", f"file{i}.bin", A[i])
    eng.blob_set_embedding("other_dim", "x.bin", [1.0, 2.0, 3.0])     # `stored.len() == embedding.len()` filter
    q = rng.standard_normal(d).astype(F)
    res = eng.blob_search_by_embedding(q, k)
    er, es = oc.search(A, q, k, M)
    assert [r.id for r in res] == [f"art{i}" for i in er]
    assert [r.filename for r in res] == [f"file{i}.bin" for i in er]
    assert np.array_equal(np.array([r.similarity for r in res], F), es)
    # similar(): k + 1 neighbours of the artifact's own embedding minus the artifact itself
    sim = eng.blob_similar("art7", 5)
    er, es = oc.search(A, A[7], 6, M)
    exp = [int(i) for i in er if i != 7][:5]
    assert [r.id for r in sim] == [f"art{i}" for i in exp] and len(sim) == 5
    with pytest.raises(E.VectorError) as e:
        eng.blob_similar("nope", 3)
    assert e.value.kind == "NotFound"
    # no validation in the reference: empty query / k == 0 find nothing
    assert eng.blob_search_by_embedding([], 5) == [] and eng.blob_search_by_embedding(q, 0) == []
    # overwrite and delete keep the mirror current
    A[3] = q
    eng.blob_set_embedding("art3", "renamed.bin", q)
    eng.blob_remove("art4")
    res = eng.blob_search_by_embedding(q, 3)
    assert res[0].id == "art3" and res[0].filename == "renamed.bin" and res[0].similarity == 1.0
    keep = np.ones(n, bool)
    keep[4] = False
    er, es = oc.search(A, q, 3, M, mask=oc.mask_from_bool(keep))
    assert [r.id for r in res] == [f"art{i}" for i in er]
