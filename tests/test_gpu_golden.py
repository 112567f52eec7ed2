"""GPU parity against the committed golden fixtures, through the C ABI (host-buffer API)."""
import numpy as np
import pytest

from tests import _golden as G

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _assert_result(rows, scores, counts, er, es, k):
    c = er.size
    assert counts == c
    assert np.array_equal(rows[:c], er)
    assert np.all(scores[:c] == es)                 # exact-rescored scores (== treats -0.0/+0.0 alike)
    assert np.all(rows[c:] == U64_MAX) and np.all(np.isneginf(scores[c:]))


def test_small_explicit():
    from neumann_amd import GpuFlatIndex
    from oracle import oracle_c as oc
    g = G.load("small_explicit.npz")
    A, Q = g["A"], g["Q"]
    with GpuFlatIndex(A.shape[1], A.shape[0]) as idx:
        idx.upload(A)
        for m in (0, 1, 2):
            for tag, keep in (("all", None), ("k50", g["keep50"]), ("k10", g["keep10"])):
                mask = None if keep is None else oc.mask_from_bool(keep)
                rows, scores, counts = idx.search(Q, 10, m, mask=mask)      # all 4 queries in one call
                for qi in range(4):
                    _assert_result(rows[qi], scores[qi], counts[qi], g[f"rows_m{m}_q{qi}_{tag}"],
                                   g[f"scores_m{m}_q{qi}_{tag}"], 10)


def test_example_vector_search():
    from neumann_amd import GpuFlatIndex
    g = G.load("example_vector_search.npz")
    with GpuFlatIndex(8, 8) as idx:
        idx.upload(g["A"])
        for m in (0, 1, 2):
            rows, scores, counts = idx.search(g["Q"], 3, m)
            for qi in range(3):
                _assert_result(rows[qi], scores[qi], counts[qi], g[f"rows_m{m}_q{qi}"], g[f"scores_m{m}_q{qi}"], 3)


@pytest.mark.parametrize("name", G.SYNTH_FILES)
@pytest.mark.parametrize("device_fill", [False, True])
def test_synth_fixture(name, device_fill):
    from neumann_amd import GpuFlatIndex, synth_rows
    g = G.load(name)
    n, dim, k = int(g["n"]), int(g["dim"]), int(g["k"])
    with GpuFlatIndex(dim, n) as idx:
        if device_fill:   # generator on the GPU + planted rows overwritten one by one
            idx.fill_synthetic(int(g["seed"]), n)
            for i, v in zip(g["planted_idx"], g["planted"]):
                idx.set_row(int(i), v)
        else:             # host copy of the same corpus uploaded through the C ABI
            idx.upload(G.rebuild_corpus(g, synth_rows))
        for m, qi, tag, mask, er, es in G.synth_cases(g):
            rows, scores, counts = idx.search(g["Q"][qi], k, m, mask=mask)
            _assert_result(rows[0], scores[0], counts[0], er, es, k)


def test_sharded_indexes_merge_to_unsharded():
    """8 row-range shards on one GPU, device merge kernel == unsharded golden (distributed.rs:413-433)."""
    import torch
    from neumann_amd import (GpuFlatIndex, merge_topk_device, merge_topk_device_packed, merge_topk_host,
                             packed_layout, synth_rows)
    g = G.load("synth_4096x768_top100.npz")
    A = G.rebuild_corpus(g, synth_rows)
    k, S = 100, 8
    per = (A.shape[0] + S - 1) // S
    shards = []
    for s in range(S):
        idx = GpuFlatIndex(A.shape[1], per, row_base=s * per)
        idx.upload(A[s * per:(s + 1) * per])
        shards.append(idx)
    try:
        for m in (0, 1, 2):
            R = np.empty((S, 1, k), dtype=np.uint64)
            Sc = np.empty((S, 1, k), dtype=np.float32)
            C = np.empty((S, 1), dtype=np.uint32)
            for s, idx in enumerate(shards):
                R[s], Sc[s], C[s] = idx.search(g["Q"][0], k, m)
            hr, hs, hc = merge_topk_host(R, Sc, C, k)
            dr, ds, dc = merge_topk_device(torch.from_numpy(R.view(np.int64)).cuda(),
                                           torch.from_numpy(Sc).cuda(),
                                           torch.from_numpy(C.view(np.int32)).cuda(), k)
            torch.cuda.synchronize()
            # the same S blocks packed end to end as ONE all-gather would deliver them (sharded.py)
            size, off_s, off_c = packed_layout(1, k)
            packed = np.zeros(S * size, dtype=np.uint8)
            for sh in range(S):
                blk = packed[sh * size:(sh + 1) * size]
                blk[:off_s] = R[sh].view(np.uint8).reshape(-1)
                blk[off_s:off_c] = Sc[sh].view(np.uint8).reshape(-1)
                blk[off_c:off_c + 4] = C[sh].view(np.uint8).reshape(-1)
            pr, ps, pc = merge_topk_device_packed(torch.from_numpy(packed).cuda(), S, 1, k)
            torch.cuda.synchronize()
            for rr, ss, cc in ((hr, hs, hc), (dr.cpu().numpy().view(np.uint64), ds.cpu().numpy(), dc.cpu().numpy()),
                               (pr.cpu().numpy().view(np.uint64), ps.cpu().numpy(), pc.cpu().numpy())):
                assert cc[0] == k
                assert np.array_equal(rr[0], g[f"rows_m{m}_q0_all"])
                assert np.all(ss[0] == g[f"scores_m{m}_q0_all"])
    finally:
        for idx in shards:
            idx.close()


# ---- widened rows against their committed fixtures (tests/golden/make_golden_next.py) ------------------------
def test_filters_golden_through_the_predicate_kernel():
    from neumann_amd import engine as E
    from tests.test_gpu_filter import to_fc
    rows, cases = G.load_filters()
    n, d = len(rows), 8
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, d)).astype(np.float32)
    eng = E.VectorEngine()
    for i, meta in enumerate(rows):
        eng.store_embedding_with_metadata(f"k{i}", A[i], meta)
    q = rng.standard_normal(d).astype(np.float32)
    for cond, selected in cases:
        res = eng.search_similar_filtered(q, n, to_fc(E, cond), E.FilteredSearchConfig.pre_filter())
        assert sorted(int(r.key[1:]) for r in res) == selected, cond       # top_k = n: exactly the selected rows
    assert eng.device_filter_evals() == len(cases)


def test_ivf_golden_on_the_gpu():
    from neumann_amd.ivf import GpuIvfFlat
    g = G.load("ivf_flat_small.npz")
    V, Q = g["V"], g["Q"]
    for tag in ("rnd", "pp"):
        with GpuIvfFlat(g[f"centroids_{tag}"], capacity_rows=len(V), nprobe=3) as ivf:
            assert np.array_equal(ivf.add(V), g[f"assign_{tag}"])
            for qi in range(6):
                for nprobe in (1, 3, 10):
                    ids, dist, counts = ivf.search(Q[qi], 15, nprobe)
                    exp = g[f"ids_{tag}_q{qi}_p{nprobe}"]
                    assert counts[0] == exp.size and np.array_equal(ids[0, :exp.size], exp)
                    assert np.array_equal(dist[0, :exp.size], g[f"dist_{tag}_q{qi}_p{nprobe}"])


def test_sparse_cos64_golden_on_the_gpu():
    from neumann_amd import GpuFlatIndex
    g = G.load("sparse_cos64_small.npz")
    A, Q = g["A"], g["Q"]
    with GpuFlatIndex(A.shape[1], A.shape[0]) as idx:
        idx.upload(A)
        rows, scores, counts = idx.search(Q, 20, 3)
        for qi in range(5):
            assert counts[qi] == 20 and np.array_equal(rows[qi], g[f"rows_q{qi}"])
            assert np.array_equal(scores[qi], g[f"scores_q{qi}"])


def test_engine_kmeans_matches_golden_centroids():
    """The C++ k-means port (nmn_engine.cpp) reproduces the committed centroids bit for bit."""
    from neumann_amd import engine as E
    g = G.load("ivf_flat_small.npz")
    V = g["V"]
    eng = E.VectorEngine()
    eng.batch_store_embeddings([f"k{i:04d}" for i in range(len(V))], V)
    for init, tag in (("random", "rnd"), ("kmeans++", "pp")):
        index, keys = eng.build_ivf_index(E.IVFBuildOptions(num_clusters=10, nprobe=3, max_iterations=8,
                                                            convergence_threshold=1e-4, seed=4242, init_method=init))
        assert keys == [f"k{i:04d}" for i in range(len(V))]       # list_keys() order = insertion order here
        assert np.array_equal(index.centroids(V.shape[1]), g[f"centroids_{tag}"])
