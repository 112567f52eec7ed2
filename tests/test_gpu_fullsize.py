"""Full-size parity: BASELINE.json configs 2, 3 and 5 at their REAL sizes, HIP path (through the C ABI) against the CPU
oracle on a host twin of the same synthetic corpus — rows identical, scores bit-equal.

    config 2   1M x 768  cosine TOP-100, one query            (+ planted near-duplicates of the query)
    config 3   10M x 768 cosine TOP-100, 64 queries per call  (the matrix-core sweep over the bf16 mirror)
    config 4   80M x 768 row-range sharded 8 ways: the FULL row count on ONE 288-GB GPU (eight logical shards of 10M rows)
    config 5   10M x 1536 Euclidean TOP-1000 with a WHERE-predicate bitmap, selectivity 1.0 / 0.5 / 0.1

The oracle (oracle/nmn_oracle.c, reference-order arithmetic of vector_engine/src/lib.rs:2049-2101, 2231-2266 and
tensor_store/src/hnsw.rs:168-229) runs on all host cores; the GPU box has 256 threads and 3 TB of RAM, so a 10M x 768
query is ~0.25 s and the 61 GB twin of config 5 fits.  Smaller hosts skip what their RAM cannot hold.
"""
import os

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu

CORES = os.cpu_count() or 1
NO_ROW = np.uint64(0xFFFFFFFFFFFFFFFF)


def _host_ram_gb():
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
    except (ValueError, OSError):
        return 0.0


def _need_ram(gb):
    if _host_ram_gb() < gb:
        pytest.skip(f"host twin needs {gb} GB of RAM, this host has {_host_ram_gb():.0f}")


def _oracle(A, q, k, metric, mask=None, literal=False):
    """literal: the reference's own form — score every row, sort all N, truncate (lib.rs:2026-2034; one thread sorts 10M
    hits in ~2 s); otherwise the per-thread partial top-k + merge, pinned to the same answers on the golden fixtures
    (tests/test_golden_oracle.py)."""
    return oc.search(A, q, k, metric, mask=mask, nthreads=CORES, partial=not literal, native=True)


def _check_query(rows, scores, counts, qi, er, es):
    c = er.size
    assert counts[qi] == c, (qi, counts[qi], c)
    assert np.array_equal(rows[qi, :c], er), (qi, np.flatnonzero(rows[qi, :c] != er)[:5])
    assert np.array_equal(scores[qi, :c].view(np.uint32), es.view(np.uint32)) or np.all(scores[qi, :c] == es), \
        (qi, float(np.abs(scores[qi, :c] - es).max()))
    assert np.all(rows[qi, c:] == NO_ROW) and np.all(np.isneginf(scores[qi, c:]))


def test_config2_1Mx768_cosine_top100_single_query():
    """BASELINE config 2 (seeds 0x5eed0001 / 0x5eed0002, 16 planted near-duplicates of 4 queries)."""
    from neumann_amd import GpuFlatIndex
    n, d, k = 1_000_000, 768, 100
    A = oc.synth(0x5EED0001, 0, n, d, nthreads=CORES)
    Q = oc.synth(0x5EED0002, 0, 8, d)
    rng = np.random.default_rng(0x5EED)
    planted = {}
    for qi in range(4):  # near-duplicates: the query scaled (cosine-identical up to rounding) and perturbed in a few places
        for j in range(16):
            row = int(rng.integers(0, n))
            v = Q[qi] * np.float32(1.0 + 0.25 * j)
            v[rng.integers(0, d, size=j)] += np.float32(1e-3)
            planted[row] = v.astype(np.float32)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x5EED0001, n)
        for row, v in planted.items():
            idx.set_row(row, v)
            A[row] = v
        want = [_oracle(A, Q[qi], k, 0, literal=True) for qi in range(8)]
        want_m = {metric: _oracle(A, Q[5], k, metric, literal=True) for metric in (1, 2)}
        # the library's default (8-bit mirror), the bf16 mirror, and — the sweep SURVEY §8(d) prices and bench.py's c2_f32_* quotes —
        # the ROW-MAJOR F32 CORPUS through the LDS-DMA ring (nmn_scan_ring.hip): every list of every sweep is the oracle's
        for mode, nbytes, sweep in ((1, 1, "valu_i8"), (2, 2, "valu_bf16"), (0, 4, "ring_f32")):
            idx.set_mirror(mode)
            for qi in range(8):
                rows, scores, counts, st = idx.search(Q[qi], k, 0, with_stats=True)
                _check_query(rows, scores, counts, 0, *want[qi])
                assert st.fallback_queries == 0
                assert st.bytes_scanned == n * d * nbytes and st.sweep == sweep, (mode, st.bytes_scanned, st.sweep)
            # Euclidean and dot product over the same resident corpus
            for metric in (1, 2):
                rows, scores, counts, st = idx.search(Q[5], k, metric, with_stats=True)
                _check_query(rows, scores, counts, 0, *want_m[metric])
                assert st.bytes_scanned == n * d * nbytes and st.sweep == sweep and st.fallback_queries == 0, (mode, metric, st.sweep)


@pytest.fixture(scope="module")
def corpus_10Mx768():
    _need_ram(48)
    n, d = 10_000_000, 768
    return oc.synth(0x5EED0003, 0, n, d, nthreads=CORES)


def test_config3_10Mx768_cosine_top100_batch64(corpus_10Mx768):
    """BASELINE config 3: 64 queries per call — one matrix-core sweep of the 8-bit mirror, of the bf16 mirror it replaced
    (nmn_index_set_mirror(2)) and of the f32 rows themselves (nmn_index_set_mirror(0)), every candidate re-scored from the f32
    corpus.  All 64 lists of all three sweeps are compared with the oracle."""
    from neumann_amd import GpuFlatIndex
    A = corpus_10Mx768
    n, d = A.shape
    k, nq = 100, 64
    Q = oc.synth(0x5EED0002, 1000, nq, d)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x5EED0003, n)
        rows, scores, counts, st = idx.search(Q, k, 0, with_stats=True)
        assert st.bytes_scanned == n * d, "a 64-query batch sweeps the 8-bit mirror"
        idx.set_mirror(2)
        rows2, scores2, counts2, st2 = idx.search(Q, k, 0, with_stats=True)
        assert st2.bytes_scanned == n * d * 2, "nmn_index_set_mirror(2): the bf16 mirror"
        assert np.array_equal(rows2, rows) and np.array_equal(scores2.view(np.uint32), scores.view(np.uint32)) and np.array_equal(counts2, counts)
        for qi in range(nq):
            er, es = _oracle(A, Q[qi], k, 0, literal=qi in (0, 63))
            _check_query(rows, scores, counts, qi, er, es)
        # config 3 as SURVEY §8(d) prices it: the 64-query batch over the ROW-MAJOR F32 CORPUS (nmn_index_set_mirror(0): the
        # matrix-core sweep reads the f32 rows, rounds them to bf16 in registers) — all 64 lists are the mirror sweeps' = the oracle's
        idx.set_mirror(0)
        rows0, scores0, counts0, st0 = idx.search(Q, k, 0, with_stats=True)
        assert st0.bytes_scanned == n * d * 4 and st0.fallback_queries == 0, "nmn_index_set_mirror(0): the f32 corpus, one sweep per batch"
        assert np.array_equal(rows0, rows) and np.array_equal(scores0.view(np.uint32), scores.view(np.uint32)) and np.array_equal(counts0, counts)
        # the headline configuration on the same corpus: one query per call — over the 8-bit mirror (the default), the bf16 mirror
        # and the f32 corpus
        for mode, nbytes in ((1, 1), (2, 2), (0, 4)):
            idx.set_mirror(mode)
            for qi in (0, 63):
                r1, s1, c1, st1 = idx.search(Q[qi], k, 0, with_stats=True)
                assert st1.bytes_scanned == n * d * nbytes
                assert np.array_equal(r1[0], rows[qi]) and np.array_equal(s1[0].view(np.uint32), scores[qi].view(np.uint32))


def test_config4_shape_eight_shards_of_10Mx768_merge_to_unsharded(corpus_10Mx768):
    """The 8-way row-range split of the same 10M rows (config 4's partitioning at 1/8 scale per shard), one shard after the
    other on this GPU, merged with nmn_merge_topk_host == the oracle's answer over the whole corpus."""
    from neumann_amd import GpuFlatIndex
    from neumann_amd.flat_index import merge_topk_host
    from neumann_amd.sharded import shard_range
    A = corpus_10Mx768
    n, d = A.shape
    k, world = 100, 8
    q = oc.synth(0x5EED0002, 2000, 1, d)[0]
    lists_r, lists_s, lists_c = [], [], []
    for rank in range(world):
        r0, r1 = shard_range(n, world, rank)
        with GpuFlatIndex(d, r1 - r0, row_base=r0) as idx:
            idx.fill_synthetic(0x5EED0003, r1 - r0)
            rr, ss, cc = idx.search(q, k, 0)
        lists_r.append(rr)
        lists_s.append(ss)
        lists_c.append(cc)
    mr, ms, mc = merge_topk_host(np.stack(lists_r), np.stack(lists_s), np.stack(lists_c), k)
    er, es = _oracle(A, q, k, 0)
    _check_query(mr, ms, mc, 0, er, es)


def _device_memory_gb():
    import torch
    free, total = torch.cuda.mem_get_info(0)
    return free / 2**30, total / 2**30


@pytest.mark.config4_full
def test_config4_80Mx768_eight_shards_of_10M_rows_on_one_gpu():
    """BASELINE config 4 at its FULL row count: 80M x 768 f32 (245.8 GB) as eight row-range shards of 10M rows — shard g owns the
    global rows [g * 10M, (g + 1) * 10M), SURVEY §8(e) — behind the one-process handle (nmn_sharded_*), all eight on device 0
    of a 288-GB MI355X (one GPU per shard is the driver's 8-GPU run; the partitioning, the per-shard pipelines, the packed
    gather and merge_top_k — query_router/src/distributed.rs:413-433 — are the same code).  The mirrors do not all fit beside
    246 GB of rows: shards that find no room decline theirs and sweep the f32 rows; the answers must not depend on it.
    Checked: (1) one query per call and a 64-query batch, TOP-100 cosine, every answer of the single queries and three of the
    batch certified exact by the product's reference-order kernels (corpus-wide rank count, bit-equal scores, order);
    (2) two single queries and two of the batch against the CPU oracle on a host twin built shard by shard (30.7 GB at a
    time) and merged with nmn_merge_topk_host; (3) rows planted in the LAST shard come back first, with global ids >= 70M."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import certificate
    from neumann_amd import GpuShardedIndex
    from neumann_amd._capi import GATHER_PEER
    from neumann_amd.flat_index import merge_topk_host
    free_gb, total_gb = _device_memory_gb()
    if free_gb < 262:
        pytest.skip(f"80M x 768 f32 needs ~250 GB of free HBM on one device; this one has {free_gb:.0f} of {total_gb:.0f} GB free")
    _need_ram(80)
    G, per, d, k, nq = 8, 10_000_000, 768, 100, 64
    n = G * per
    seed = 0x5EED0003
    Q1 = oc.synth(0x5EED0002, 4000, 4, d)
    QB = oc.synth(0x5EED0002, 5000, nq, d)
    # near-duplicates of query 0 in the last shard: a scaled copy (cosine 1 up to rounding) and two perturbed ones
    rng = np.random.default_rng(0x5EED0004)
    planted = {}
    for j, local in enumerate((9_999_999, 1_234_567, 64)):
        v = Q1[0] * np.float32(1.0 + 0.5 * j)
        v[rng.integers(0, d, size=3 * j)] += np.float32(2e-3)
        planted[local] = v.astype(np.float32)
    with GpuShardedIndex(d, n, G, devices=[0] * G, gather=GATHER_PEER) as sh:
        sh.fill_synthetic(seed, n)
        assert sh.rows == n and [sh.shard_rows(g) for g in range(G)] == [per] * G
        assert [sh.global_row(g, 0) for g in range(G)] == [g * per for g in range(G)]
        last = sh.shard(G - 1)
        for local, v in planted.items():
            last.set_row(local, v)
        hbm = [sh.shard(g).hbm_bytes() for g in range(G)]
        assert sum(h[0] for h in hbm) == n * d * 4
        shards = [sh.shard(g) for g in range(G)]

        # (1) one query per call
        singles = []
        elem_bytes = set()
        for qi in range(4):
            rows, scores, counts, st = sh.search(Q1[qi], k, 0, with_stats=True)
            assert counts[0] == k and st.rows_scanned == n
            elem_bytes.add(st.bytes_scanned / (n * d))
            c = certificate(shards, Q1[qi], 0, rows, scores, counts, 1, None)
            assert c["exact_topk_certified"], (qi, c)
            singles.append((rows.copy(), scores.copy(), counts.copy()))
        # the planted rows of the last shard lead query 0's list, under their GLOBAL ids
        top3 = set(int(r) for r in singles[0][0][0, :3])
        assert top3 == {(G - 1) * per + local for local in planted}, top3
        # 100 best of 80M uniformly spread rows: every list reaches into the last shard's range (P(miss) = (7/8)^100 ~ 1e-6)
        assert all(int(r[0].max()) >= 70_000_000 for r, _, _ in singles)

        # 64 queries per call (the matrix-core sweep on every shard: over its mirror where it kept one, else over its f32 rows — nmn_scan_mfma_f32.hip)
        rows_b, scores_b, counts_b, st_b = sh.search(QB, k, 0, with_stats=True)
        assert np.all(counts_b == k) and st_b.rows_scanned == n
        for qi in (0, 31, 63):
            c = certificate(shards, QB[qi], 0, rows_b[qi:qi + 1], scores_b[qi:qi + 1], counts_b[qi:qi + 1], 1, None)
            assert c["exact_topk_certified"], (qi, c)
        # the same lists from single-query calls (a different sweep on most shards): bit-equal
        r1, s1, c1 = sh.search(QB[31], k, 0)
        assert np.array_equal(r1[0], rows_b[31]) and np.array_equal(s1[0].view(np.uint32), scores_b[31].view(np.uint32))
        # which mirrors the shards ended up with is the library's business; that they do not all fit is this test's premise
        hbm_after = [sh.shard(g).hbm_bytes() for g in range(G)]
        with_mirror = sum(1 for h in hbm_after if h[1] > 0)
        assert with_mirror < G, "eight 10M x 768 shards with mirrors cannot fit 288 GB: somebody must have declined"
        print(f"config 4 on one GPU: {with_mirror} of {G} shards hold a mirror; bytes/element swept by the single queries: "
              f"{sorted(elem_bytes)}; HBM {sum(sum(h) for h in hbm_after) / 2**30:.1f} GiB in shards")

    # (2) the oracle, shard by shard on a host twin of 10M rows at a time, merged as ResultMerger::merge_top_k does
    checks = [("single", 0, Q1[0]), ("single", 1, Q1[1]), ("batch", 0, QB[0]), ("batch", 63, QB[63])]
    per_shard = {i: ([], [], []) for i in range(len(checks))}
    for g in range(G):
        A = oc.synth(seed, g * per, per, d, nthreads=CORES)
        if g == G - 1:
            for local, v in planted.items():
                A[local] = v
        for i, (_, _, q) in enumerate(checks):
            er, es = oc.search(A, q, k, 0, nthreads=CORES, partial=True, native=True, row_base=g * per)
            rr = np.full(k, NO_ROW, dtype=np.uint64)
            ss = np.full(k, -np.inf, dtype=np.float32)
            rr[:er.size] = er
            ss[:er.size] = es
            per_shard[i][0].append(rr[None])
            per_shard[i][1].append(ss[None])
            per_shard[i][2].append(np.array([er.size], dtype=np.uint32))
        del A
    for i, (kind, qi, _) in enumerate(checks):
        mr, ms, mc = merge_topk_host(np.stack(per_shard[i][0]), np.stack(per_shard[i][1]), np.stack(per_shard[i][2]), k)
        got = singles[qi] if kind == "single" else (rows_b[qi:qi + 1], scores_b[qi:qi + 1], counts_b[qi:qi + 1])
        _check_query(got[0], got[1], got[2], 0, mr[0, :mc[0]], ms[0, :mc[0]])


def test_config5_10Mx1536_l2_top1000_masked():
    """BASELINE config 5: Euclidean TOP-1000 over 10M x 1536 with a selection bitmap of selectivity 1.0 / 0.5 / 0.1 / 0.02
    (relational_engine's layout: bit i of word i/64, LSB first) — on the library's default sweep (the 8-bit mirror), on the bf16
    mirror and on the ROW-MAJOR F32 CORPUS (nmn_index_set_mirror(0): the ring sweep without a bitmap, scan_kernel's survivor walk
    under one — the kernels bench.py's c5_mask*_f32_* figures time), each against the oracle (search_with_pre_filter's survivor
    scan, lib.rs:3514-3557).  The sweep reads the KEPT rows only: bytes_scanned = kept x dim x (1 | 2 | 4)."""
    from neumann_amd import GpuFlatIndex
    _need_ram(96)
    n, d, k = 10_000_000, 1536, 1000
    A = oc.synth(0x5EED0005, 0, n, d, nthreads=CORES)
    Q = oc.synth(0x5EED0002, 3000, 4, d)
    rng = np.random.default_rng(0x5EED0005)
    with GpuFlatIndex(d, n) as idx:
        idx.fill_synthetic(0x5EED0005, n)
        for qi, sel in enumerate((1.0, 0.5, 0.1, 0.02)):
            keep = None if sel >= 1.0 else rng.random(n) < sel
            mask = None if keep is None else oc.mask_from_bool(keep)
            kept = n if keep is None else int(keep.sum())
            er, es = _oracle(A, Q[qi], k, 1, mask=mask, literal=sel == 0.1)
            for mode, nbytes, sweeps in ((1, 1, ("valu_i8",)), (0, 4, ("ring_f32",) if keep is None else ("valu_f32",)), (2, 2, ("valu_bf16",))):
                idx.set_mirror(mode)
                rows, scores, counts, st = idx.search(Q[qi], k, 1, mask=mask, with_stats=True)
                _check_query(rows, scores, counts, 0, er, es)
                assert st.fallback_queries == 0, (sel, mode)
                assert st.rows_scanned == kept and st.bytes_scanned == kept * d * nbytes, (sel, mode, st.rows_scanned, st.bytes_scanned)
                assert st.sweep in sweeps, (sel, mode, st.sweep)
