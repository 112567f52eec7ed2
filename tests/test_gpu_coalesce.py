"""Concurrent host-buffer searches (`Arc<VectorEngine>` shared by many threads, query_router/src/lib.rs:710,
5615-5666): the shim merges callers that arrive while the shard is busy into one query batch.  Every caller must
get exactly what it gets alone — here: what the oracle says — whatever batch it rode in."""
import threading

import numpy as np
import pytest

from oracle import oracle_c as oc

pytestmark = pytest.mark.gpu
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hammer(idx, jobs, n_threads):
    """jobs: list of (query, k, metric, mask); every thread walks its share; returns results in job order."""
    out = [None] * len(jobs)
    errs = []
    start = threading.Barrier(n_threads)

    def work(t):
        try:
            start.wait()
            for j in range(t, len(jobs), n_threads):
                q, k, metric, mask = jobs[j]
                out[j] = idx.search(q, k, metric, mask=mask)
        except Exception as e:  # noqa: BLE001 - reported below
            errs.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    return out


@pytest.mark.parametrize("n,d,threads", [(20_000, 128, 32), (300_000, 512, 24)])  # two / one batches in flight
def test_concurrent_callers_get_their_own_exact_answers(n, d, threads):
    from neumann_amd import GpuFlatIndex
    A = oc.synth(91, 0, n, d)
    rng = np.random.default_rng(3)
    mask = rng.integers(0, 2**64, size=(n + 63) // 64, dtype=np.uint64)
    jobs = []
    for j in range(threads * 6):
        q = oc.synth(92, j, 1, d)[0]
        metric = (0, 0, 2, 1)[j % 4]                    # mostly the metrics that share the MFMA sweep
        k = (1, 10, 100, 37, 250)[j % 5]
        jobs.append((q, k, metric, mask if j % 7 == 3 else None))
    # (single_launch=False: a lone call on a small shard would otherwise finish in one kernel, ~25 us, before a second thread
    #  arrives — this test is about the calls that DO meet; whether they meet is timing, so the hammering is repeated if not)
    with GpuFlatIndex(d, n, single_launch=False) as idx:
        idx.upload(A)
        for attempt in range(5):
            got = _hammer(idx, jobs, threads)
            batches, merged = idx.coalesce_stats()
            if batches > 0:
                break
        assert batches > 0 and merged >= 2 * batches, (batches, merged)
        # every 8th job against the oracle (the oracle is slow), all of them against a single-threaded run
        for j, (q, k, metric, m) in enumerate(jobs):
            rows, scores, counts = got[j]
            r1, s1, c1 = idx.search(q, k, metric, mask=m)
            assert counts[0] == c1[0]
            assert np.array_equal(rows, r1) and np.array_equal(scores.view(np.uint32), s1.view(np.uint32)), j
            if j % 8 == 0:
                er, es = oc.search(A, q, k, metric, mask=m)
                c = er.size
                assert counts[0] == c
                assert np.array_equal(rows[0, :c], er) and np.all(scores[0, :c] == es)
                assert np.all(rows[0, c:] == U64_MAX)


def test_multi_query_calls_merge_too():
    from neumann_amd import GpuFlatIndex
    n, d = 50_000, 256
    A = oc.synth(93, 0, n, d)
    jobs = [(oc.synth(94, 3 * j, 3, d), 20 + j % 3, 0, None) for j in range(64)]
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        got = _hammer(idx, jobs, 16)
        for j, (Q, k, metric, _) in enumerate(jobs):
            rows, scores, counts = got[j]
            r1, s1, c1 = idx.search(Q, k, metric)
            assert np.array_equal(rows, r1) and np.array_equal(scores.view(np.uint32), s1.view(np.uint32))
            assert np.array_equal(counts, c1)
        er, es = oc.search(A, jobs[5][0][1], jobs[5][1], 0)
        assert np.array_equal(got[5][0][1, :er.size], er) and np.all(got[5][1][1, :er.size] == es)


def test_an_invalid_call_does_not_poison_its_neighbours():
    from neumann_amd import GpuFlatIndex, _capi
    n, d = 10_000, 64
    A = oc.synth(95, 0, n, d)
    with GpuFlatIndex(d, n) as idx:
        idx.upload(A)
        good = [(oc.synth(96, j, 1, d)[0], 5, 0, None) for j in range(40)]
        bad_seen = []

        def bad():
            for _ in range(20):
                try:
                    idx.search(np.zeros(d, np.float32), 0, 0)   # k == 0: rejected before it can join a batch
                except _capi.NeumannGpuError as e:
                    bad_seen.append(e.status)

        t = threading.Thread(target=bad)
        t.start()
        got = _hammer(idx, good, 8)
        t.join()
        assert bad_seen and all(c == _capi.ERR_INVALID_TOP_K for c in bad_seen)
        for j, (q, k, metric, _) in enumerate(good):
            er, es = oc.search(A, q, k, metric)
            assert np.array_equal(got[j][0][0, :er.size], er) and np.all(got[j][1][0, :er.size] == es)


def test_writers_drain_the_queue_and_hand_the_slot_back():
    """set_row / upload wait for the running batch, keep new ones from starting, and restart the queue when done.  The
    writer re-writes rows with their own content, so every answer must stay what it was."""
    from neumann_amd import GpuFlatIndex
    n, d = 60_000, 256
    A = oc.synth(97, 0, n, d)
    jobs = [(oc.synth(98, j, 1, d)[0], 10 + j % 4, (0, 1, 2)[j % 3], None) for j in range(24 * 8)]
    with GpuFlatIndex(d, n + 1000) as idx:
        idx.upload(A)
        want = [idx.search(q, k, m) for q, k, m, _ in jobs]
        stop = threading.Event()
        writes = [0]

        def writer():
            r = 0
            while not stop.is_set():
                idx.set_row(r % n, A[r % n])
                if r % 50 == 0:
                    idx.upload(A[n - 16:], row0=n - 16)     # overwrite the tail in place
                r += 1
                writes[0] += 1

        t = threading.Thread(target=writer)
        t.start()
        try:
            got = _hammer(idx, jobs, 24)
        finally:
            stop.set()
            t.join()
        assert writes[0] > 0
        for (r0, s0, c0), (r1, s1, c1) in zip(want, got):
            assert np.array_equal(r0, r1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and c0[0] == c1[0]


@pytest.mark.parametrize("n,d,mirror", [(40_000, 768, 1), (30_000, 256, 1), (25_000, 200, 1),  # matrix-core sweep from 3 / 5 queries / never
                                        (40_000, 768, 0), (30_000, 256, 0)])  # ... over the f32 rows themselves (no mirror, round 5)
def test_differently_filtered_callers_share_a_sweep(n, d, mirror):
    """Concurrent searches whose WHERE bitmaps differ (each in device memory, as the predicate kernel leaves them)
    ride one matrix-core sweep that reads one bitmap per query; where that sweep does not apply they must still
    come out right (query by query)."""
    import torch
    from neumann_amd import GpuFlatIndex
    A = oc.synth(101, 0, n, d)
    rng = np.random.default_rng(12)
    words = (n + 63) // 64
    n_masks = 12
    host_masks = [oc.mask_from_bool(rng.random(n) < sel) for sel in np.linspace(0.02, 0.9, n_masks)]
    host_masks[3][:] = 0                                    # a filter nothing passes
    dev = torch.device("cuda:0")
    dev_masks = [torch.from_numpy(m[:words].view(np.int64).copy()).to(dev) for m in host_masks]
    torch.cuda.synchronize()
    jobs = []
    for j in range(24 * 5):
        mi = j % (n_masks + 1)                              # the last one: no filter at all
        jobs.append((oc.synth(102, j, 1, d)[0], (5, 40, 100)[j % 3], (0, 1, 2)[j % 3 if d != 200 else 0], mi))
    with GpuFlatIndex(d, n, single_launch=False) as idx:   # (see test_concurrent_callers_get_their_own_exact_answers)
        idx.set_mirror(mirror)
        idx.upload(A)
        out = [None] * len(jobs)
        errs = []

        def work(t, start):
            try:
                start.wait()
                for j in range(t, len(jobs), 24):
                    q, k, metric, mi = jobs[j]
                    if mi == n_masks:
                        out[j] = idx.search(q, k, metric)
                    else:
                        out[j] = idx.search_dmask(q, k, metric, dev_masks[mi].data_ptr())
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        for attempt in range(5):   # whether calls meet is timing: hammer again if none did
            start = threading.Barrier(24)
            threads = [threading.Thread(target=work, args=(t, start)) for t in range(24)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            assert not errs, errs
            batches, merged = idx.coalesce_stats()
            if batches > 0:
                break
        assert batches > 0
        for j, (q, k, metric, mi) in enumerate(jobs):
            rows, scores, counts = out[j]
            er, es = oc.search(A, q, k, metric, mask=None if mi == n_masks else host_masks[mi])
            c = er.size
            assert counts[0] == c, (j, mi, counts[0], c)
            assert np.array_equal(rows[0, :c], er), (j, mi)
            assert np.all(scores[0, :c] == es), (j, mi)
            assert np.all(rows[0, c:] == U64_MAX)
        if mirror == 0:
            assert idx.hbm_bytes()[1] == 0, "a mirror was built under set_mirror(0)"


@pytest.mark.parametrize("n,d", [(30_000, 768), (40_000, 128)])
def test_chaos_of_concurrent_call_kinds_is_batch_invariant(n, d):
    """Every kind of host-buffer search at once — three metrics, k from 1 to 300, 1-3 queries per call, no filter, one
    shared host bitmap, several device bitmaps, predicates — from 32 threads, with a writer re-writing rows in place.
    Whatever batches form, each call must return what the same call returns alone afterwards (and the oracle agrees on
    a sample)."""
    import torch
    from neumann_amd import GpuFlatIndex
    from neumann_amd import columns as g
    A = oc.synth(201, 0, n, d)
    rng = np.random.default_rng(77)
    words = (n + 63) // 64
    host_mask = oc.mask_from_bool(rng.random(n) < 0.4)
    dev_bool = [rng.random(n) < s for s in (0.05, 0.5, 0.95)]
    dev_host = [oc.mask_from_bool(b) for b in dev_bool]
    dev = torch.device("cuda:0")
    dev_masks = [torch.from_numpy(m[:words].view(np.int64).copy()).to(dev) for m in dev_host]
    torch.cuda.synchronize()
    bucket = np.arange(n) % 7
    jobs = []
    for j in range(32 * 12):
        kind = int(rng.integers(0, 4))
        nq = int(rng.choice([1, 1, 1, 2, 3]))
        Q = oc.synth(202, 3 * j, nq, d)
        k = int(rng.choice([1, 7, 40, 100, 300]))
        metric = int(rng.integers(0, 3))
        jobs.append((kind, Q, k, metric, int(rng.integers(0, 3)), int(rng.integers(0, 7))))
    with GpuFlatIndex(d, n) as idx, g.GpuColumns(n) as gc:
        idx.upload(A)
        c = gc.add_column()
        gc.write(c, 0, np.full(n, g.CELL_INT, np.uint8), bucket.astype(np.uint64))
        gc.write_valid(0, np.full(words, 0xFFFFFFFFFFFFFFFF, np.uint64))

        def run(job):
            kind, Q, k, metric, mi, b = job
            if kind == 0:
                return idx.search(Q, k, metric)
            if kind == 1:
                return idx.search(Q, k, metric, mask=host_mask)
            if kind == 2:
                return idx.search_dmask(Q, k, metric, dev_masks[mi].data_ptr())
            return idx.search_pred(gc, [(g.PRED_CMP, g.CMP_EQ, g.CELL_INT, c, b, 0)], [], Q, k, metric)[:3]

        out = [None] * len(jobs)
        errs = []
        stop = threading.Event()
        start = threading.Barrier(33)

        def work(t):
            try:
                start.wait()
                for j in range(t, len(jobs), 32):
                    out[j] = run(jobs[j])
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def writer():
            start.wait()
            r = 0
            while not stop.is_set():
                idx.set_row(r % n, A[r % n])
                r += 97

        th = [threading.Thread(target=work, args=(t,)) for t in range(32)] + [threading.Thread(target=writer)]
        for x in th:
            x.start()
        for x in th[:-1]:
            x.join()
        stop.set()
        th[-1].join()
        assert not errs, errs
        for j, job in enumerate(jobs):
            want = run(job)
            for a, b_ in zip(want, out[j]):
                assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b_).view(np.uint8)), (j, job[0], job[2], job[3])
            if j % 16 == 0:
                kind, Q, k, metric, mi, b = job
                m = None if kind == 0 else host_mask if kind == 1 else dev_host[mi] if kind == 2 else oc.mask_from_bool(bucket == b)
                er, es = oc.search(A, Q[0], k, metric, mask=m)
                assert np.array_equal(out[j][0][0, :er.size], er) and np.all(out[j][1][0, :er.size] == es), j
