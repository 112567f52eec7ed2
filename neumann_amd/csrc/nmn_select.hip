// nmn_select.hip — top-k selection around the scan: threshold pick + candidate collection, the
// final (exact score, row) sort, the exact-fallback selection and the multi-shard merge.
//
// Replaces `results.sort_by(score desc); results.truncate(k)` over ALL N results
// (vector_engine/src/lib.rs:2027-2034, 2093-2100) and `ResultMerger::merge_top_k`
// (query_router/src/distributed.rs:413-433).  Order everywhere: score descending, ties by ascending
// row id; NaN scores last.
#include <algorithm>
#include <type_traits>

#include "nmn_select_dev.h"

namespace nmn {

#ifndef NMN_SELECT_VR
#define NMN_SELECT_VR 16
#endif
constexpr uint32_t kListCap = 4096;  // LDS work lists (waves, tiles); both bounded by kMaxScanWaves / cand_cap

// The same two digits with four barriers instead of ten: two padded histograms (hA, hB: kHistPad words each, any LDS nobody else is
// using during the call), every wave picks the crossing bins for itself (wave_pick).  On return both may be overwritten.
// MEASURED SLOWER in select_kernel (profiles/r05i_select_phases.txt: a pick over <= 4096 keys 4.1 -> 4.7-6.0 us): the workgroup is
// 16 waves on 4 SIMDs, so what every wave does redundantly costs 4 x its instructions of SIMD time — wave_pick's ~200 VALU
// instructions (32 bins per lane) outweigh the six barriers saved.  Phases of this kernel are instruction-bound, not barrier-bound.
// Kept for sample_bound_kernel's register form only (a lone launch between two sweeps, where it measured the same).
template <typename KeyAt>
__device__ uint32_t radix2w(KeyAt key_at, uint32_t n, uint32_t kk, uint32_t* hA, uint32_t* hB) {
    const uint32_t tid = threadIdx.x;
    constexpr int V = 8;
    for (uint32_t b = tid; b < kHistPad; b += kSelThreads) {
        hA[b] = 0;
        hB[b] = 0;
    }
    __syncthreads();
    for (uint32_t e0 = tid; e0 < n; e0 += kSelThreads * V) {
        uint32_t kv[V];
#pragma unroll
        for (int u = 0; u < V; u++) {
            const uint32_t e = e0 + (uint32_t)u * kSelThreads;
            kv[u] = e < n ? key_at(e) : kKeyMasked;
        }
#pragma unroll
        for (int u = 0; u < V; u++) hist_add_wave(hA, kv[u] != kKeyMasked, hpad(kv[u] >> 21));
    }
    __syncthreads();
    const PickResult p1 = wave_pick(hA, kk);
    for (uint32_t e0 = tid; e0 < n; e0 += kSelThreads * V) {
        uint32_t kv[V];
#pragma unroll
        for (int u = 0; u < V; u++) {
            const uint32_t e = e0 + (uint32_t)u * kSelThreads;
            kv[u] = e < n ? key_at(e) : kKeyMasked;
        }
#pragma unroll
        for (int u = 0; u < V; u++) hist_add_wave(hB, kv[u] != kKeyMasked && (kv[u] >> 21) == p1.bin, hpad((kv[u] >> 10) & 2047u));
    }
    __syncthreads();
    const PickResult p2 = wave_pick(hB, kk - p1.above);
    __syncthreads();  // every wave is done with the histograms
    return (p1.bin << 21) | (p2.bin << 10);
}

// count of valid (non-zero) gathered keys
template <typename KeyAt>
__device__ uint32_t count_valid(KeyAt key_at, uint32_t n, uint32_t* s_word) {
    const uint32_t tid = threadIdx.x;
    constexpr int V = 8;
    if (tid == 0) *s_word = 0;
    __syncthreads();
    uint32_t loc = 0;
    for (uint32_t e0 = tid; e0 < n; e0 += kSelThreads * V) {
        uint32_t kv[V];
#pragma unroll
        for (int u = 0; u < V; u++) {
            const uint32_t e = e0 + (uint32_t)u * kSelThreads;
            kv[u] = e < n ? key_at(e) : kKeyMasked;
        }
#pragma unroll
        for (int u = 0; u < V; u++) loc += kv[u] != kKeyMasked;
    }
    if (loc) atomicAdd(s_word, loc);
    __syncthreads();
    return *s_word;
}

// ---- main-path selection: one workgroup per query ---------------------------------------------
// The scan left a three-level maximum hierarchy: scores[row] -> tmax[64-row tile] -> wmax[scan wave =
// `tiles_per_wave` consecutive tiles] (<= 4096 entries, kept in LDS here).  The k-th largest key of ANY
// level is a lower bound on the k-th best approximate score (k groups of that level each hold a row at
// least that good), and everything below a valid bound can be dropped before looking one level down.
// Top-down refinement, all lists compacted in LDS:
//   Tw = k-th largest wave maximum                  (LDS only)
//   LT = tiles >= Tw-margin inside waves >= Tw-margin, T2 = k-th largest of LT   (reads those waves' tmax)
//   LR = rows  >= T2-margin inside tiles >= T2-margin, T3 = k-th largest of LR   (reads those tiles' scores)
//   candidates = LR entries >= T3-margin  — i.e. k rows plus those within the rounding margin.
// Expected list sizes are k(1+small) at every level, whatever the shard size; global reads at 10M rows,
// k = 100: 16 KB wmax + ~100 x 39 tile maxima + ~105 x 256 B of scores.  If a compact list overflows
// (more than 8192 groups within the margin: heavy ties, or fewer valid waves than k with many valid
// tiles) the generic three-level code below takes over.
// ---- sampling bound of the batched sweep -------------------------------------------------------
// tmax_sample[q][i] = maximum key of sampled tile i (every S-th tile).  If k sampled tiles reach T, k rows
// of the corpus reach T, so no row scoring below T - margin can be in the top-k: the main sweep only
// writes the scores of tiles whose maximum reaches skip_key[q] = key(score(T) - margin).
__global__ void __launch_bounds__(kSelThreads) sample_bound_kernel(const uint32_t* __restrict__ tmax_sample,
                                                                   uint64_t stride, uint32_t n_sample,
                                                                   const QInfo* __restrict__ qinfo, uint32_t k,
                                                                   uint32_t* __restrict__ skip_key, int combine_max) {
    __shared__ uint32_t hist[kHistPad], hist2[kHistPad];
    __shared__ PickResult pick;
    __shared__ uint32_t s_cnt;
    const uint32_t q = blockIdx.x;
    const uint32_t* keys = tmax_sample + (uint64_t)q * stride;
    uint32_t skip = kKeyNaN;
    constexpr int NV = 40;  // keys per thread the register form holds
    if (n_sample <= (uint32_t)kSelThreads * NV) {
        // Up to 40 960 keys: ONE round trip to memory — every key of the query is loaded into registers at once, and the count and
        // both radix digits work from there.  (The generic form below reads the keys three times, eight per thread and trip: the
        // refinement between the two launches of a batched sweep — 39 000 tile maxima per query at 10M rows, nothing else on the
        // device while it runs — took 110 us of a 1.5 ms batch.)
        const uint32_t tid = threadIdx.x;
        uint32_t kv[NV];
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const uint32_t e = tid + (uint32_t)u * kSelThreads;
            kv[u] = e < n_sample ? keys[e] : kKeyMasked;
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        uint32_t loc = 0;
#pragma unroll
        for (int u = 0; u < NV; u++) loc += kv[u] != kKeyMasked;
        if (loc) atomicAdd(&s_cnt, loc);
        __syncthreads();
        if (s_cnt >= k) {  // (block-uniform)
            // two padded histograms, every wave picks for itself (nmn_select_dev.h): three barriers for both digits
            for (uint32_t b = tid; b < kHistPad; b += kSelThreads) {
                hist[b] = 0;
                hist2[b] = 0;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NV; u++) hist_add_wave(hist, kv[u] != kKeyMasked, hpad(kv[u] >> 21));
            __syncthreads();
            const PickResult p1 = wave_pick(hist, k);
#pragma unroll
            for (int u = 0; u < NV; u++) hist_add_wave(hist2, kv[u] != kKeyMasked && (kv[u] >> 21) == p1.bin, hpad((kv[u] >> 10) & 2047u));
            __syncthreads();
            const PickResult p2 = wave_pick(hist2, k - p1.above);
            skip = margin_key((p1.bin << 21) | (p2.bin << 10), qinfo[q]);
        }
    } else {
        auto key_at = [&](uint32_t e) { return keys[e]; };
        const uint32_t valid = count_valid(key_at, n_sample, &s_cnt);
        if (valid >= k) skip = margin_key(radix2(key_at, n_sample, k, hist, &pick), qinfo[q]);
    }
    // combine_max: a second, tighter bound from tile maxima of the main sweep itself never loosens the one already there
    if (threadIdx.x == 0) skip_key[q] = combine_max ? max(skip_key[q], skip) : skip;
}

hipError_t launch_sample_bound(const uint32_t* tmax_sample, uint64_t stride, uint32_t n_sample, const QInfo* qinfo,
                               uint32_t nq, uint32_t k, uint32_t* skip_key, hipStream_t s, int combine_max) {
    hipLaunchKernelGGL(sample_bound_kernel, dim3(nq), dim3(kSelThreads), 0, s, tmax_sample, stride, n_sample, qinfo, k,
                       skip_key, combine_max);
    return hipGetLastError();
}

// measurement build (tools/build_variant.sh seltrace "-DNMN_SELECT_TRACE" nmn_select): query 0 prints where its selection's time goes
// (100 MHz wall clock ticks = 10 ns), one line per call
#ifdef NMN_SELECT_TRACE
#define SEL_MARK(i) do { if (q == 0 && tid == 0) sel_t[i] = wall_clock64(); } while (0)
#else
#define SEL_MARK(i) do { } while (0)
#endif
constexpr uint32_t kCompCap = 8192;
constexpr uint32_t kFlatTiles = 16384;     // shards of up to this many tiles (1M rows) select from ALL their tile maxima at once (flat path)
constexpr uint32_t kFlatGroupK = 256;      // ... and for k up to this, from the k-th largest of the 1024 per-thread group maxima
constexpr uint32_t kShortRowList = 1024;  // row lists up to this long skip the third radix pick
constexpr uint32_t kBailTiles = 1024;  // tiles within the margin beyond which a selection hands over to the crowd kernels (when they follow)
constexpr size_t kSelectLds = 2 * kCompCap * 8 + kMaxScanWaves * 4 + kHistPad * 4;  // 152.25 KiB

__global__ void __launch_bounds__(kSelThreads) select_kernel(SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel_lds[];
    unsigned long long* LT = sel_lds;                              // (key << 32 | tile)
    unsigned long long* LR = sel_lds + kCompCap;                   // (key << 32 | row)
    uint32_t* wk = reinterpret_cast<uint32_t*>(sel_lds + 2 * kCompCap);
    uint32_t* hist = wk + kMaxScanWaves;
    uint32_t* la = reinterpret_cast<uint32_t*>(LR);                // passing waves (dead before LR is filled)
    uint32_t* lb = la + kListCap;                                  // generic path: tile list
    __shared__ PickResult pick;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_vw;
    const uint32_t S = p.split ? p.split : 1u;  // parts per query (see SelectParams::split)
    const uint32_t q = blockIdx.x / S, part = blockIdx.x % S;
    const uint32_t tid = threadIdx.x;
    const uint32_t nql = p.nql;
#ifdef NMN_SELECT_TRACE
    unsigned long long sel_t[10] = {};
#endif
    SEL_MARK(0);
    if (p.retry && p.qstate[q].overflow != 1) {  // block-uniform: this query's first selection stood (or its crowd list did)
        // ... or it was found hopeless for the retry (below): the retry sweep skipped it, from here on it is an ordinary overflow
        if (tid == 0 && p.qstate[q].overflow == 4u) p.qstate[q].overflow = 1u;
        return;
    }
    if (p.half_stats && tid == 0 && part == 0) atomicAdd(p.half_stats + (p.retry ? 1 : 0), 1u);  // feeds the shard's mirror on/off switch
    if (p.crowd_count_reset && tid == 0 && part == 0) p.crowd_count_reset[q] = 0u;  // the crowd kernels behind this selection count from zero
    if (p.fb_sync_reset && q == 0 && part == 0 && tid == 0) {  // arrival counter + abort flag of the fallback_select launch that follows on this stream
        p.fb_sync_reset[0] = 0ull;
        p.fb_sync_reset[1] = 0ull;
    }
    auto score_bits = [&](uint64_t row) -> uint32_t { return p.scores[score_at(row, q, nql)]; };
    const uint32_t* tmax = p.tmax + (uint64_t)q * p.tmax_stride;
    const uint32_t* wmax = p.wmax + (uint64_t)q * p.wmax_stride;
    const uint32_t W = p.n_waves, tpw = p.tiles_per_wave, n_tiles = p.n_tiles;
    const bool strided = p.strided != 0;
    auto tile_of = [&](uint32_t w, uint32_t j) -> uint32_t { return strided ? j * W + w : w * tpw + j; };  // tile j of scan wave w
    const QInfo qi = p.qinfo[q];
    uint32_t* out = p.cand_rows + (size_t)q * p.cand_cap;
    // rank of the threshold: k, plus (f64 artifact similarity only) the rows whose approximate score is forced to
    // +inf because f32 cannot vouch for them — they occupy the top ranks without being real contenders
    const uint32_t k = p.k + (p.k_extra ? *p.k_extra : 0u);
    constexpr int V = 8;

    const uint32_t skip = p.skip_key ? p.skip_key[q] : kKeyNaN;
    uint32_t vw = 0, Tw = kKeyNaN, Twm = kKeyNaN, nA = 0;  // (nA: passing waves — the trace build prints it; 0 on the flat path)
    bool flat = false;
    // ---- flat path (round 6): shards of <= 16 384 tiles (1M rows) --------------------------------------------------------------
    // The three-level walk below pays a radix pick per level (4-5 us each on this 16-wave workgroup) and, behind the WAVE level's
    // loose bound (the k-th largest of a few hundred wave maxima), gathers the scores of 5-7 k tiles where k (1 + small) hold a
    // candidate: at 1M x 768, k = 100 the selection took 33-36 us — load 2, picks 4 + 4 + 5, tile maxima 7, row scores 11
    // (profiles/r05k_select_phases.txt) — a quarter of a lone call's 130-us sweep.  A shard this small has few enough TILES to put
    // every tile maximum of the query into LDS at once (64 KiB, where the row list will live afterwards): ONE pick over them gives
    // the tile-level bound directly — the k-th largest tile maximum, the tightest bound the maxima can give —, the tiles that reach
    // it (~k) are compacted from LDS, and the row gather reads ~k x 64 scores instead of ~6 k x 64.  No wave level at all.
    // Lists that overflow (thousands of tiles inside the margin) and selections that hand over to the crowd kernels continue in
    // the code below exactly as before: LT, ct, Tw / Twm mean the same things there.  NMN_NO_FLAT_SELECT=1 (host side): the A/B.
    // ---- split selection (round 6): S workgroups per query on shards too large for the flat path ----------------------------------
    // One workgroup walks the 625 KB of a 10M-row query's tile maxima at ~40 GB/s (profiles/r06aa_*); S of them hold a part each in
    // LDS.  They meet once: every thread's maximum over ITS tiles of the part is folded into split_sg[q][thread] (atomicMax: 1024
    // super-groups, each a disjoint set of tiles), an arrival counter says when all parts are in, and every part picks the SAME bound
    // — the k-th largest super-group maximum: k super-groups, k tiles — for itself.  Then a part is a flat selection over its own
    // tiles, except that its candidates go behind the other parts' (a slice reserved by one global atomic) and the last part to finish
    // writes the query's state.  Anything unusual in a part — a list that overflows, a crowd to hand over — flags the query
    // `overflow` with the common tile-level threshold: the crowd kernels / the host's follow-up take it from there.
    // A part's end, ONE device-scope atomic: split_ctr[q] holds (parts finished << 32 | candidates reserved so far) as one 64-bit word
    // behind the arrival counter; adding (1 << 32 | mine) reserves this part's slice of the query's candidate list (the old low half
    // is its start) and tells whether it is the LAST part (the old high half) — which then knows the total, writes the query's state
    // and zeroes the meeting places (every part has read the super-group maxima before it came here, and touches this word once).
    // A part in trouble reserves cand_cap + 1 candidates: the total overflows the list, which is what its trouble amounts to.
    // No fences: the hand-overs are atomics, the candidate rows are for the NEXT kernel (ordered by the kernel boundary).
    // Returns the start of the part's slice.
    auto split_reserve_and_finish = [&](uint32_t mine, bool trouble, uint32_t thr_common) -> uint32_t {
        __shared__ uint32_t s_base, s_last;
        __syncthreads();
        if (tid == 0) {
            unsigned long long* word = reinterpret_cast<unsigned long long*>(p.split_ctr + 4u * q + 2u);
            const uint32_t add = trouble ? p.cand_cap + 1u : mine;
            const unsigned long long old = atomicAdd(word, (1ull << 32) | (unsigned long long)add);
            s_base = (uint32_t)(old & 0xFFFFFFFFull);
            s_last = (uint32_t)(old >> 32) == S - 1u ? 1u : 0u;
            if (s_last) {
                const uint32_t total = s_base + add;
                QState st;
                st.n_valid = vw;
                st.thr_key = thr_common;
                st.overflow = total > p.cand_cap ? 1u : 0u;
                st.cand_count = st.overflow ? 0u : total;
                p.qstate[q] = st;
                if (p.count_overflows && p.half_stats && st.overflow) atomicAdd(p.half_stats + 1, 1u);
                if (p.l2_hint && q == 0) {
                    const float tau = key_to_score(thr_common);
                    *p.l2_hint = (tau > 0.0f && tau <= 1.0f) ? 1.0f / tau - 1.0f : 0.0f;
                }
                p.split_ctr[4u * q + 0u] = 0u;
                *word = 0ull;
#ifdef NMN_SELECT_TRACE
                if (q == 0) {
                    const unsigned long long t_now = wall_clock64();
                    printf("select (split, last of %u parts: part %u) cand=%u | ticks: load %llu meet %llu pick %llu compaction %llu rest %llu total %llu\n", S, part,
                           total, sel_t[1] - sel_t[0], sel_t[2] - sel_t[1], sel_t[3] - sel_t[2], sel_t[4] - sel_t[3], t_now - sel_t[4], t_now - sel_t[0]);
                }
#endif
            }
        }
        __syncthreads();
        if (s_last) p.split_sg[(size_t)q * kSelThreads + tid] = kKeyMasked;
        return s_base;
    };
    if (S > 1) {
        uint32_t* tk = reinterpret_cast<uint32_t*>(LR);
        uint32_t* ctr = p.split_ctr + 4u * q;
        const uint32_t per = ((n_tiles + S - 1u) / S + 3u) & ~3u;  // tiles per part: a multiple of 4 (16-byte loads), <= kFlatTiles (launch_select)
        const uint32_t tA = min(part * per, n_tiles), tB = min(tA + per, n_tiles), nloc = tB - tA;
        if (tid == 0) { s_vw = 0; s_w[0] = 0; s_w[1] = 0; s_w[2] = 0; s_w[3] = 0; }
        __syncthreads();
        uint32_t my_valid = 0, my_max = kKeyMasked;
        {
            const uint4* t4 = reinterpret_cast<const uint4*>(tmax + tA);
            const uint32_t n4 = (nloc + 3u) >> 2;
            uint4 v4[kFlatTiles / 4 / kSelThreads];
#pragma unroll
            for (int u = 0; u < (int)(kFlatTiles / 4 / kSelThreads); u++) {
                const uint32_t e = tid + (uint32_t)u * kSelThreads;
                v4[u] = e < n4 ? t4[e] : make_uint4(kKeyMasked, kKeyMasked, kKeyMasked, kKeyMasked);
            }
#pragma unroll
            for (int u = 0; u < (int)(kFlatTiles / 4 / kSelThreads); u++) {
                const uint32_t e = tid + (uint32_t)u * kSelThreads;
                if (e >= n4) continue;
                uint32_t kk4[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (e * 4u + (uint32_t)c >= nloc) kk4[c] = kKeyMasked;
                    my_valid += kk4[c] != kKeyMasked;
                    my_max = max(my_max, kk4[c]);
                }
                *reinterpret_cast<uint4*>(tk + e * 4u) = make_uint4(kk4[0], kk4[1], kk4[2], kk4[3]);
            }
        }
        // (a RETURNING atomic: the wave waits for it, so it has been performed when the arrival below is counted — a release fence per
        //  thread would write the L2 back 1024 x S times; cf. NMN_PRED_TICKET in nmn_api.hip)
        uint32_t seen_before = 0;
        if (my_max != kKeyMasked) seen_before = atomicMax(p.split_sg + (size_t)q * kSelThreads + tid, my_max);
        asm volatile("" ::"v"(seen_before));
        {
            uint32_t t = my_valid;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += (uint32_t)__shfl_xor((int)t, off);
            if ((tid & 63u) == 0 && t) atomicAdd(&s_vw, t);
        }
        __syncthreads();
        vw = s_vw;
        SEL_MARK(1);
        if (tid == 0) {
            atomicAdd(ctr + 0, 1u);
            // all parts are resident together (S <= 16 workgroups per query, lone callers only): wait for them — but never for ever:
            // a part that gives up after ~2 ms picks its bound from whatever super-group maxima are in (still a valid bound)
            const unsigned long long t_end = wall_clock64() + 200000ull;
            while (__hip_atomic_load(ctr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S && wall_clock64() < t_end) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        SEL_MARK(2);
        {
            const uint32_t g = __hip_atomic_load(p.split_sg + (size_t)q * kSelThreads + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wk[tid] = g;
            uint32_t gv = g != kKeyMasked ? 1u : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) gv += (uint32_t)__shfl_xor((int)gv, off);
            if ((tid & 63u) == 0 && gv) atomicAdd(&s_w[0], gv);
        }
        __syncthreads();
        if (s_w[0] >= k) Tw = radix2([&](uint32_t e) { return wk[e]; }, (uint32_t)kSelThreads, k, hist, &pick);  // (block-uniform)
        Twm = max(margin_key(Tw, qi), skip);
        SEL_MARK(3);
        for (uint32_t b0 = tid & ~63u; b0 < nloc; b0 += kSelThreads) {
            const uint32_t e = b0 + (tid & 63u);
            const uint32_t key = e < nloc ? tk[e] : kKeyMasked;
            const bool pr = key != kKeyMasked && key >= Twm;
            const uint32_t pos = wave_append(pr, &s_w[1]);
            if (pr && pos < kCompCap) LT[pos] = ((unsigned long long)key << 32) | (tA + e);
        }
        flat = true;
    }
    if (S == 1 && p.flat && n_tiles <= kFlatTiles) {
        uint32_t* tk = reinterpret_cast<uint32_t*>(LR);  // [n_tiles] tile maxima of the query (dead before LR is filled)
        if (tid == 0) { s_vw = 0; s_w[0] = 0; s_w[1] = 0; s_w[2] = 0; s_w[3] = 0; }
        __syncthreads();
        uint32_t my_valid = 0, my_max = kKeyMasked;  // (my_max: the largest of this thread's <= 16 tile maxima; kKeyMasked = 0 is below every key)
        if ((p.tmax_stride & 3ull) == 0ull) {
            const uint4* t4 = reinterpret_cast<const uint4*>(tmax);
            const uint32_t n4 = (n_tiles + 3u) >> 2;  // <= 4096: at most four 16-byte loads per thread, all in flight
            uint4 v4[kFlatTiles / 4 / kSelThreads];
#pragma unroll
            for (int u = 0; u < (int)(kFlatTiles / 4 / kSelThreads); u++) {
                const uint32_t e = tid + (uint32_t)u * kSelThreads;
                v4[u] = e < n4 ? t4[e] : make_uint4(kKeyMasked, kKeyMasked, kKeyMasked, kKeyMasked);
            }
#pragma unroll
            for (int u = 0; u < (int)(kFlatTiles / 4 / kSelThreads); u++) {
                const uint32_t e = tid + (uint32_t)u * kSelThreads;
                if (e >= n4) continue;
                uint32_t kk4[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (e * 4u + (uint32_t)c >= n_tiles) kk4[c] = kKeyMasked;  // (the words behind the last tile are not this query's)
                    my_valid += kk4[c] != kKeyMasked;
                    my_max = max(my_max, kk4[c]);
                }
                *reinterpret_cast<uint4*>(tk + e * 4u) = make_uint4(kk4[0], kk4[1], kk4[2], kk4[3]);
            }
        } else {
            for (uint32_t e = tid; e < n_tiles; e += kSelThreads) {
                const uint32_t key = tmax[e];
                tk[e] = key;
                my_valid += key != kKeyMasked;
                my_max = max(my_max, key);
            }
        }
        wk[tid] = my_max;  // the group maxima: one per thread, disjoint sets of tiles
        {
            uint32_t t = my_valid, gv = my_max != kKeyMasked ? 1u : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                t += (uint32_t)__shfl_xor((int)t, off);
                gv += (uint32_t)__shfl_xor((int)gv, off);
            }
            if ((tid & 63u) == 0 && t) {
                atomicAdd(&s_vw, t);
                atomicAdd(&s_w[0], gv);
            }
        }
        __syncthreads();
        vw = s_vw;  // (here: the valid TILES)
        SEL_MARK(2);
        const uint32_t vgroups = s_w[0];  // threads holding at least one valid tile
        SEL_MARK(1);
        if (vw == 0) {
            if (tid == 0) { p.qstate[q].cand_count = 0; p.qstate[q].n_valid = 0; }
            return;
        }
        // The bound: the k-th largest tile maximum — or, for small k, the k-th largest GROUP maximum (a thread's <= 16 tiles are a
        // group: k groups whose best tile reaches T are k tiles that do, so T is a valid lower bound as well, and since a group rarely
        // holds two of the best k tiles it admits k (1 + k / 2048) tiles instead of k).  The pick over 1024 keys, one per thread,
        // is 4 us; over all 15 625 tile maxima it measured 11 (profiles/r06b_select_flat.txt).
        if (k <= kFlatGroupK && vgroups >= k) Tw = radix2([&](uint32_t e) { return wk[e]; }, (uint32_t)kSelThreads, k, hist, &pick);
        else if (vw >= k) Tw = radix2([&](uint32_t e) { return tk[e]; }, n_tiles, k, hist, &pick);
        Twm = max(margin_key(Tw, qi), skip);
        SEL_MARK(3);
        for (uint32_t b0 = tid & ~63u; b0 < n_tiles; b0 += kSelThreads) {  // (on the wave's first index: wave_append needs whole waves)
            const uint32_t e = b0 + (tid & 63u);
            const uint32_t key = e < n_tiles ? tk[e] : kKeyMasked;
            const bool pr = key != kKeyMasked && key >= Twm;
            const uint32_t pos = wave_append(pr, &s_w[1]);
            if (pr && pos < kCompCap) LT[pos] = ((unsigned long long)key << 32) | e;
        }
        flat = true;
    }
    // (Larger shards keep the three-level walk.  The same idea with the tile maxima STREAMED twice instead of held in LDS — pass 1:
    //  per-thread group maxima, pick, pass 2: compaction — was built and measured at 10M rows: one workgroup walks the 625 KB of a
    //  query's tile maxima at ~40 GB/s, 12.5 + 18.5 us for the two passes: 40-49 us against the walk's 28 (f32 rows) / 45 (8-bit);
    //  level at 3M rows.  profiles/r06aa_select_streamed_flat_path.txt.  Removed.)
    // the wave maxima into LDS, counted on the way (one phase: the count used to be a second walk over them, two barriers more)
    auto load_wave_maxima = [&]() -> uint32_t {
        uint32_t valid = 0;
        uint32_t wv4[kMaxScanWaves / kSelThreads];
#pragma unroll
        for (int u = 0; u < (int)(kMaxScanWaves / kSelThreads); u++) {
            const uint32_t i = tid + (uint32_t)u * kSelThreads;
            wv4[u] = i < W ? wmax[i] : kKeyMasked;
        }
#pragma unroll
        for (int u = 0; u < (int)(kMaxScanWaves / kSelThreads); u++) {
            wk[tid + (uint32_t)u * kSelThreads] = wv4[u];
            valid += wv4[u] != kKeyMasked;
        }
        return valid;
    };
    if (!flat) {
    if (tid == 0) s_vw = 0;
    const uint32_t my_valid = load_wave_maxima();
    __syncthreads();
    SEL_MARK(1);
    {
        // one LDS atomic per wave
        uint32_t t = my_valid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += (uint32_t)__shfl_xor((int)t, off);
        if ((tid & 63u) == 0 && t) atomicAdd(&s_vw, t);
    }
    __syncthreads();
    vw = s_vw;  // (s_vw is written nowhere else: no race with the counters reset below)
    SEL_MARK(2);
    if (vw == 0) {
        if (tid == 0) { p.qstate[q].cand_count = 0; p.qstate[q].n_valid = 0; }
        return;
    }
    // ---- level W ---------------------------------------------------------------------------
    if (vw >= k) Tw = radix2([&](uint32_t e) { return wk[e]; }, W, k, hist, &pick);
    // scores of tiles whose maximum is below `skip` were never written by the batched sweep: no threshold that
    // gates a read of scores[] may fall below it (rows below it cannot be in the top-k anyway)
    SEL_MARK(3);
    Twm = max(margin_key(Tw, qi), skip);
    if (tid == 0) { s_w[0] = 0; s_w[1] = 0; s_w[2] = 0; s_w[3] = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < W; i += kSelThreads)
        if (wk[i] != kKeyMasked && wk[i] >= Twm) la[atomicAdd(&s_w[0], 1u)] = i;  // <= W <= kListCap
    __syncthreads();
    nA = s_w[0];
    // ---- level T: compact (key,tile) of tiles >= Twm in passing waves -------------------------
    {
        // The tile maxima of the passing waves: nA x tpw of them (k = 100 at 10M rows: ~800 x 39; k = 1000 of ~4000 waves: the wave
        // level prunes next to nothing, ~3300 x 39 = 128 k) through this one workgroup.  No integer division per element — a flat
        // index split by / tpw and % tpw cost this loop 60 of its 100 us at 128 k elements (round 4, profiles/r04j_*): the lanes of
        // a wave take the dimension that is contiguous in memory — the tiles j of one scan wave, or (strided sweeps: tile j of wave w
        // is j * W + w) neighbouring passing waves of one j — in pieces of P = a power of two, 64 / P pieces per wave, and the
        // workgroup's 16 waves stride the other dimension, V pieces in flight each.
        const uint32_t wv = tid >> 6, ln = tid & 63u;
        // ... and when the wave level prunes less than half of the tiles (k = 1000: 82 % of the waves pass) the passing waves are not
        // looked up at all: the query's tile maxima are read LINEARLY, 16 bytes per lane (a tile that reaches Twm sits in a wave that
        // does, so the set is the same) — 38 loads per wave instead of 205 dependent ones: 75-95 us -> see profiles/r04j_*.
        const bool dense = (uint64_t)nA * tpw * 2ull >= (uint64_t)n_tiles && (p.tmax_stride & 3ull) == 0ull;
        if (dense) {
            const uint4* t4 = reinterpret_cast<const uint4*>(tmax);
            const uint32_t n4 = (n_tiles + 3u) >> 2;
            // (the loop runs on the WAVE's first index: every lane of a wave takes every trip, the scan below reads lane 63)
            for (uint32_t b0 = tid & ~63u; b0 < n4; b0 += kSelThreads * V) {
                const uint32_t e0 = b0 + ln;
                uint4 v4[V];
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                    v4[u] = e < n4 ? t4[e] : make_uint4(kKeyMasked, kKeyMasked, kKeyMasked, kKeyMasked);
                }
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t t0 = (e0 + (uint32_t)u * kSelThreads) * 4u;
                    const uint32_t kk4[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
                    uint32_t cnt = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) cnt += (t0 + (uint32_t)c < n_tiles && kk4[c] != kKeyMasked && kk4[c] >= Twm) ? 1u : 0u;
                    // one LDS atomic per wave and load: an inclusive scan of the lanes' counts places every lane's entries
                    uint32_t incl = cnt;
#pragma unroll
                    for (uint32_t dd = 1; dd < 64; dd <<= 1) {
                        const uint32_t t = (uint32_t)__shfl_up((int)incl, (int)dd);
                        if (ln >= dd) incl += t;
                    }
                    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
                    if (total == 0) continue;  // (wave-uniform)
                    uint32_t base = 0;
                    if (ln == 0) base = atomicAdd(&s_w[1], total);
                    base = (uint32_t)__shfl((int)base, 0);
                    uint32_t pos = base + incl - cnt;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (t0 + (uint32_t)c < n_tiles && kk4[c] != kKeyMasked && kk4[c] >= Twm) {
                            if (pos < kCompCap) LT[pos] = ((unsigned long long)kk4[c] << 32) | (t0 + (uint32_t)c);
                            pos++;
                        }
                    }
                }
            }
        }
        // Contiguous ranges (unmasked sweeps), tpw >= 8: the tile maxima of a passing wave are tpw consecutive words — 16 bytes per
        // lane, ceil(tpw / 4) lanes per passing wave, several passing waves per load instruction.  10M x 768: 750 passing waves x 39
        // tiles were six round trips of 39-of-64-lane dword loads (25 us of the selection's 62, profiles/r04y_*); now two.
#ifndef NMN_SELECT_TILES_DWORD
        const bool quads = !dense && !strided && tpw >= 8u;
#else
        const bool quads = false;
#endif
        if (quads) {
            const uint32_t q4 = (tpw + 3u) >> 2;  // quads per passing wave
            uint32_t lgQ = 0;
            while (lgQ < 6u && (1u << lgQ) < q4) lgQ++;
            const uint32_t PQ = 1u << lgQ, perw = 64u >> lgQ;  // lanes per passing wave, passing waves per wave and load
            const uint32_t quad = ln & (PQ - 1u), sub = ln >> lgQ;
            struct __attribute__((packed, aligned(4))) Quad { uint32_t v[4]; };
            for (uint32_t q40 = 0; q40 < q4; q40 += 64u) {  // (only ranges of more than 256 tiles per wave loop here)
                const uint32_t qd = q40 + quad;
                for (uint32_t o0 = wv * perw; o0 < nA; o0 += (kSelThreads / 64) * perw * V) {
                    uint32_t tk[V][4], tb[V];
#pragma unroll
                    for (int u = 0; u < V; u++) {
                        const uint32_t o = o0 + (uint32_t)u * (kSelThreads / 64) * perw + sub;
                        const bool ok = qd < q4 && o < nA;
                        tb[u] = ok ? la[o] * tpw + qd * 4u : 0xFFFFFFFFu;
#pragma unroll
                        for (int c = 0; c < 4; c++) tk[u][c] = kKeyMasked;
                        if (ok) {
                            const uint32_t lim = min(la[o] * tpw + tpw, n_tiles);  // end of this wave's range
                            if (tb[u] + 3u < lim) {
                                const Quad v = *reinterpret_cast<const Quad*>(tmax + tb[u]);
#pragma unroll
                                for (int c = 0; c < 4; c++) tk[u][c] = v.v[c];
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; c++)
                                    if (tb[u] + (uint32_t)c < lim) tk[u][c] = tmax[tb[u] + (uint32_t)c];
                            }
                        }
                    }
                    // (one LDS atomic per wave and trip for all 4 * V elements of its lanes; it was one per element)
                    uint32_t cnt = 0;
#pragma unroll
                    for (int u = 0; u < V; u++)
#pragma unroll
                        for (int c = 0; c < 4; c++) cnt += (tk[u][c] != kKeyMasked && tk[u][c] >= Twm) ? 1u : 0u;
                    uint32_t pos = wave_append_cnt(cnt, &s_w[1]);
#pragma unroll
                    for (int u = 0; u < V; u++) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if (tk[u][c] != kKeyMasked && tk[u][c] >= Twm) {
                                if (pos < kCompCap) LT[pos] = ((unsigned long long)tk[u][c] << 32) | (tb[u] + (uint32_t)c);
                                pos++;
                            }
                        }
                    }
                }
            }
        }
        const uint32_t n_in = (dense || quads) ? 0u : (strided ? nA : tpw), n_out = strided ? tpw : nA;   // inner: across lanes; outer: across waves / iterations
        uint32_t lgP = 0;
        while (lgP < 6u && (1u << lgP) < n_in) lgP++;
        const uint32_t P = 1u << lgP, per = 64u >> lgP;          // lanes per piece, pieces (outer indices) per wave and step
        const uint32_t in_l = ln & (P - 1u), sub = ln >> lgP;
        for (uint32_t in0 = 0; in0 < n_in; in0 += 64u) {          // (only rows of more than 64 tiles per wave / passing waves loop here)
            const uint32_t in = in0 + in_l;
            for (uint32_t o0 = wv * per; o0 < n_out; o0 += (kSelThreads / 64) * per * V) {
                uint32_t tk[V], tt[V];
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t o = o0 + (uint32_t)u * (kSelThreads / 64) * per + sub;
                    const bool ok = in < n_in && o < n_out;
                    tt[u] = ok ? (strided ? tile_of(la[in], o) : tile_of(la[o], in)) : 0xFFFFFFFFu;
                    tk[u] = tt[u] < n_tiles ? tmax[tt[u]] : kKeyMasked;
                }
                // (appends are counted per wave and trip: one LDS atomic for all V elements of all its lanes)
                uint32_t cnt = 0;
#pragma unroll
                for (int u = 0; u < V; u++) cnt += (tk[u] != kKeyMasked && tk[u] >= Twm) ? 1u : 0u;
                uint32_t pos = wave_append_cnt(cnt, &s_w[1]);
#pragma unroll
                for (int u = 0; u < V; u++) {
                    if (tk[u] != kKeyMasked && tk[u] >= Twm) {
                        if (pos < kCompCap) LT[pos] = ((unsigned long long)tk[u] << 32) | tt[u];
                        pos++;
                    }
                }
            }
        }
    }
    }  // (!flat)
    __syncthreads();
    SEL_MARK(4);
    const uint32_t ct = s_w[1];
    uint32_t Tc = Twm;
    bool done = false;
    // Many tiles within the margin (k = 1000 under an 8-bit margin: 6 000 - 16 000): everything below would be ONE workgroup
    // reading ct x 64 scores (and, past 8192 tiles, the tile maxima three more times) — 0.39 ms at 10 000 tiles against a
    // 0.28 ms sweep.  When the crowd kernels follow this selection they do that walk with the whole device: leave them the
    // tile-level bound — the k-th largest maximum among the tiles gathered so far (any subset gives a valid lower bound on the
    // k-th best score) — and go.  crowd_alloc turns the query into a crowd (every row >= the bound is re-scored exactly), or,
    // if that is more than an eighth of the shard, leaves it to the f32 retry like any other overflow.
    if (S > 1 && ((p.crowd_follows && !p.retry && ct > kBailTiles) || ct > kCompCap)) {  // (block-uniform) a part in trouble: the query overflows
        (void)split_reserve_and_finish(0u, true, Twm);
        return;
    }
    if (p.crowd_follows && !p.retry && ct > kBailTiles) {
        const uint32_t have = min(ct, kCompCap);
        uint32_t T2 = Tw;
        if (have > k) T2 = radix2([&](uint32_t e) { return (uint32_t)(LT[e] >> 32); }, have, k, hist, &pick);
        const uint32_t T2m = max(max(margin_key(T2, qi), skip), Twm);
        if (tid == 0) {
            QState st;
            st.n_valid = vw;
            st.thr_key = T2m;
            st.overflow = 1u;
            st.cand_count = 0u;
            p.qstate[q] = st;
            if (p.count_overflows && p.half_stats) atomicAdd(p.half_stats + 1, 1u);
            if (p.l2_hint && q == 0) {
                const float tau = key_to_score(T2m);
                *p.l2_hint = (tau > 0.0f && tau <= 1.0f) ? 1.0f / tau - 1.0f : 0.0f;
            }
#ifdef NMN_SELECT_TRACE
            if (q == 0) {
                sel_t[5] = wall_clock64();
                printf("select (hands over) W=%u vw=%u nA=%u ct=%u strided=%d | ticks: load %llu count %llu radixW %llu tiles %llu radixT %llu total %llu\n", W, vw, nA,
                       ct, (int)strided, sel_t[1] - sel_t[0], sel_t[2] - sel_t[1], sel_t[3] - sel_t[2], sel_t[4] - sel_t[3], sel_t[5] - sel_t[4], sel_t[5] - sel_t[0]);
            }
#endif
        }
        return;
    }
    if (ct <= kCompCap) {
        uint32_t T2 = Tw;
        // (flat / split: Tw IS the tile-level bound — unless no bound could be formed there: fewer than k super-groups hold a tile when
        //  a bitmap keeps a few runs of rows, an IVF probe's lists; then the tiles of THIS list give one, as on the walk)
        if (ct > k && (!flat || Tw == kKeyNaN)) T2 = radix2([&](uint32_t e) { return (uint32_t)(LT[e] >> 32); }, ct, k, hist, &pick);
        const uint32_t T2m = max(margin_key(T2, qi), skip);
        Tc = T2m;
        SEL_MARK(5);
        // ---- level R: compact (key,row) of rows >= T2m in tiles >= T2m (la is dead: LR may be written)
        __syncthreads();
        const uint32_t tot = ct * kTileRows;
        // (this gather is the longest chain of the kernel: ct x 64 scores through ONE workgroup, a memory round trip per batch of
        //  loads — under the 8-bit margin ct is ~700 at 1M rows, k = 100: 16 loads per thread in flight instead of 8)
#ifndef NMN_SELECT_ROWS_DWORD
        // 16 bytes per lane: the 64 scores of a tile are one 256-byte block of scores[] (score_at), so 16 lanes take a tile and a
        // load instruction four tiles per wave — a quarter of the loads (and of the LT look-ups) of the dword form below, which is
        // kept for the A/B (-DNMN_SELECT_ROWS_DWORD): profiles/r04y_* has this gather at 17 of the selection's 41 us at 1M x 768.
        (void)tot;
        // (12 loads per thread in flight: 768 tiles per round trip — k = 100 is one or two trips; one LDS atomic per wave and trip)
        constexpr int VR4 = 12;
        const uint32_t tot4 = ct * (kTileRows / 4u);
        for (uint32_t b0 = tid & ~63u; b0 < tot4; b0 += kSelThreads * VR4) {  // (on the wave's first index: every lane takes every trip)
            const uint32_t e0 = b0 + (tid & 63u);
            uint4 kb[VR4];
#pragma unroll
            for (int u = 0; u < VR4; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                kb[u] = make_uint4(kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits);
                if (e < tot4) {
                    const unsigned long long ent = LT[e >> 4];
                    if ((uint32_t)(ent >> 32) >= T2m)
                        kb[u] = *reinterpret_cast<const uint4*>(p.scores + score_at((uint64_t)(uint32_t)(ent & 0xFFFFFFFFull) * kTileRows + (e & 15u) * 4u, q, nql));
                }
            }
            // "key >= T2m" as ONE float compare per score: the key order is the score order (-0.0 == +0.0 on both sides), a NaN score
            // and the sentinel (a NaN pattern) compare false as their keys (<= kKeyNaN) do — valid whenever T2m is a score's key; a threshold
            // at or below the NaN key (fewer than k valid tiles) takes the key form.  The keys of the few that pass are formed below.
            const bool by_float = T2m >= kKeyNegInf;  // (keys between the NaN key and key(-inf) are no score's: key form)
            const float tau2 = key_to_score(T2m);
            uint32_t m[VR4];  // which of the four scores of load u pass
            if (by_float) {
#pragma unroll
                for (int u = 0; u < VR4; u++)
                    m[u] = (u2f(kb[u].x) >= tau2 ? 1u : 0u) | (u2f(kb[u].y) >= tau2 ? 2u : 0u) | (u2f(kb[u].z) >= tau2 ? 4u : 0u) | (u2f(kb[u].w) >= tau2 ? 8u : 0u);
            } else {
#pragma unroll
                for (int u = 0; u < VR4; u++) {
                    const uint32_t bits4[4] = {kb[u].x, kb[u].y, kb[u].z, kb[u].w};
                    m[u] = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t key = bits_to_key(bits4[c]);
                        m[u] |= (key != kKeyMasked && key >= T2m) ? (1u << c) : 0u;
                    }
                }
            }
            uint32_t cnt = 0;
#pragma unroll
            for (int u = 0; u < VR4; u++) cnt += (uint32_t)__builtin_popcount(m[u]);
            uint32_t pos = wave_append_cnt(cnt, &s_w[2]);
            if (cnt) {
#pragma unroll
                for (int u = 0; u < VR4; u++) {
                    if (m[u] == 0u) continue;
                    const uint32_t bits4[4] = {kb[u].x, kb[u].y, kb[u].z, kb[u].w};
                    const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                    const uint32_t row0 = (uint32_t)(LT[min(e, tot4 - 1u) >> 4] & 0xFFFFFFFFull) * kTileRows + (e & 15u) * 4u;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if ((m[u] >> c) & 1u) {
                            if (pos < kCompCap) LR[pos] = ((unsigned long long)bits_to_key(bits4[c]) << 32) | (row0 + (uint32_t)c);
                            pos++;
                        }
                    }
                }
            }
        }
#else
        constexpr int VR = NMN_SELECT_VR;
        for (uint32_t e0 = tid; e0 < tot; e0 += kSelThreads * VR) {
            uint32_t kb[VR];  // (only the loaded scores wait in registers: the row of an element is read from LT again when it is appended)
#pragma unroll
            for (int u = 0; u < VR; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                kb[u] = kScoreSentinelBits;
                if (e < tot) {
                    const unsigned long long ent = LT[e >> 6];
                    if ((uint32_t)(ent >> 32) >= T2m) kb[u] = score_bits((uint32_t)(ent & 0xFFFFFFFFull) * kTileRows + (e & 63u));
                }
            }
#pragma unroll
            for (int u = 0; u < VR; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                const uint32_t key = bits_to_key(kb[u]);
                const bool pr = key != kKeyMasked && key >= T2m;
                const uint32_t pos = wave_append(pr, &s_w[2]);
                if (pr && pos < kCompCap) LR[pos] = ((unsigned long long)key << 32) | ((uint32_t)(LT[min(e, tot - 1u) >> 6] & 0xFFFFFFFFull) * kTileRows + (e & 63u));
            }
        }
#endif
        __syncthreads();
        SEL_MARK(6);
        const uint32_t cr = s_w[2];
        if (cr <= kCompCap) {
            uint32_t T3 = T2;
            // (A row list this short goes to the rescore as it is: the third pick — 5 us — pruned NOTHING on ordinary data (cand == cr in
            //  every line of profiles/r05k_select_phases.txt: one row per passing tile reaches the tile bound, and the row bound moves
            //  by less than the margin), and <= 1024 candidates are one rescore step and one entry per thread of final_kernel's sort.)
            if (cr > k && cr > min(kShortRowList, p.cand_cap)) T3 = radix2([&](uint32_t e) { return (uint32_t)(LR[e] >> 32); }, cr, k, hist, &pick);
            SEL_MARK(7);
            Tc = max(margin_key(T3, qi), skip);  // >= T2m: every row that can matter is in LR
            if (S > 1) {  // a part: count its candidates, reserve their slice of the query's list (and finish: one atomic), write them there
                uint32_t mine = 0;
                for (uint32_t e = tid; e < cr; e += kSelThreads) mine += ((uint32_t)(LR[e] >> 32) >= Tc) ? 1u : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
                if ((tid & 63u) == 0 && mine) atomicAdd(&s_w[3], mine);
                __syncthreads();
                const uint32_t c_local = s_w[3];
                const uint32_t base = split_reserve_and_finish(c_local, false, T2m);  // (barriers inside: s_w[3] may be reused below)
                if (tid == 0) s_w[3] = 0;
                __syncthreads();
                for (uint32_t e = tid; e < cr; e += kSelThreads) {
                    const unsigned long long ent = LR[e];
                    const bool pr = (uint32_t)(ent >> 32) >= Tc;
                    const uint32_t pos = wave_append(pr, &s_w[3]);
                    if (pr && base + pos < p.cand_cap) out[base + pos] = (uint32_t)(ent & 0xFFFFFFFFull);
                }
                return;
            }
            for (uint32_t e = tid; e < cr; e += kSelThreads) {
                const unsigned long long ent = LR[e];
                const bool pr = (uint32_t)(ent >> 32) >= Tc;
                const uint32_t pos = wave_append(pr, &s_w[3]);
                if (pr && pos < p.cand_cap) out[pos] = (uint32_t)(ent & 0xFFFFFFFFull);
            }
            __syncthreads();
            if (tid == 0) {
                const uint32_t c = s_w[3];
                QState st;
                st.n_valid = vw;
                st.thr_key = Tc;
                st.overflow = c > p.cand_cap ? 1u : 0u;
                st.cand_count = c > p.cand_cap ? 0u : c;
                p.qstate[q] = st;
                if (p.count_overflows && p.half_stats && st.overflow) atomicAdd(p.half_stats + 1, 1u);
                if (p.l2_hint && q == 0) {  // the threshold DISTANCE of this selection (score 1 / (1 + d)): qprep's estimator choice
                    const float tau = key_to_score(Tc);
                    *p.l2_hint = (tau > 0.0f && tau <= 1.0f) ? 1.0f / tau - 1.0f : 0.0f;
                }
#ifdef NMN_SELECT_TRACE
                if (q == 0) {
                    sel_t[8] = wall_clock64();
                    printf("select W=%u vw=%u nA=%u ct=%u cr=%u cand=%u | ticks: load %llu count %llu radixW %llu tiles %llu radixT %llu rows %llu radixR %llu out %llu total %llu\n",
                           W, vw, nA, ct, cr, c, sel_t[1] - sel_t[0], sel_t[2] - sel_t[1], sel_t[3] - sel_t[2], sel_t[4] - sel_t[3], sel_t[5] - sel_t[4],
                           sel_t[6] - sel_t[5], sel_t[7] - sel_t[6], sel_t[8] - sel_t[7], sel_t[8] - sel_t[0]);
                }
#endif
            }
            done = true;
        }
    }
    if (done) return;
    if (S > 1) {  // (a row list that overflowed)
        (void)split_reserve_and_finish(0u, true, Twm);
        return;
    }

    // ---- generic path (a compact list overflowed) ------------------------------------------------
    __syncthreads();
    if (flat) {  // (block-uniform; the flat path never loaded the wave level: Twm — its tile-level threshold — gates waves as well)
        (void)load_wave_maxima();
        __syncthreads();
    }
    if (ct > kCompCap) {
        // tile-level bound from the un-compacted tile keys of every valid wave
        if (tid == 0) s_w[1] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < W; i += kSelThreads)
            if (wk[i] != kKeyMasked && wk[i] >= Twm) la[atomicAdd(&s_w[1], 1u)] = i;
        __syncthreads();
        const uint32_t slots = s_w[1] * tpw;
        auto tile_key = [&](uint32_t e) -> uint32_t {
            const uint32_t t = tile_of(la[e / tpw], e % tpw);
            const uint32_t key = t < n_tiles ? tmax[t] : kKeyMasked;
            return key >= Twm ? key : kKeyMasked;
        };
        const uint32_t vt = count_valid(tile_key, slots, &s_w[2]);
        if (vt > k) Tc = max(margin_key(radix2(tile_key, slots, k, hist, &pick), qi), skip);
    }
    // collection: waves -> tiles -> rows, all >= Tc
    if (tid == 0) { s_w[0] = 0; s_w[1] = 0; s_w[2] = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < W; i += kSelThreads)
        if (wk[i] != kKeyMasked && wk[i] >= Tc) la[atomicAdd(&s_w[0], 1u)] = i;
    __syncthreads();
    {
        const uint32_t slots = s_w[0] * tpw;
        for (uint32_t e0 = tid; e0 < slots; e0 += kSelThreads * V) {
            uint32_t tk[V], tt[V];
#pragma unroll
            for (int u = 0; u < V; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                tt[u] = e < slots ? tile_of(la[e / tpw], e % tpw) : 0xFFFFFFFFu;
                tk[u] = tt[u] < n_tiles ? tmax[tt[u]] : kKeyMasked;
            }
#pragma unroll
            for (int u = 0; u < V; u++) {
                const bool pr = tk[u] != kKeyMasked && tk[u] >= Tc;
                const uint32_t pos = wave_append(pr, &s_w[1]);
                if (pr && pos < kListCap) lb[pos] = tt[u];
            }
        }
    }
    __syncthreads();
    const uint32_t nB_raw = s_w[1];
    const uint32_t nB = min(nB_raw, kListCap);
    {
        const uint32_t tot = nB * kTileRows;
        for (uint32_t e0 = tid; e0 < tot; e0 += kSelThreads * V) {
            uint32_t kb[V];
#pragma unroll
            for (int u = 0; u < V; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                kb[u] = e < tot ? score_bits((uint64_t)lb[e >> 6] * kTileRows + (e & 63u)) : kScoreSentinelBits;
            }
#pragma unroll
            for (int u = 0; u < V; u++) {
                const uint32_t e = e0 + (uint32_t)u * kSelThreads;
                const uint32_t key = bits_to_key(kb[u]);
                const bool pr = e < tot && key != kKeyMasked && key >= Tc;
                const uint32_t pos = wave_append(pr, &s_w[2]);
                if (pr && pos < p.cand_cap) out[pos] = lb[e >> 6] * kTileRows + (e & 63u);
            }
        }
    }
    __syncthreads();
    const uint32_t c = s_w[2];
    const bool over = c > p.cand_cap || nB_raw > kListCap;  // > 4096 passing tiles means > 4096 candidates
    // An overflow on the bf16 mirror is normally answered by the f32 retry sweep (a narrower margin).  Not when more than
    // kListCap TILES hold a row with the very same top key: approximate scores that agree bit for bit in thousands of tiles
    // belong to copies of one row, and copies tie in f32 as well — the retry would read the whole corpus to overflow again
    // (10M identical rows: 5.1 ms of 14).  Such a query is flagged 4: the retry sweep and the crowd kernels skip it, the
    // retry's selection turns it back into 1, the exact fallback answers it.
    bool hopeless = false;
    if (over && p.retry_follows) {  // (block-uniform)
        __syncthreads();
        if (tid == 0) { s_w[0] = 0; s_w[3] = 0; }
        __syncthreads();
        uint32_t top = 0;
        for (uint32_t i = tid; i < W; i += kSelThreads)
            if (wk[i] != kKeyMasked) top = max(top, wk[i]);
        atomicMax(&s_w[3], top);
        __syncthreads();
        top = s_w[3];
        uint32_t ties = 0;
        const uint32_t slots = W * tpw;
        for (uint32_t e = tid; e < slots; e += kSelThreads) {
            const uint32_t i = e / tpw, t = tile_of(i, e % tpw);
            if (wk[i] == top && t < n_tiles && tmax[t] == top) ties++;
        }
        if (ties) atomicAdd(&s_w[0], ties);
        __syncthreads();
        hopeless = s_w[0] > kListCap;
    }
    if (tid == 0) {
        QState st;
        st.n_valid = vw;
        st.thr_key = Tc;
        st.overflow = over ? (hopeless ? 4u : 1u) : 0u;
        st.cand_count = over ? 0u : c;
        p.qstate[q] = st;
        if (p.count_overflows && p.half_stats && over) atomicAdd(p.half_stats + 1, 1u);
        if (p.l2_hint && q == 0) {
            const float tau = key_to_score(Tc);
            *p.l2_hint = (tau > 0.0f && tau <= 1.0f) ? 1.0f / tau - 1.0f : 0.0f;
        }
    }
}

hipError_t launch_select(const SelectParams& p, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSelectLds);
    if (e != hipSuccess) return e;
    static const bool no_flat = getenv("NMN_NO_FLAT_SELECT") != nullptr;  // (A/B switch of the flat path)
    static const bool no_split = getenv("NMN_NO_SPLIT_SELECT") != nullptr;  // (A/B switch of the split selection)
    SelectParams pf = p;
    pf.flat = no_flat ? 0 : 1;
    // split: lone callers (<= 4 queries: the parts of ALL queries must be resident together, and a batch's selections already run
    // side by side), k <= 256 (the group bound), 16 385 .. 16 x 16 384 tiles, 16-byte aligned rows of tmax, no extra rank
    const uint32_t parts = (p.n_tiles + kFlatTiles - 1) / kFlatTiles;
    pf.split = (!no_flat && !no_split && p.split_sg && p.split_ctr && p.nq <= 4 && p.k <= kFlatGroupK && !p.k_extra && parts >= 2 && parts <= 16 &&
                (p.tmax_stride & 3ull) == 0ull) ? parts : 0u;
    hipLaunchKernelGGL(select_kernel, dim3(p.nq * (pf.split ? pf.split : 1u)), dim3(kSelThreads), kSelectLds, s, pf);
    return hipGetLastError();
}

// ---- crowd path: the rows within the margin, when there are more than cand_cap of them ---------------
// Walk of the tiles of query q whose maximum reaches Tc (64 tile maxima per wave and step, then the qualifying tiles one
// by one, 64 rows each): f(row, pred) for every row of such a tile, pred = its approximate key >= Tc.  Tiles below Tc
// hold no candidate (and, on the matrix-core path, possibly no scores at all: Tc >= skip_key).
template <class F>
__device__ __forceinline__ void crowd_walk(const CrowdParams& p, uint32_t q, uint32_t Tc, F&& f) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t* tmax = p.tmax + (uint64_t)q * p.tmax_stride;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t t0 = wave * 64; t0 < p.n_tiles; t0 += n_waves * 64) {
        const uint64_t t = t0 + lane;
        const uint32_t tk = t < p.n_tiles ? tmax[t] : kKeyMasked;
        unsigned long long hot = __ballot(tk != kKeyMasked && tk >= Tc);
        while (hot) {
            const int b = __builtin_ctzll(hot);
            hot &= hot - 1;
            const uint64_t row = (t0 + (uint64_t)b) * kTileRows + lane;
            const uint32_t key = row < p.n_rows ? bits_to_key(p.scores[score_at(row, q, p.nql)]) : kKeyMasked;
            f(row, key != kKeyMasked && key >= Tc);
        }
    }
}

__global__ __launch_bounds__(256) void crowd_count_kernel(CrowdParams p) {
    const uint32_t q = blockIdx.y;
    const QState st = p.qstate[q];
    if (st.overflow != 1) return;
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    crowd_walk(p, q, st.thr_key, [&](uint64_t, bool pred) {
        const unsigned long long m = __ballot(pred);
        if ((threadIdx.x & 63u) == 0) mine += (uint32_t)__builtin_popcountll(m);
    });
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        p.wg_count[(size_t)q * gridDim.x + blockIdx.x] = s_cnt;  // crowd_fill_kernel (same grid, same walk) starts its slice part here
        if (s_cnt) {
            // saturating: a count beyond the pool is just "too many"
            const uint32_t old = atomicAdd(&p.count[q], s_cnt);
            if (old + s_cnt < old) p.count[q] = 0xFFFFFFFFu;
        }
    }
}

// Every workgroup writes its rows into ITS part of the query's slice: the parts' starts are the prefix sums of the counts
// crowd_count_kernel left per workgroup (same grid, same walk, hence the same rows).  One append cursor per QUERY in global
// memory — one atomic per hot tile, all on one address — made this kernel 115 us at 10 000 tiles, ten times the count.
// The slices of the pool (first come first served in query order) are worked out HERE, by every workgroup for itself from the
// read-only totals — the one-workgroup launch that used to sit between count and fill is gone, and with it 4.4 us of every
// search (a dependent launch costs that much even when it returns at once).  select_kernel zeroes the totals.
__global__ __launch_bounds__(256) void crowd_fill_kernel(CrowdParams p) {
    const uint32_t q = blockIdx.y;
    {
        const uint32_t ov = p.qstate[q].overflow;
        if (ov != 1 && ov != 2) return;  // (2: workgroup 0 of this query has already published the decision made below)
    }
    __shared__ uint32_t s_part[256], s_cur, s_off, s_total;
    {
        uint32_t c = 0;
        if (threadIdx.x <= q) {  // nq <= 256
            c = p.count[threadIdx.x];
            const uint32_t o = p.qstate[threadIdx.x].overflow;
            if (o != 1 && o != 2) c = 0;
            // a threshold that lets more than an eighth of the shard through is not a crowd around the query but a useless
            // margin (one row of enormous norm under a Euclidean metric): that is the f32 retry's case
            if ((uint64_t)c * 8u > p.n_rows) c = 0;
        }
        s_part[threadIdx.x] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t cursor = 0, off = 0xFFFFFFFFu;
            for (uint32_t i = 0; i <= q && i < 256u; i++) {
                const uint32_t ci = s_part[i];
                if (ci != 0 && ci <= p.pool_cap - cursor) {
                    if (i == q) off = cursor;
                    cursor += ci;
                }
            }
            s_off = off;
            s_total = s_part[q < 256u ? q : 255u];
        }
        __syncthreads();
    }
    const uint32_t off = s_off, total = s_total;
    if (off == 0xFFFFFFFFu) return;  // no slice: the query stays an ordinary overflow (f32 retry / exact fallback)
    __syncthreads();
    {
        const uint32_t* wc = p.wg_count + (size_t)q * gridDim.x;
        uint32_t acc = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) acc += wc[i];
        s_part[threadIdx.x] = acc;
        __syncthreads();
        for (uint32_t o2 = 128; o2 > 0; o2 >>= 1) {
            if (threadIdx.x < o2) s_part[threadIdx.x] += s_part[threadIdx.x + o2];
            __syncthreads();
        }
        if (threadIdx.x == 0) s_cur = s_part[0];
        __syncthreads();
    }
    uint32_t* dst = p.pool_rows + off;
    const uint32_t thr = p.qstate[q].thr_key;
    crowd_walk(p, q, thr, [&](uint64_t row, bool pred) {
        const uint32_t pos = wave_append(pred, &s_cur);  // (LDS: the workgroup's four waves)
        if (pred && pos < total) dst[pos] = (uint32_t)row;
    });
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // what rescore / final read
        p.offset[q] = off;
        p.qstate[q].cand_count = total;
        __threadfence();
        p.qstate[q].overflow = 2;
    }
}

hipError_t launch_crowd_collect(const CrowdParams& p, hipStream_t s) {
    // ~4096 workgroups in all: with many queries in the pass the (normally empty) launches stay cheap
    const uint32_t gx_cap = std::max<uint32_t>(8, std::min<uint32_t>(kCrowdMaxGrid, 4096 / std::max<uint32_t>(p.nq, 1)));
    const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>((p.n_tiles + 64 * 4 - 1) / (64 * 4), gx_cap));
    hipLaunchKernelGGL(crowd_count_kernel, dim3(gx, p.nq), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crowd_fill_kernel, dim3(gx, p.nq), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- exact-fallback selection (device function of final_kernel) ----------------------------------
// For a query whose candidate list overflowed, rescore_kernel has replaced scores[] by the EXACT score
// of every row.  Composite keys (score key << 32 | ~row) are unique, so the k-th largest composite is
// well defined and {composite >= it} is exactly the (score desc, row asc) top-k whatever the number of
// ties: a 64-bit radix select, 11+11+10 bits of score key, then (only if the ties at the k-th score
// straddle it) 11+11+10 bits of ~row.  One workgroup walks all rows up to six times: slow by design,
// this path only runs when more than cand_cap rows sit within the rounding margin of the k-th score.
// One walk of a query's exact scores by the whole workgroup: thread t takes rows 4t..4t+3 of every chunk of
// 4096 rows (one 16-byte load: the four rows sit together in the tile-major layout), four chunks in flight per
// thread.  f(row, key) runs once per row slot in [0, round_up(n_pad, 4096)), key = kKeyMasked outside the shard
// or the filter; every lane of every wave makes the same number of calls (wave_append inside f is legal).
template <class F>
__device__ __forceinline__ void walk_scores(const uint32_t* __restrict__ scores, uint32_t q, uint32_t nql,
                                            uint64_t n_pad, F&& f) {
    constexpr uint64_t kChunk = 4ull * kSelThreads;
    const uint64_t t4 = 4ull * threadIdx.x;
    uint64_t base = 0;
    for (; base + 4 * kChunk <= n_pad; base += 4 * kChunk) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            v[u] = *reinterpret_cast<const uint4*>(scores + score_at(base + u * kChunk + t4, q, nql));
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + u * kChunk + t4;
            f(i, bits_to_key(v[u].x));
            f(i + 1, bits_to_key(v[u].y));
            f(i + 2, bits_to_key(v[u].z));
            f(i + 3, bits_to_key(v[u].w));
        }
    }
    for (; base < n_pad; base += kChunk) {
        const uint64_t i = base + t4;
        uint4 v = make_uint4(kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits);
        if (i < n_pad) v = *reinterpret_cast<const uint4*>(scores + score_at(i, q, nql));  // n_pad % 64 == 0
        f(i, bits_to_key(v.x));
        f(i + 1, bits_to_key(v.y));
        f(i + 2, bits_to_key(v.z));
        f(i + 3, bits_to_key(v.w));
    }
}

// the exact scores of every row (exact fallback) ...
__device__ uint32_t exact_select_into(const uint32_t* __restrict__ scores, uint32_t q, uint32_t nql, uint64_t n_rows,
                                      uint32_t k, unsigned long long* list, uint32_t* hist, PickResult* pick,
                                      uint32_t* s_misc) {
    const uint64_t n_pad = (n_rows + 63) & ~63ull;
    return exact_select_walk([&](auto&& f) { walk_scores(scores, q, nql, n_pad, f); }, k, list, hist, pick, s_misc);
}
// ... or of the rows of a crowd slice
__device__ uint32_t crowd_select_into(const uint32_t* __restrict__ rows, const float* __restrict__ scores, uint32_t n,
                                      uint32_t k, unsigned long long* list, uint32_t* hist, PickResult* pick,
                                      uint32_t* s_misc) {
    const uint32_t n_round = (n + kSelThreads - 1) / kSelThreads * kSelThreads;
    return exact_select_walk(
        [&](auto&& f) {
            for (uint32_t e = threadIdx.x; e < n_round; e += kSelThreads) {
                const bool in = e < n;
                f(in ? rows[e] : 0u, in ? score_to_key(scores[e]) : kKeyMasked);
            }
        },
        k, list, hist, pick, s_misc);
}

// ---- exact-fallback selection on the WHOLE device (large shards) ----------------------------------------------------------
// The same 64-bit radix select as exact_select_walk, by kFbGrid workgroups instead of one: a query whose neighbourhood
// defeats every margin (all rows identical, a degenerate corpus) used to cost one compute unit seven walks over all its
// exact scores — ~10 ms at 10M rows after the 7 ms exact scan.  Here every workgroup histograms its slice of the scores in
// LDS, adds the non-empty bins to a global histogram, meets the others at a grid barrier, and then each workgroup picks the
// digit for itself from the global histogram (identical arithmetic: no second barrier); at most six digits, then one pass
// appends the composites >= the k-th to the query's list, which final_kernel sorts.  One launch per search on shards of
// >= 2^18 rows, returning at once when no query is flagged; the grid (64 workgroups: a quarter of the compute units, so that launches of several streams fit side by side) is co-resident by
// construction, the barrier is a monotonic counter that is never reset (a launch starts at the multiple of the grid size
// the previous one left it at).
constexpr uint32_t kFbGrid = 64;  // eight such launches (eight streams) stay co-resident: 1024 threads each, 2048 per CU

// The grid barrier.  sync[0] counts arrivals and is ZEROED before every launch (by select_kernel, which precedes this
// kernel on the same stream, or by a memset on the large-k path), sync[1] is the launch's abort flag.  Nothing but an
// idle device guarantees that the 64 workgroups are resident together: up to NMN_MAX_SHARDS logical shards, each with a
// stream of its own, may flag a query at the same moment, and partially resident grids that wait for each other would
// hang the device.  So the wait is BOUNDED (FallbackParams::timeout_ticks of the 100 MHz wall clock, ~100 ms): the
// first workgroup to run out of patience raises the abort flag, every workgroup leaves at its next look, the queries
// not finished keep overflow == 1 and final_kernel selects them with one workgroup (slower, same answer).  The large-k
// path, which has no such second line, launches this kernel cooperatively (co-residency guaranteed by the runtime).
__device__ __forceinline__ bool fb_grid_barrier(unsigned long long* sync, unsigned long long* target, unsigned long long timeout,
                                                uint32_t* s_abort) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // this workgroup's histogram adds / list appends are visible device-wide before it arrives
        atomicAdd(sync, 1ull);
        *target += kFbGrid;
        const unsigned long long t0 = wall_clock64();
        uint32_t bad = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < *target) {
            if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
                bad = 1;
                break;
            }
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(sync + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *s_abort = bad;
        __threadfence();
    }
    __syncthreads();
    return *s_abort == 0;
}

__global__ void __launch_bounds__(kSelThreads) fallback_select_kernel(FallbackParams p) {
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_target;
    __shared__ uint32_t s_abort;
    const uint32_t tid = threadIdx.x, wg = blockIdx.x;
    // anything to do?  (qstate was written by earlier kernels of this stream: every workgroup sees the same flags)
    bool any = p.all != 0;
    for (uint32_t q = 0; q < p.nq && !any; q++) any = p.qstate[q].overflow == 1u;
    if (!any) return;
    if (tid == 0) {
        s_target = 0ull;  // the counter was zeroed before this launch
        s_abort = 0u;
    }
    __syncthreads();
    const unsigned long long patience = p.timeout_ticks ? p.timeout_ticks : 10000000ull;  // 100 ms of the 100 MHz wall clock
    const uint64_t n_pad = (p.n_rows + 63) & ~63ull;
    const uint64_t n_tiles = n_pad / 64, tiles_per = (n_tiles + kFbGrid - 1) / kFbGrid;
    const uint64_t i0 = min((uint64_t)wg * tiles_per, n_tiles) * 64, i1 = min((uint64_t)(wg + 1) * tiles_per, n_tiles) * 64;
    auto comp = [](uint64_t i, uint32_t key) -> unsigned long long {
        return key == kKeyMasked ? 0ull : (((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
    };
    const int shifts[6] = {53, 42, 32, 21, 10, 0};
    const int widths[6] = {11, 11, 10, 11, 11, 10};
    const uint32_t list_cap = p.list_cap ? p.list_cap : (uint32_t)NMN_MAX_TOP_K;
    for (uint32_t q = 0; q < p.nq; q++) {
        if (!p.all && p.qstate[q].overflow != 1u) continue;
        // zero the global histograms and counters of this query's run, then meet
        for (uint32_t b = wg * kSelThreads + tid; b < 6u * kBins + 2u; b += kFbGrid * kSelThreads) p.ghist[b] = 0u;
        if (!fb_grid_barrier(p.sync, &s_target, patience, &s_abort)) return;
        uint32_t* const g_rows = p.ghist + 6 * kBins;      // participating rows
        uint32_t* const g_fill = p.ghist + 6 * kBins + 1;  // entries appended to the list
        unsigned long long prefix = 0ull;
        uint32_t need = 0, kk = 0;
        bool empty = false;
        for (int d = 0; d < 6; d++) {
            const int nb = 1 << widths[d];
            for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            const int hi_shift = shifts[d] + widths[d];
            uint32_t loc = 0;
            for (uint64_t i = i0 + 4ull * tid; i < i1; i += 4ull * kSelThreads) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.scores + score_at(i, q, p.nql));
                const uint32_t kv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned long long c = comp(i + e, bits_to_key(kv[e]));
                    if (c != 0ull) loc++;
                    const bool in = c != 0ull && !(hi_shift < 64 && (c >> hi_shift) != (prefix >> hi_shift));
                    const uint32_t bin = (uint32_t)(c >> shifts[d]) & (uint32_t)(nb - 1);
                    hist_add_wave(hist, in, bin);  // (one LDS atomic for all the lanes that share a bin: nmn_select_dev.h)
                }
            }
            if (d == 0 && loc) atomicAdd(&s_cnt, loc);
            __syncthreads();
            for (int b = tid; b < nb; b += kSelThreads)
                if (hist[b]) atomicAdd(&p.ghist[d * kBins + b], hist[b]);
            if (d == 0 && tid == 0 && s_cnt) atomicAdd(g_rows, s_cnt);
            if (!fb_grid_barrier(p.sync, &s_target, patience, &s_abort)) return;
            // every workgroup picks the digit from the (now complete) global histogram
            for (int b = tid; b < kBins; b += kSelThreads)
                hist[b] = b < nb ? __hip_atomic_load(&p.ghist[d * kBins + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __syncthreads();
            if (d == 0) {
                const uint32_t rows = __hip_atomic_load(g_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                kk = min(p.k, rows);
                need = kk;
                if (kk == 0) {
                    empty = true;
                    break;
                }
            }
            pick_bin(hist, nb, need, &pick);
            __syncthreads();
            prefix |= (unsigned long long)pick.bin << shifts[d];
            need -= pick.above;
            const bool done = d == 2 && hist[pick.bin] == need;  // the score key is fixed and every row holding it is wanted
            __syncthreads();
            if (done) break;
        }
        if (!empty) {
            // collect: composites >= prefix (exactly kk of them device-wide)
            for (uint64_t i = i0 + 4ull * tid; i < ((i1 - i0 + 4ull * kSelThreads - 1) / (4ull * kSelThreads)) * (4ull * kSelThreads) + i0;
                 i += 4ull * kSelThreads) {
                uint4 v = make_uint4(kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits);
                if (i < i1) v = *reinterpret_cast<const uint4*>(p.scores + score_at(i, q, p.nql));
                const uint32_t kv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned long long c = comp(i + e, bits_to_key(kv[e]));
                    const bool pred = c != 0ull && c >= prefix;
                    const uint32_t pos = wave_append(pred, g_fill);
                    if (pred && pos < list_cap) p.list[(size_t)q * list_cap + pos] = c;
                }
            }
        }
        // the list is complete (and the histograms may be reused by the next query)
        if (!fb_grid_barrier(p.sync, &s_target, patience, &s_abort)) return;
        if (wg == 0 && tid == 0) {
            const uint32_t got = empty ? 0u : __hip_atomic_load(g_fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.list_count[q] = min(got, list_cap);
            if (!p.all) p.qstate[q].overflow = 3u;  // final_kernel: the list is ready
        }
    }
}

static unsigned long long fb_timeout_ticks() {  // NMN_FB_TIMEOUT_TICKS: the test of the abort path sets it to 1
    static const unsigned long long v = [] {
        const char* e = getenv("NMN_FB_TIMEOUT_TICKS");
        const long long x = e ? atoll(e) : 0;
        return (unsigned long long)(x > 0 ? x : 0);
    }();
    return v;
}

hipError_t launch_fallback_select(const FallbackParams& p_in, hipStream_t s) {
    FallbackParams p = p_in;
    if (!p.timeout_ticks) p.timeout_ticks = fb_timeout_ticks();
    if (p.all) {
        // no second line behind this selection (the caller sorts what it selected): the counters are zeroed here and the
        // grid is launched cooperatively — resident together or not at all (an error the caller answers with the full sort)
        hipError_t e = hipMemsetAsync(p.sync, 0, 16, s);
        if (e != hipSuccess) return e;
        p.timeout_ticks = ~0ull >> 1;
        void* args[] = {&p};
        return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fallback_select_kernel), dim3(kFbGrid), dim3(kSelThreads), args, 0, s);
    }
    hipLaunchKernelGGL(fallback_select_kernel, dim3(kFbGrid), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

// ---- final sort: candidates by (exact score desc, row asc) -> top-k ---------------------------
__global__ void __launch_bounds__(kSelThreads) final_kernel(FinalParams p) {
    __shared__ unsigned long long list[NMN_MAX_TOP_K];
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_misc[2];
    const uint32_t q = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    uint32_t n;
    const uint32_t mode = p.qstate[q].overflow;
    // (a polling host caller: this workgroup's result stores — all threads' — are ordered before its arrival, the last arrival
    //  publishes the sequence word)
    auto publish_done = [&]() {
        if (!p.done_word) return;
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = atomicAdd(p.done_ctr, 1u);
            if (t == gridDim.x - 1u) {
                *p.done_ctr = 0u;
                __threadfence_system();
                __hip_atomic_store(p.done_word, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    if (mode && p.short_chain) {  // (block-uniform) the short chain: nobody computed what the lists below would need — the host follows up
        if (tid == 0) p.out_counts[q] = 0xFFFFFFFFu;
        publish_done();
        return;
    }
    if (mode == 2) {
        const uint32_t off = p.crowd_offset[q];
        n = crowd_select_into(p.crowd_rows + off, p.crowd_scores + off, p.qstate[q].cand_count, p.k, list, hist, &pick, s_misc);
    } else if (mode == 3) {  // exact fallback, selected by the device-wide radix select (fallback_select_kernel)
        n = min(p.fb_count[q], (uint32_t)NMN_MAX_TOP_K);
        for (uint32_t i = tid; i < n; i += kSelThreads) list[i] = p.fb_list[(size_t)q * NMN_MAX_TOP_K + i];
        if (tid == 0) {
            p.qstate[q].cand_count = n;
            p.qstate[q].overflow = 1u;  // what the statistics report: this query took the exact fallback
        }
    } else if (mode) {
        n = exact_select_into(p.scores, q, p.nql, p.n_rows, p.k, list, hist, &pick, s_misc);
        if (tid == 0) p.qstate[q].cand_count = n;
    } else {
        n = min(p.qstate[q].cand_count, min(p.cand_cap, (uint32_t)NMN_MAX_TOP_K));
        for (uint32_t i = tid; i < n; i += kSelThreads) {
            const uint32_t row = p.cand_rows[(size_t)q * p.cand_cap + i];
            const uint32_t key = score_to_key(p.cand_scores[(size_t)q * p.cand_cap + i]);
            list[i] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - row);
        }
    }
    __syncthreads();
    // More candidates than threads (an IVF probe of clustered lists, a margin that lets a few thousand rows through) and k below
    // that: sort_and_emit would take the workgroup-wide bitonic network over 2048 / 4096 slots — 36 us of an IVF probe's 173
    // (profiles/r05q_*).  Only the k best are wanted: a two-digit radix pick over the candidates' keys gives a key T with at least k
    // candidates at or above it (and, ties aside, few more); those are compacted and sorted by runs and ranks like any short list.
    // More than one per thread still at or above T (thousands of equal scores): the network, as before.
    if (mode == 0 && n > (uint32_t)kSelThreads && p.k < (uint32_t)kSelThreads) {  // (block-uniform)
        __shared__ unsigned long long top[kSelThreads];
        const uint32_t T = radix2([&](uint32_t e) { return (uint32_t)(list[e] >> 32); }, n, p.k, hist, &pick);
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        for (uint32_t b0 = tid & ~63u; b0 < n; b0 += kSelThreads) {  // (on the wave's first index: wave_append needs whole waves)
            const uint32_t i = b0 + (tid & 63u);
            const unsigned long long v = i < n ? list[i] : 0ull;
            const bool pr = i < n && (uint32_t)(v >> 32) >= T;
            const uint32_t pos = wave_append(pr, &s_misc[0]);
            if (pr && pos < (uint32_t)kSelThreads) top[pos] = v;
        }
        __syncthreads();
        const uint32_t c = s_misc[0];
        if (c <= (uint32_t)kSelThreads) {  // (>= k by construction)
            if (tid < c) list[tid] = top[tid];
            n = c;
        }
        __syncthreads();
    }
    sort_and_emit(list, n, n, p.k, p.row_base, p.out_rows + (size_t)q * p.k, p.out_scores + (size_t)q * p.k, p.out_counts + q);
    publish_done();
}

hipError_t launch_final(const FinalParams& p, hipStream_t s) {
    hipLaunchKernelGGL(final_kernel, dim3(p.nq), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

// ---- shard merge (merge_top_k, distributed.rs:413-433) ----------------------------------------
// Inputs [list][query][k] are each already in final order; rows are unique across shards, so
// (key desc, row asc) is a strict total order and an element's output slot is simply the number of
// elements of all lists that precede it (binary search per list).
__device__ __forceinline__ bool hit_before(uint32_t ka, uint64_t ra, uint32_t kb, uint64_t rb) {
    return ka > kb || (ka == kb && ra < rb);
}

struct MergeView {
    const char* rows;
    const char* scores;
    const char* counts;
    uint64_t stride_r, stride_s, stride_c;  // bytes between consecutive lists
    __device__ const uint64_t* R(uint32_t l) const { return reinterpret_cast<const uint64_t*>(rows + l * stride_r); }
    __device__ const float* S(uint32_t l) const { return reinterpret_cast<const float*>(scores + l * stride_s); }
    __device__ const uint32_t* C(uint32_t l) const { return reinterpret_cast<const uint32_t*>(counts + l * stride_c); }
};

__global__ void __launch_bounds__(256) merge_kernel(MergeView v, uint32_t n_lists, uint32_t nq, uint32_t k,
                                                    uint64_t* __restrict__ out_rows, float* __restrict__ out_scores,
                                                    uint32_t* __restrict__ out_counts) {
    const uint32_t q = blockIdx.x;
    uint32_t total = 0;
    for (uint32_t l = 0; l < n_lists; l++) total += min(v.C(l)[q], k);
    const uint32_t cnt = min(total, k);
    for (uint32_t e = threadIdx.x; e < n_lists * k; e += blockDim.x) {
        const uint32_t l = e / k, i = e - l * k;
        const uint32_t cl = min(v.C(l)[q], k);
        if (i >= cl) continue;
        const size_t base = (size_t)q * k;
        const float sc = v.S(l)[base + i];
        const uint32_t key = score_to_key(sc);
        const uint64_t row = v.R(l)[base + i];
        uint32_t rank = i;
        for (uint32_t m = 0; m < n_lists; m++) {
            if (m == l) continue;
            const uint32_t cm = min(v.C(m)[q], k);
            const uint64_t* rm = v.R(m) + base;
            const float* sm = v.S(m) + base;
            uint32_t lo = 0, hi = cm;  // first index in list m that does NOT precede e
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (hit_before(score_to_key(sm[mid]), rm[mid], key, row)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_rows[(size_t)q * k + rank] = row;
            out_scores[(size_t)q * k + rank] = sc;
        }
    }
    for (uint32_t i = cnt + threadIdx.x; i < k; i += blockDim.x) {
        out_rows[(size_t)q * k + i] = UINT64_MAX;
        out_scores[(size_t)q * k + i] = u2f(0xFF800000u);
    }
    if (threadIdx.x == 0) out_counts[q] = cnt;
}

hipError_t launch_merge(const uint64_t* rows, const float* scores, const uint32_t* counts, uint64_t list_stride_bytes,
                        uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* out_rows, float* out_scores,
                        uint32_t* out_counts, hipStream_t s) {
    MergeView v;
    v.rows = reinterpret_cast<const char*>(rows);
    v.scores = reinterpret_cast<const char*>(scores);
    v.counts = reinterpret_cast<const char*>(counts);
    v.stride_r = list_stride_bytes ? list_stride_bytes : (uint64_t)nq * k * 8;
    v.stride_s = list_stride_bytes ? list_stride_bytes : (uint64_t)nq * k * 4;
    v.stride_c = list_stride_bytes ? list_stride_bytes : (uint64_t)nq * 4;
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(256), 0, s, v, n_lists, nq, k, out_rows, out_scores, out_counts);
    return hipGetLastError();
}

// ---- count of rows above / equal to a score (certificate) -------------------------------------
__global__ void __launch_bounds__(256) count_cmp_kernel(const uint32_t* __restrict__ scores, uint64_t n_rows,
                                                        uint32_t ref_key, unsigned long long* __restrict__ out2) {
    unsigned long long gt = 0, eq = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = bits_to_key(scores[i]);
        if (key == kKeyMasked) continue;
        gt += key > ref_key;
        eq += key == ref_key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        gt += __shfl_down(gt, off);
        eq += __shfl_down(eq, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (gt) atomicAdd(&out2[0], gt);
        if (eq) atomicAdd(&out2[1], eq);
    }
}

// rows whose stored f32 magnitude is outside the range in which the approximate cosine can be trusted for the
// f64 artifact similarity (same test as scan_kernel's epilogue); masked-out rows may be counted too — a larger
// rank only lowers the threshold
__global__ __launch_bounds__(256) void count_untrusted_kernel(const float* __restrict__ norms, uint64_t n_rows,
                                                              uint32_t* __restrict__ out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (uint64_t)gridDim.x * 256) {
        const float vn = norms[i];
        c += (vn >= 1e-15f && vn <= 1e18f) ? 0u : 1u;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    __shared__ uint32_t ws[4];
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (ws[0] + ws[1] + ws[2] + ws[3])) atomicAdd(out, ws[0] + ws[1] + ws[2] + ws[3]);
}

hipError_t launch_count_untrusted(const float* norms, uint64_t n_rows, uint32_t* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, 4, s);
    if (e != hipSuccess || n_rows == 0) return e;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_rows + 255) / 256, 1024);
    hipLaunchKernelGGL(count_untrusted_kernel, dim3(blocks), dim3(256), 0, s, norms, n_rows, out);
    return hipGetLastError();
}

hipError_t launch_count_cmp(const uint32_t* scores, uint64_t n_rows, float score, unsigned long long* out2,
                            hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    uint64_t blocks = (n_rows + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(count_cmp_kernel, dim3((unsigned)blocks), dim3(256), 0, s, scores, n_rows,
                       score_to_key(score), out2);
    return hipGetLastError();
}

}  // namespace nmn
