// nmn_scan.hip — the HBM-bound kernel of the SIMILAR TOP-K path: one streaming pass over the
// row-major f32 corpus producing an APPROXIMATE score per row (any summation order, FMA allowed)
// plus the maximum score key of every 64-row tile.  Exactness is restored later by nmn_exact.hip.
//
// Replaces the per-key hot loop of the reference (vector_engine/src/lib.rs:2115-2228: store.get ->
// extract_vector -> compute_score per key) with one resident-matrix sweep.
//
// Mapping (wave64, CDNA4): a wave owns whole 64-row tiles.  Inside a tile it takes 16 steps of 4
// rows; in a step each 16-lane DPP row of the wave reads ONE corpus row with 16-byte loads (lane j
// reads float4 columns j, j+16, j+32, ... -> every load instruction covers 4 rows x 256 contiguous
// bytes = whole 128-B lines), multiplies against the query tile staged in LDS (the 4 DPP rows read
// the same LDS addresses, i.e. broadcast), and reduces across the 16 lanes with four row_ror DPP
// adds.  Lane L keeps the dot of tile row (L&15)*4 + (L>>4), so after 16 steps the wave finishes
// 64 scores at once: one permuted-but-contiguous 256-B norm load, one 256-B score store, one wave
// max.  Nothing is read twice; algorithmic bytes = rows * dim * 4.
#include <algorithm>

#include "nmn_internal.h"

namespace nmn {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// all-reduce over a 16-lane DPP row (row_ror 8,4,2,1)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x128>(v);
    v += dpp_f<0x124>(v);
    v += dpp_f<0x122>(v);
    v += dpp_f<0x121>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_u<0x128>(v));
    v = max(v, dpp_u<0x124>(v));
    v = max(v, dpp_u<0x122>(v));
    v = max(v, dpp_u<0x121>(v));
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

template <bool NT>
__device__ __forceinline__ v4f load4(const v4f* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// dot (or squared-difference sum) of ONE corpus row against NQ staged queries, this lane's share:
// lane j of a 16-lane DPP row takes float4 columns j, j+16, ...; the caller reduces across the row.
// HALF: the row comes from the bf16 mirror (nmn_api.hip: `half`) — a 16-byte chunk holds 8 elements, each dword
// element 2i in its low and 2i+1 in its high 16 bits; `ld4` then counts the row's 16-byte chunks (ld / 8) and `qld4`
// the float4s of one staged query (ld / 4).
template <int METRIC, int NQ, int CH, bool FULL, bool NT, bool HALF>
__device__ __forceinline__ void row_partial(const v4f* __restrict__ rowp, bool active, uint32_t j, uint32_t ld4,
                                            uint32_t qld4, const v4f* __restrict__ qs4, float (&acc)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = 0.f;
    for (uint32_t c0 = 0; c0 < ld4; c0 += 16u * CH) {
        v4f x[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t col = c0 + (uint32_t)c * 16u + j;
            const bool ok = (FULL || col < ld4) && active;
            if (ok) x[c] = load4<NT>(rowp + col);
            else x[c] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            uint32_t col = c0 + (uint32_t)c * 16u + j;
            if constexpr (!FULL) col = min(col, ld4 - 1u);  // x is zero there; keep the LDS read in range
            if constexpr (HALF) {
                const uint32_t w0 = __float_as_uint(x[c].x), w1 = __float_as_uint(x[c].y), w2 = __float_as_uint(x[c].z),
                               w3 = __float_as_uint(x[c].w);
                const v4f lo = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xFFFF0000u), __uint_as_float(w1 << 16),
                                __uint_as_float(w1 & 0xFFFF0000u)};
                const v4f hi = {__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xFFFF0000u), __uint_as_float(w3 << 16),
                                __uint_as_float(w3 & 0xFFFF0000u)};
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const v4f qa = qs4[(uint32_t)q * qld4 + 2u * col], qb = qs4[(uint32_t)q * qld4 + 2u * col + 1u];
                    if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
                        const v4f da = lo - qa, db = hi - qb;
                        acc[q] = __builtin_fmaf(da.x, da.x, acc[q]);
                        acc[q] = __builtin_fmaf(da.y, da.y, acc[q]);
                        acc[q] = __builtin_fmaf(da.z, da.z, acc[q]);
                        acc[q] = __builtin_fmaf(da.w, da.w, acc[q]);
                        acc[q] = __builtin_fmaf(db.x, db.x, acc[q]);
                        acc[q] = __builtin_fmaf(db.y, db.y, acc[q]);
                        acc[q] = __builtin_fmaf(db.z, db.z, acc[q]);
                        acc[q] = __builtin_fmaf(db.w, db.w, acc[q]);
                    } else {
                        acc[q] = __builtin_fmaf(lo.x, qa.x, acc[q]);
                        acc[q] = __builtin_fmaf(lo.y, qa.y, acc[q]);
                        acc[q] = __builtin_fmaf(lo.z, qa.z, acc[q]);
                        acc[q] = __builtin_fmaf(lo.w, qa.w, acc[q]);
                        acc[q] = __builtin_fmaf(hi.x, qb.x, acc[q]);
                        acc[q] = __builtin_fmaf(hi.y, qb.y, acc[q]);
                        acc[q] = __builtin_fmaf(hi.z, qb.z, acc[q]);
                        acc[q] = __builtin_fmaf(hi.w, qb.w, acc[q]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const v4f qv = qs4[(uint32_t)q * ld4 + col];
                if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
                    const v4f d = x[c] - qv;
                    acc[q] = __builtin_fmaf(d.x, d.x, acc[q]);
                    acc[q] = __builtin_fmaf(d.y, d.y, acc[q]);
                    acc[q] = __builtin_fmaf(d.z, d.z, acc[q]);
                    acc[q] = __builtin_fmaf(d.w, d.w, acc[q]);
                } else {
                    acc[q] = __builtin_fmaf(x[c].x, qv.x, acc[q]);
                    acc[q] = __builtin_fmaf(x[c].y, qv.y, acc[q]);
                    acc[q] = __builtin_fmaf(x[c].z, qv.z, acc[q]);
                    acc[q] = __builtin_fmaf(x[c].w, qv.w, acc[q]);
                }
            }
        }
    }
}

// Sparse predicate bitmaps (a selective WHERE, an IVF probe): a tile with at most this many participating rows
// is processed in COMPACTED steps — the 4 DPP rows of the wave always read 4 participating corpus rows — instead
// of 16 fixed steps in which most 16-lane groups would idle and the wave would keep a quarter of its loads in flight.
constexpr uint32_t kCompactMaxRows = 40;
// The survivor walk of sparse bitmaps (round 5; nmn_scan_i8.hip has had it since round 3): a wave lists the participating rows of up to
// 64 of its tiles at once (at most kWalkRows: eight full tiles always fit) and reads them four per step straight down the list — every
// step full whatever the tile borders, no per-tile chain of dependent round trips (bitmap -> rows -> store); the scores are parked in
// LDS and the tiles' 256-byte score blocks and maxima written at the end.  Taken when the 64 tiles hold on average at most
// kWalkDense participating rows each (denser tiles are cheaper tile by tile); NMN_NO_WALK=1 (nmn_api.hip) turns it off.
constexpr uint32_t kWalkRows = 512;
#ifndef NMN_F32_WALK_DENSE
#define NMN_F32_WALK_DENSE 20u
#endif
constexpr uint32_t kWalkDense = NMN_F32_WALK_DENSE;
template <int N>
__device__ __forceinline__ float row_share(float v) {  // lane N of the caller's 16-lane DPP row, to all of its lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xF, 0xF, false));
}

// METRIC: nmn_metric.  MASKED: predicate bitmap present.  NQ: queries per pass over the corpus.
// CH: float4 loads per lane per chunk (CH KiB in flight per wave).  FULL: ld4 % (16*CH) == 0, no
// column predicate.  NT: non-temporal corpus loads.
template <int METRIC, bool MASKED, int NQ, int CH, bool FULL, bool NT, bool HALF = false>
__global__ void __launch_bounds__(256) scan_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [NQ][ld]
    const uint32_t ld = p.ld, ld4 = ld >> 2;
    // the streamed matrix: the f32 corpus, or its bf16 mirror (half the bytes; the rounding is covered by the margin)
    const float* const mat = HALF ? p.corpus_half : p.corpus;
    const uint32_t mat_ld = HALF ? ld >> 1 : ld;      // row stride in floats
    const uint32_t chunks = HALF ? ld >> 3 : ld4;     // 16-byte chunks per row
    const uint32_t q0 = blockIdx.y * NQ;
    if (p.retry_state) {  // f32 retry after a bf16 pass: nothing to do unless one of this block's queries overflowed
        bool any = false;
#pragma unroll
        for (int q = 0; q < NQ; q++) any = any || (q0 + q < p.nq && p.retry_state[q0 + q].overflow == 1);
        if (!any) return;
    }
    {
        v4f* qs4 = reinterpret_cast<v4f*>(qs);
        for (uint32_t i = threadIdx.x; i < NQ * ld4; i += 256) {
            uint32_t q = i / ld4, c = i - q * ld4;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (q0 + q < p.nq) v = reinterpret_cast<const v4f*>(p.qpad + (size_t)(q0 + q) * ld)[c];
            qs4[i] = v;
        }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t j = lane & 15u, grp = lane >> 4;
    // the wave's tiles: a contiguous range, or (p.strided: masked sweeps) every W-th tile starting at its own number
    const uint32_t n_waves_all = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    const bool strided = MASKED && p.strided != 0;
    if (wave >= n_waves_all) return;
    auto tile_at = [&](uint32_t jt) -> uint32_t { return strided ? jt * n_waves_all + wave : wave * p.tiles_per_wave + jt; };
    const v4f* qs4 = reinterpret_cast<const v4f*>(qs);

    float qmag[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) qmag[q] = (q0 + q < p.nq) ? p.qinfo[q0 + q].qmag : 0.f;

    uint32_t wmax[NQ];  // maximum key over every tile of this wave (third level of the max hierarchy)
#pragma unroll
    for (int q = 0; q < NQ; q++) wmax[q] = kKeyMasked;

    uint64_t mcache = 0;
    for (uint32_t rel = 0; rel < p.tiles_per_wave; rel++) {
        const uint32_t tile = tile_at(rel);
        if (tile >= p.n_tiles) break;
        const uint64_t r0 = (uint64_t)tile * kTileRows;
        uint64_t mword = ~0ull;
        if constexpr (MASKED) {
            // the wave's bitmap words are fetched 64 tiles at a time (lane L keeps the word of its tile rel + L): a word per tile
            // from memory put one more load latency on every tile's dependent chain (word -> row addresses -> rows)
            if ((rel & 63u) == 0) {
                const uint32_t tl = tile_at(rel + lane);
                const bool tl_ok = rel + lane < p.tiles_per_wave && tl < p.n_tiles;
                mcache = tl_ok ? p.mask[tl] : 0ull;
                if (tl_ok) {
                    const uint64_t left = p.n_rows - (uint64_t)tl * kTileRows;
                    if (left < 64) mcache &= (1ull << left) - 1ull;
                }
                // (one or two queries per pass; the f64 artifact metric keeps its own trust rules in the tile-by-tile epilogue)
                bool walk_here = NQ <= 2 && p.walk != 0 && p.metric != NMN_METRIC_SPARSE_COSINE_F64;
                const uint32_t n_here = min(64u, p.tiles_per_wave - rel);
                const uint32_t cnt = (uint32_t)__builtin_popcountll(mcache);
                if (walk_here) {
                    uint32_t tot = cnt;
#pragma unroll
                    for (uint32_t d = 1; d < 64; d <<= 1) tot += (uint32_t)__shfl_xor((int)tot, d);
                    walk_here = tot <= kWalkDense * n_here;
                }
                if (walk_here) {  // (wave-uniform) ---- the survivor walk: these (up to) 64 tiles at once
                    float* const wbase = qs + (size_t)NQ * ld + 4 * (NQ * 64) + 4 * 64;  // behind the queries, the staging arrays and the rank tables
                    uint16_t* const wtab = reinterpret_cast<uint16_t*>(wbase) + (threadIdx.x >> 6) * kWalkRows;
                    uint32_t* const wsc = reinterpret_cast<uint32_t*>(wbase + 4 * (kWalkRows / 2)) + (threadIdx.x >> 6) * (NQ * kWalkRows);
                    uint32_t incl = cnt;
#pragma unroll
                    for (uint32_t d = 1; d < 64; d <<= 1) {
                        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
                        if (lane >= d) incl += t;
                    }
                    uint32_t base_lane = 0, base_cnt = 0;
                    while (base_lane < n_here) {
                        const uint64_t fit = __ballot(lane >= base_lane && lane < n_here && incl - base_cnt <= kWalkRows);
                        const uint32_t end_lane = base_lane + (uint32_t)__builtin_popcountll(fit);  // (>= base_lane + 8 or n_here)
                        const uint32_t S = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(end_lane - 1)) - base_cnt;
                        const bool mine = lane >= base_lane && lane < end_lane;
                        const uint32_t off = incl - cnt - base_cnt;
                        if (mine) {
                            uint64_t w = mcache;
                            uint32_t o = off;
                            while (w) {
                                wtab[o++] = (uint16_t)((lane << 6) | (uint32_t)__builtin_ctzll(w));
                                w &= w - 1;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t s0 = 0; s0 < S; s0 += 4u) {
                            const uint32_t e = s0 + grp < S ? (uint32_t)wtab[s0 + grp] : 0xFFFFu;
                            const bool active = e != 0xFFFFu;
                            const uint64_t row = active ? (uint64_t)tile_at(rel + (e >> 6)) * kTileRows + (e & 63u) : 0ull;
                            float f = 0.f;  // lane 0 of the row's sixteen fetches its magnitude, ahead of the row itself
                            if constexpr (METRIC == NMN_METRIC_COSINE) {
                                if (active && j == 0) f = p.norms[row];
                            }
                            float acc[NQ];
                            row_partial<METRIC, NQ, CH, FULL, NT, HALF>(reinterpret_cast<const v4f*>(mat + row * (uint64_t)mat_ld), active, j, chunks, ld4, qs4, acc);
                            const float vn = METRIC == NMN_METRIC_COSINE ? row_share<0>(f) : 1.f;
#pragma unroll
                            for (int q = 0; q < NQ; q++) {
                                const float dot = row16_sum(acc[q]);
                                float sc;
                                if constexpr (METRIC == NMN_METRIC_COSINE) {
                                    sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : dot / (qmag[q] * vn);
                                } else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
                                    const float dist = sqrtf(fmaxf(dot, 0.f));
                                    sc = p.metric == kMetricNegL2 ? -dist : 1.0f / (1.0f + dist);
                                } else {
                                    sc = dot;
                                }
                                if (j == 0 && active) wsc[q * kWalkRows + s0 + grp] = f2u(sc);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        // the tiles' score blocks (non-participating rows: the sentinel), tile by tile
                        for (uint32_t tl2 = base_lane; tl2 < end_lane; tl2++) {
                            const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mcache, (int)tl2);
                            const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mcache >> 32), (int)tl2);
                            const uint64_t w = ((uint64_t)whi << 32) | wlo;
                            if (w == 0ull) continue;  // (its maximum says so below; nobody reads the scores of such a tile)
                            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)tl2);
                            const bool set = ((w >> lane) & 1ull) != 0;
                            const uint32_t rank = (uint32_t)__builtin_popcountll(w & ((1ull << lane) - 1ull));
                            const uint64_t row = (uint64_t)tile_at(rel + tl2) * kTileRows + lane;
#pragma unroll
                            for (int q = 0; q < NQ; q++) {
                                const uint32_t bits = set ? wsc[q * kWalkRows + o + rank] : kScoreSentinelBits;
                                if (q0 + q < p.nq) p.scores[score_at(row, q0 + q, p.nql)] = bits;
                            }
                        }
                        // tile maxima: lane L walks its own tile's entries
                        uint32_t m[NQ];
#pragma unroll
                        for (int q = 0; q < NQ; q++) m[q] = kKeyMasked;
                        if (mine) {
                            for (uint32_t i = 0; i < cnt; i++) {
#pragma unroll
                                for (int q = 0; q < NQ; q++) m[q] = max(m[q], score_to_key(u2f(wsc[q * kWalkRows + off + i])));
                            }
                        }
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            if (q0 + q < p.nq) {
                                if (mine && tl_ok) p.tmax[(uint64_t)(q0 + q) * p.tmax_stride + tl] = m[q];
                                wmax[q] = max(wmax[q], wave_max_u32(m[q]));
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        base_lane = end_lane;
                        base_cnt += S;
                    }
                    rel += 63u;  // (the loop's own increment completes the 64)
                    continue;
                }
            }
            const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)mcache, (int)(rel & 63u));
            const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(mcache >> 32), (int)(rel & 63u));
            mword = ((uint64_t)hi << 32) | lo;
        }
        {
            const uint64_t left = p.n_rows - r0;
            if (left < 64) mword &= (1ull << left) - 1ull;
        }
        if constexpr (MASKED) {
            // a tile the bitmap excludes entirely (half of the tiles at selectivity 0.01): its maximum says "nobody takes part",
            // which is all anybody reads of it — selection, crowd list and certificate gate every read of scores[] by the
            // tile maximum, the exact fallback rewrites the scores of the whole shard itself.  No sentinel store, no epilogue.
            if (mword == 0ull) {  // (wave-uniform)
                if (lane < (uint32_t)NQ && q0 + lane < p.nq) p.tmax[(uint64_t)(q0 + lane) * p.tmax_stride + tile] = kKeyMasked;
                continue;
            }
        }
        float mydot[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) mydot[q] = 0.f;

        bool compacted = false;
        if constexpr (MASKED) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(mword);
            if (cnt <= kCompactMaxRows) {  // wave-uniform
                compacted = true;
                float* stg = qs + (size_t)NQ * ld + (threadIdx.x >> 6) * (NQ * 64);  // this wave's [NQ][64] staging
                // rank -> row table of the tile, built once: lane L, if its row takes part, writes its bit position at its rank
                // (round 1 found each step's four rows with four ballots per step)
                uint32_t* tab = reinterpret_cast<uint32_t*>(qs + (size_t)NQ * ld + 4 * (NQ * 64)) + (threadIdx.x >> 6) * 64;
                const bool myset = ((mword >> lane) & 1ull) != 0;
                const uint32_t myrank = (uint32_t)__builtin_popcountll(mword & ((1ull << lane) - 1ull));
                if (myset) tab[myrank] = lane;
                __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave complete in order; this only pins the compiler)
                for (uint32_t s0 = 0; s0 < cnt; s0 += 4u) {
                    const bool active = s0 + grp < cnt;
                    const uint32_t pos = active ? tab[s0 + grp] : 0u;
                    const v4f* rowp = reinterpret_cast<const v4f*>(mat + (r0 + pos) * (uint64_t)mat_ld);
                    float acc[NQ];
                    row_partial<METRIC, NQ, CH, FULL, NT, HALF>(rowp, active, j, chunks, ld4, qs4, acc);
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const float t = row16_sum(acc[q]);
                        if (j == 0 && active) stg[q * 64 + (int)pos] = t;
                    }
                }
                // hand every result to the lane that finishes its row (lane L <- tile row (L&15)*4 + (L>>4))
                const uint32_t bit = j * 4u + grp;
                if ((mword >> bit) & 1ull) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) mydot[q] = stg[q * 64 + (int)bit];
                }
            }
        }
        if (!compacted) {
#pragma unroll 2
            for (uint32_t s = 0; s < 16; s++) {
                if constexpr (MASKED) {
                    if (((mword >> (s * 4)) & 0xFull) == 0) continue;  // wave-uniform: 4 rows all excluded
                }
                const uint32_t rbit = s * 4 + grp;
                const bool active = MASKED ? ((mword >> rbit) & 1ull) != 0 : true;
                const v4f* rowp = reinterpret_cast<const v4f*>(mat + (r0 + rbit) * (uint64_t)mat_ld);
                float acc[NQ];
                row_partial<METRIC, NQ, CH, FULL, NT, HALF>(rowp, active, j, chunks, ld4, qs4, acc);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float t = row16_sum(acc[q]);
                    if (j == s) mydot[q] = t;
                }
            }
        }

        // lane L finishes tile row (L&15)*4 + (L>>4)
        const uint32_t mybit = j * 4u + grp;
        const bool valid = ((mword >> mybit) & 1ull) != 0;
        const uint64_t myrow = r0 + mybit;
        float vn = 1.f;
        if constexpr (METRIC == NMN_METRIC_COSINE) vn = valid ? p.norms[myrow] : 1.f;  // f32 |v|: fine as an approximation of the f64 one too
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (q0 + q >= p.nq) break;
            float sc;
            if constexpr (METRIC == NMN_METRIC_COSINE) {
                sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : mydot[q] / (qmag[q] * vn);
                // f64 artifact similarity: the reference computes in f64, so magnitudes whose SQUARES leave the f32
                // range (overflow to inf, underflow to 0), a NaN anywhere, or a zero magnitude say nothing about the
                // f64 result: such rows are forced into the candidates and the exact f64 rescore decides.
                if (p.metric == NMN_METRIC_SPARSE_COSINE_F64) {
                    const bool trust = vn >= 1e-15f && vn <= 1e18f && qmag[q] >= 1e-15f && qmag[q] <= 1e18f &&
                                       __builtin_fabsf(mydot[q]) <= 3.0e38f;
                    if (!trust) sc = __builtin_inff();
                }
            } else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
                const float dist = sqrtf(fmaxf(mydot[q], 0.f));
                sc = p.metric == kMetricNegL2 ? -dist : 1.0f / (1.0f + dist);  // IVF probe ranks by distance
            } else {
                sc = mydot[q];
            }
            const uint32_t key = valid ? score_to_key(sc) : kKeyMasked;
            p.scores[score_at(myrow, q0 + q, p.nql)] = valid ? f2u(sc) : kScoreSentinelBits;
            const uint32_t m = wave_max_u32(key);
            if (lane == 0) p.tmax[(uint64_t)(q0 + q) * p.tmax_stride + tile] = m;
            wmax[q] = max(wmax[q], m);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
            if (q0 + q < p.nq) p.wmax[(size_t)(q0 + q) * p.wmax_stride + wave] = wmax[q];
    }
}

template <int METRIC, bool MASKED, int NQ, int CH, bool FULL, bool NT, bool HALF = false>
static hipError_t launch_one(const ScanParams& p, hipStream_t s) {
    const uint32_t waves = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    dim3 grid((waves + 3) / 4, (p.nq + NQ - 1) / NQ);
    // queries, plus (masked kernels) one [NQ][64] staging array and one 64-entry rank -> row table per wave for the compacted tiles
    const size_t lds = (size_t)NQ * p.ld * sizeof(float) + (MASKED ? 4 * NQ * 64 * sizeof(float) + 4 * 64 * sizeof(uint32_t) +  // + rank tables
                                                                       (NQ <= 2 ? 4 * kWalkRows * sizeof(uint16_t) + 4 * NQ * kWalkRows * sizeof(uint32_t) : 0) : 0);  // + the walk's list and parked scores
    auto kern = scan_kernel<METRIC, MASKED, NQ, CH, FULL, NT, HALF>;
    // Masked strided sweeps hold TWO workgroups per CU, not the four their registers allow (round 6): the LDS request is raised to 57 KiB (more than a third of a CU's),
    // which is how a launch says so.  With four, all 1024 workgroups are resident from the start — 48 MB of row loads in flight, three
    // times what the memory system needs — and nothing is left of the CUs for the OTHER stream's selection / rescore tail; with two, the
    // second half of the workgroups goes wherever a CU frees up first.  10M x 1536 Euclidean TOP-1000 over the f32 rows, bench.py's
    // pipelined loop, three rounds in one call: selectivity 0.5 206.1 -> 208.5 q/s, 0.1 957 -> 974, 0.02 3 592 -> 3 856 (+7 %: the tail
    // no longer waits for the sweep to end), 0.25 +1.9 %, 0.004 -2.3 % (3 us); three per CU loses at 0.1, one per CU at 0.02
    // (profiles/r06ap_*).  The f32 rows only: the bf16 mirror's shorter sweeps gain 4 % at 0.5 and lose 3-4 % at 0.1 / 0.02, the 8-bit
    // mirror's +2 % / -3 % (nmn_scan_i8.hip keeps the knob, off).  NMN_SCAN_WGS_PER_CU=0: the A/B.
    static const int wgs_per_cu = [] { const char* e = getenv("NMN_SCAN_WGS_PER_CU"); return e ? atoi(e) : 2; }();
    size_t lds_total = lds;
    if (MASKED && !HALF && p.strided && wgs_per_cu > 0 && grid.x > 256u * (unsigned)wgs_per_cu)
        lds_total = std::max<size_t>(lds, ((size_t)160 * 1024 / (size_t)(wgs_per_cu + 1) + 4096) & ~(size_t)1023);  // more than a (n+1)-th of the CU's LDS: n fit
    if (lds_total > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_total, s, p);
    return hipGetLastError();
}

template <int METRIC, bool MASKED, int NQ>
static hipError_t launch_layout(const ScanParams& p, hipStream_t s, bool nt) {
    const uint32_t ld4 = p.ld >> 2;
    if (ld4 % (16 * 12) == 0) {
        if (nt) return launch_one<METRIC, MASKED, NQ, 12, true, true>(p, s);
        return launch_one<METRIC, MASKED, NQ, 12, true, false>(p, s);
    }
    if (ld4 % (16 * 8) == 0) return launch_one<METRIC, MASKED, NQ, 8, true, false>(p, s);
    if (ld4 % (16 * 2) == 0) return launch_one<METRIC, MASKED, NQ, 2, true, false>(p, s);
    return launch_one<METRIC, MASKED, NQ, 4, false, false>(p, s);
}

// bf16 mirror: ld / 8 chunks per row (768 -> 96 = 16 * 6, 1536 -> 192 = 16 * 12)
template <int METRIC, bool MASKED, int NQ>
static hipError_t launch_layout_half(const ScanParams& p, hipStream_t s, bool nt) {
    {
        const uint32_t lc = p.ld >> 3;
        if (lc % (16 * 12) == 0) {
            // (masked sweeps too: the rows a bitmap keeps are read once like any others — scattered 3 KiB rows stream at
            //  6.56 TB/s non-temporal against 6.19 with the default policy, tools/micro/read_bw.hip "listed rows")
            if (nt) return launch_one<METRIC, MASKED, NQ, 12, true, true, true>(p, s);
            return launch_one<METRIC, MASKED, NQ, 12, true, false, true>(p, s);
        }
        if (lc % (16 * 6) == 0) {
            if (nt) return launch_one<METRIC, MASKED, NQ, 6, true, true, true>(p, s);
            return launch_one<METRIC, MASKED, NQ, 6, true, false, true>(p, s);
        }
        if (lc % (16 * 2) == 0) return launch_one<METRIC, MASKED, NQ, 2, true, false, true>(p, s);
        return launch_one<METRIC, MASKED, NQ, 4, false, false, true>(p, s);
    }
}

template <int METRIC, bool MASKED>
static hipError_t launch_nq(const ScanParams& p, hipStream_t s, bool nt) {
    if (p.corpus_half) {
        if (p.nq >= 3) return launch_layout_half<METRIC, MASKED, 4>(p, s, nt);
        if (p.nq == 2) return launch_layout_half<METRIC, MASKED, 2>(p, s, nt);
        return launch_layout_half<METRIC, MASKED, 1>(p, s, nt);
    }
    if (p.nq >= 3) return launch_layout<METRIC, MASKED, 4>(p, s, nt);
    if (p.nq == 2) return launch_layout<METRIC, MASKED, 2>(p, s, nt);
    return launch_layout<METRIC, MASKED, 1>(p, s, nt);
}

template <int METRIC>
static hipError_t launch_mask(const ScanParams& p, hipStream_t s, bool nt) {
    return p.mask ? launch_nq<METRIC, true>(p, s, nt) : launch_nq<METRIC, false>(p, s, nt);
}

static bool scan_nt_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("NMN_SCAN_NT");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v != 0;
}

// ---- the bf16 mirror ----------------------------------------------------------------------------------
// half[row][c] = bf16(corpus[row][c]), round to nearest even, same row order, half the row stride.  The approximate
// sweep of 1-4 queries reads THIS matrix: the result stays exact because the rounding (relative 2^-8 per element:
// at most 2^-8 |q||v| on a dot product, at most 2^-8 |v| on a Euclidean distance) is added to the candidate margin
// (qprep_kernel) and every candidate is re-scored from the f32 corpus.
// row_err2[r - row0] = |v - bf16(v)|^2 of row r: the candidate margins use the
// MEASURED rounding error of the mirror — max_r |e_r| and max_r |e_r| / |v_r| — which is rigorous like the worst case
// 2^-8 |v| but about 2.5x smaller (the rounding error of a bf16 is uniform in +-half an ulp, not always the maximum).
// `lpr` lanes (a power of two <= 64) share one row and fold their partial sums with shuffles: one plain store per
// row (atomics per 8 elements made this kernel 8x slower than its traffic).
__global__ void __launch_bounds__(256) half_rows_kernel(const float* __restrict__ corpus, float* __restrict__ half,
                                                        uint32_t ld, uint64_t row0, uint64_t n, float* __restrict__ row_err2,
                                                        uint32_t lpr) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const uint32_t per_row = ld >> 3;  // 8-element groups per row
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & (lpr - 1u), slot = lane / lpr, rows_per_wave = 64u / lpr;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t rb = wave * rows_per_wave; rb < n; rb += n_waves * rows_per_wave) {
        const uint64_t ri = rb + slot;
        const bool live = ri < n;
        const uint64_t r = row0 + (live ? ri : 0);
        float err2 = 0.f;
        for (uint32_t g = sub; live && g < per_row; g += lpr) {
            const v4f a = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u);
            const v4f b = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u + 4u);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t pk[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f2 pr = {x[2 * t], x[2 * t + 1]};
                pk[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf2));
                const float e0 = x[2 * t] - __uint_as_float(pk[t] << 16), e1 = x[2 * t + 1] - __uint_as_float(pk[t] & 0xFFFF0000u);
                err2 = __builtin_fmaf(e0, e0, err2);
                err2 = __builtin_fmaf(e1, e1, err2);
            }
            const v4f out = {__uint_as_float(pk[0]), __uint_as_float(pk[1]), __uint_as_float(pk[2]), __uint_as_float(pk[3])};
            *reinterpret_cast<v4f*>(half + r * (ld >> 1) + g * 4u) = out;
        }
        for (uint32_t off = lpr >> 1; off > 0; off >>= 1) err2 += __shfl_xor(err2, (int)off);
        if (live && sub == 0) row_err2[ri] = err2;
    }
}

// err_bits[0] = max_r |e_r|, err_bits[1] = max_r |e_r| / |v_r| (both with slack for the summation order of the atomics)
__global__ void __launch_bounds__(256) half_err_kernel(const float* __restrict__ row_err2, const float* __restrict__ norms,
                                                       uint64_t row0, uint64_t n, uint32_t* __restrict__ err_bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float e = __builtin_sqrtf(row_err2[i]) * 1.0005f;
        const float vn = norms[row0 + i];
        if (e == e && e > 0.f) {  // e >= 0: bit order == value order
            atomicMax(err_bits, __float_as_uint(e));
            if (vn > 0.f) {
                const float rel = e / vn * 1.0005f;
                if (rel == rel) atomicMax(err_bits + 1, __float_as_uint(rel));
            }
        }
    }
}

hipError_t launch_half_rows(const float* corpus, float* half, uint32_t ld, uint64_t row0, uint64_t n, const float* norms,
                            float* row_err2_scratch, uint32_t* err_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t lpr = 1;
    while (lpr < 64u && lpr < (ld >> 3)) lpr <<= 1;
    hipLaunchKernelGGL(half_rows_kernel, dim3(256 * 16), dim3(256), 0, s, corpus, half, ld, row0, n, row_err2_scratch, lpr);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(half_err_kernel, dim3(blocks), dim3(256), 0, s, row_err2_scratch, norms, row0, n, err_bits);
    return hipGetLastError();
}

// every metric the VALU sweep serves, as long as the rows are made of whole 16-byte bf16 chunks
bool scan_half_supported(uint32_t ld, int metric) {
    (void)metric;
    return ld % 8u == 0;
}

// ---- read-ceiling probe -----------------------------------------------------------------------------
// The scan's access pattern with the arithmetic removed: every wave streams whole rows (16 lanes per row, 4 rows
// per step, CH non-temporal 16-byte loads in flight per lane) and folds them into one register.  Its bandwidth
// is what a read-only sweep over THIS shard can reach on this device; bench.py quotes the scan against it as
// well as against the nominal HBM peak.
__global__ void __launch_bounds__(256) read_probe_kernel(const float* __restrict__ corpus, uint64_t n_rows, uint32_t ld,
                                                         float* __restrict__ sink) {
    constexpr int CH = 12;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, grp = lane >> 4;
    const uint32_t ld4 = ld >> 2;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t per = ((n_rows + n_waves - 1) / n_waves + 3) & ~3ull;
    const uint64_t r0 = wave * per, r1 = min(r0 + per, n_rows);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t r = r0 + grp; r < r1; r += 4) {
        const v4f* rowp = reinterpret_cast<const v4f*>(corpus + r * (uint64_t)ld);
        for (uint32_t c0 = 0; c0 < ld4; c0 += 16u * CH) {
            v4f x[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t col = c0 + (uint32_t)c * 16u + j;
                x[c] = col < ld4 ? load4<true>(rowp + col) : (v4f){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < CH; c++) acc += x[c];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345678e-33f) sink[0] = acc.x;  // practically never: keeps the loads alive
}

hipError_t launch_read_probe(const float* corpus, uint64_t n_rows, uint32_t ld, float* sink, hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(read_probe_kernel, dim3(kMaxScanWaves / 4), dim3(256), 0, s, corpus, n_rows, ld, sink);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanParams& p, hipStream_t s) {
    const bool nt = scan_nt_enabled();
    switch (p.metric) {
        case NMN_METRIC_COSINE:
        case NMN_METRIC_SPARSE_COSINE_F64: return launch_mask<NMN_METRIC_COSINE>(p, s, nt);  // same sweep
        case NMN_METRIC_EUCLIDEAN:
        case kMetricNegL2: return launch_mask<NMN_METRIC_EUCLIDEAN>(p, s, nt);  // same sweep, epilogue picks -d over 1/(1+d)
        default: return launch_mask<NMN_METRIC_DOT_PRODUCT>(p, s, nt);
    }
}

}  // namespace nmn
