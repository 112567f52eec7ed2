// nmn_scan_i8.hip — the single-query sweep over the shard's 8-BIT mirror: one byte per corpus element instead of the two of
// the bf16 mirror (nmn_scan.hip) or the four of the f32 corpus.  The sweep is HBM-bound and runs at ~0.95 of what a plain
// read of the same buffer achieves, so the only lever left on queries/s is the number of bytes a query has to move.
//
// Still the reference's answer, bit for bit: the sweep only has to be APPROXIMATELY right.  What it replaces is the per-key
// loop of vector_engine/src/lib.rs:2115-2228; the scores it produces select candidates, every candidate is re-scored from
// the f32 corpus in the reference's operation order (nmn_exact.hip; tensor_store/src/hnsw.rs:168-229, lib.rs:2231-2266),
// and the candidate margin carries the MEASURED error of this representation:
//   row r is stored as  v_r = s_r * c_r + e_r,  c_r in [-127, 127]^d (int8),  s_r = max_i |v_r[i]| / 127  (f32, per row),
//   ||e_r|| and ||e_r|| / ||v_r|| are measured while the mirror is written and their maxima kept (q8_err_bits);
//   the query is split  q = s_q * (h + l / 256) + e_q  with h, l int8 vectors (||e_q|| / ||q|| ~ 3e-5, measured per query),
//   so the sweep's dot product  s_q s_r (h.c + (l.c) / 256)  is EXACT integer arithmetic (v_dot4_i32_i8: four multiply-adds
//   per instruction, int32 accumulators) and differs from q.v by at most ||q|| ||v_r|| (rho_q (1 + rho_v) + rho_v).
// For N(0,1)-like rows of 768 elements rho_v is about 0.009 (bf16 mirror: 0.0015): ~1 000 candidates instead of ~270 at
// 10M rows, k = 100 — still one rescore launch.  A corpus whose rows are dominated by a few large elements gets a large
// measured rho_v, overflows the candidate lists, and the shard goes back to the bf16 mirror by itself (nmn_api.hip).
// Euclidean scores: the sweep computes the distance between the STORED representations exactly,
// ||q~ - v~||^2 = ||q~||^2 + ||v~||^2 - 2 q~.v~ (||v~||^2 = s_r^2 c_r.c_r kept per row, ||q~||^2 per query), so by the triangle
// inequality the DISTANCE is off by at most ||e_q|| + ||e_r|| — an absolute error in distance space (QInfo.pad), which stays
// small for near neighbours where a bound on q.e_r (2 ||q|| ||e_r|| on the squared distance) would swamp them; the f32
// roundings of the three-term expression are bounded in squared-distance space on top of it (QInfo.pad_sq).
//
// Mapping: the one of scan_kernel — a wave owns whole 64-row tiles, 16 steps x 4 rows, each 16-lane DPP row reads ONE corpus
// row with 16-byte non-temporal loads (lane j takes chunks j, j + 16, ...; a chunk is 16 elements), row_ror reductions, lane
// L finishes tile row (L & 15) * 4 + (L >> 4).  Rows of up to 1536 elements keep the query's h / l planes in REGISTERS
// (8 VGPRs per chunk and lane: 24 at 768, 48 at 1536): no LDS traffic at all in the loop.  Algorithmic bytes per row:
// dim (+ 4 B scale, 4 B magnitude read, 4 B score written); per launch rows * dim = 7.68 GB at 10M x 768.
#include <algorithm>
#include <type_traits>

#include "nmn_internal.h"

namespace nmn {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float row16_sum(float v) {  // all-reduce over a 16-lane DPP row (row_ror 8, 4, 2, 1)
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
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// 16 int8 of a row against 16 int8 of the query's h plane and of its l plane: 8 x v_dot4_i32_i8
__device__ __forceinline__ void dot16(const v4i x, const v4i qh, const v4i ql, int& hi, int& lo) {
    hi = __builtin_amdgcn_sdot4(x.x, qh.x, hi, false);
    lo = __builtin_amdgcn_sdot4(x.x, ql.x, lo, false);
    hi = __builtin_amdgcn_sdot4(x.y, qh.y, hi, false);
    lo = __builtin_amdgcn_sdot4(x.y, ql.y, lo, false);
    hi = __builtin_amdgcn_sdot4(x.z, qh.z, hi, false);
    lo = __builtin_amdgcn_sdot4(x.z, ql.z, lo, false);
    hi = __builtin_amdgcn_sdot4(x.w, qh.w, hi, false);
    lo = __builtin_amdgcn_sdot4(x.w, ql.w, lo, false);
}

// this lane's share of ONE row against NQ queries: h.c and l.c folded into one float per query, h.c + (l.c) / 256 (both
// partial sums are far below 2^24: exact conversions, one rounding of 2^-24 relative in the addition)
// QREG: the query planes of this lane's chunks live in qh / ql; else they are read from the LDS image qs4
// (query q: h plane at [q * 2 * chunks, +chunks), l plane behind it)
template <int NQ, int CH, bool SINGLE, bool QREG>
__device__ __forceinline__ void row_partial_i8(const v4i* __restrict__ rowp, bool active, uint32_t j, uint32_t chunks,
                                               const v4i* __restrict__ qs4, const v4i (&qh)[QREG ? CH : 1],
                                               const v4i (&ql)[QREG ? CH : 1], float (&acc)[NQ]) {
    int hi[NQ], lo[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) hi[q] = lo[q] = 0;
    if constexpr (SINGLE) {
        v4i x[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (active) x[c] = __builtin_nontemporal_load(rowp + (uint32_t)c * 16u + j);
            else x[c] = (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if constexpr (QREG) {
                    dot16(x[c], qh[c], ql[c], hi[q], lo[q]);
                } else {
                    const uint32_t col = (uint32_t)c * 16u + j;
                    dot16(x[c], qs4[(uint32_t)q * 2u * chunks + col], qs4[(uint32_t)q * 2u * chunks + chunks + col], hi[q], lo[q]);
                }
            }
        }
    } else {
        // (rows of an odd number of 128-element halves — 384, 640, 896 ... — end in a chunk group only lanes 0-7 have a part of)
        for (uint32_t c0 = 0; c0 < chunks; c0 += 16u * CH) {
            v4i x[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (active && c0 + (uint32_t)c * 16u + j < chunks) x[c] = __builtin_nontemporal_load(rowp + c0 + (uint32_t)c * 16u + j);
                else x[c] = (v4i){0, 0, 0, 0};
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t col = min(c0 + (uint32_t)c * 16u + j, chunks - 1u);  // (beyond the row: x is zero, any query chunk will do)
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    dot16(x[c], qs4[(uint32_t)q * 2u * chunks + col], qs4[(uint32_t)q * 2u * chunks + chunks + col], hi[q], lo[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = (float)hi[q] + (float)lo[q] * 0.00390625f;
}

// the two halves of row_partial_i8 for rows of exactly 16 * CH chunks, so that the loads of the NEXT steps can be issued before
// the products of the current ones (the pipelined loop of scan_i8_kernel)
template <int CH>
__device__ __forceinline__ void load_row(const v4i* __restrict__ rowp, uint32_t j, v4i (&x)[CH]) {
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = __builtin_nontemporal_load(rowp + (uint32_t)c * 16u + j);
}
template <int NQ, int CH, bool QREG>
__device__ __forceinline__ void dot_row(const v4i (&x)[CH], uint32_t j, uint32_t chunks, const v4i* __restrict__ qs4,
                                        const v4i (&qh)[QREG ? CH : 1], const v4i (&ql)[QREG ? CH : 1], float (&acc)[NQ]) {
    int hi[NQ], lo[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) hi[q] = lo[q] = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if constexpr (QREG) {
                dot16(x[c], qh[c], ql[c], hi[q], lo[q]);
            } else {
                const uint32_t col = (uint32_t)c * 16u + j;
                dot16(x[c], qs4[(uint32_t)q * 2u * chunks + col], qs4[(uint32_t)q * 2u * chunks + chunks + col], hi[q], lo[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = (float)hi[q] + (float)lo[q] * 0.00390625f;
}

// Euclidean score from |q~|^2, |v~|^2 and q~.v~
__device__ __forceinline__ float l2_from_dot(float qq, float vv, float dot, bool neg) {
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, dot, qq + vv), 0.0f);
    const float d = __builtin_amdgcn_sqrtf(d2);
    return neg ? -d : __builtin_amdgcn_rcpf(1.0f + d);
}

constexpr uint32_t kCompactMaxRows = 40;  // tiles with at most this many participating rows run compacted steps (nmn_scan.hip)
// The survivor walk of masked sweeps: a wave lists the participating rows of up to 64 of its tiles at once (at most kWalkRows,
// so 8 full tiles always fit) and reads them four per step, every step full and independent of the tile borders.
#ifndef NMN_I8_WALK_ROWS
#define NMN_I8_WALK_ROWS 512
#endif
constexpr uint32_t kWalkRows = NMN_I8_WALK_ROWS;
#ifndef NMN_I8_MASKED_QREG_CH   // masked sweeps keep the query planes in registers up to this many chunk groups (else they are read from LDS)
#define NMN_I8_MASKED_QREG_CH 6
#endif
#ifndef NMN_I8_NARROW           // 1: rows of 128 elements take the eight-lanes-per-row steps (0: the 16-lane mapping, for the A/B)
#define NMN_I8_NARROW 1
#endif
#ifndef NMN_I8_QREG_CH          // ... and unmasked sweeps (measurement knob: 0 = the planes always come from LDS)
#define NMN_I8_QREG_CH 6
#endif
#ifndef NMN_I8_WALK_DENSE
#define NMN_I8_WALK_DENSE 20u
#endif
#ifndef NMN_I8_WALK_PIPE_CH
#define NMN_I8_WALK_PIPE_CH 3
#endif
template <int N>
__device__ __forceinline__ float row_share(float v) {  // lane N of the caller's 16-lane DPP row, to all of its lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + N, 0xF, 0xF, false));
}

// METRIC: cosine / Euclidean (also the IVF list scan's -d) / dot.  MASKED: predicate bitmap.  NQ: 1 or 2 queries per sweep.
// CH: 16-byte loads per lane and row step; SINGLE: the row is exactly 16 * CH chunks (no column loop).
template <int METRIC, bool MASKED, int NQ, int CH, bool SINGLE>
__global__ void __launch_bounds__(256) scan_i8_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) int qsi[];  // [NQ][2][chunks] x 16 B: h plane, l plane of every query
    constexpr bool QREG = SINGLE && NQ == 1 && (MASKED ? CH <= NMN_I8_MASKED_QREG_CH : CH <= NMN_I8_QREG_CH);
    const uint32_t ld = p.ld, chunks = ld >> 4;
    const uint32_t q0 = blockIdx.y * NQ;
    {
        v4i* qs4w = reinterpret_cast<v4i*>(qsi);
        const uint32_t per_q = 2u * chunks;
        for (uint32_t i = threadIdx.x; i < NQ * per_q; i += 256) {
            const uint32_t q = i / per_q, c = i - q * per_q;
            v4i v = {0, 0, 0, 0};
            if (q0 + q < p.nq) v = reinterpret_cast<const v4i*>(p.qi8 + (size_t)(q0 + q) * (ld >> 1))[c];
            qs4w[i] = v;
        }
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t j = lane & 15u, grp = lane >> 4;
    // the wave's tiles: a contiguous range [t0, t1), or (p.strided: masked sweeps) every W-th tile starting at its own number
    const uint32_t n_waves_all = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    const bool strided = MASKED && p.strided != 0;
    if (wave >= n_waves_all) return;
    const uint32_t t0 = wave * p.tiles_per_wave;
    const uint32_t t1 = min(t0 + p.tiles_per_wave, p.n_tiles);
    auto tile_at = [&](uint32_t jt) -> uint32_t { return strided ? jt * n_waves_all + wave : t0 + jt; };
    const v4i* qs4 = reinterpret_cast<const v4i*>(qsi);
    const int8_t* const mat = p.corpus_i8;

    v4i qh[QREG ? CH : 1], ql[QREG ? CH : 1];
    if constexpr (QREG) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            qh[c] = qs4[(uint32_t)c * 16u + j];
            ql[c] = qs4[chunks + (uint32_t)c * 16u + j];
        }
    } else {
        qh[0] = ql[0] = (v4i){0, 0, 0, 0};
    }

    float qmag[NQ], qsc[NQ], qq8[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        qmag[q] = (q0 + q < p.nq) ? p.qinfo[q0 + q].qmag : 0.f;
        qsc[q] = (q0 + q < p.nq) ? p.qinfo[q0 + q].qscale : 0.f;
        qq8[q] = (q0 + q < p.nq) ? p.qinfo[q0 + q].qq8 : 0.f;
    }
    const bool neg = p.metric == kMetricNegL2;
    bool any_est_b = false;  // some query of this block measures |q|^2 + |v|^2 - 2 q~.v~ (qprep_kernel: "which Euclidean estimator")
#pragma unroll
    for (int q = 0; q < NQ; q++) any_est_b = any_est_b || qq8[q] < 0.f;

    uint32_t wmax[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) wmax[q] = kKeyMasked;

#ifndef NMN_I8_NO_PIPE
    // Unmasked sweeps over rows of one chunk group: the loads run ONE BATCH AHEAD of the products.  A batch is kB row steps
    // (4 rows each, CH 16-byte loads per lane and step); while batch b is multiplied the loads of batch b + 1 — the next
    // tile's first batch at the end of a tile — are already in flight, so the wave never drains its queue between batches
    // (the plain loop below ends every batch at vmcnt(0): 0.79 of the HBM peak at 10M x 768 against 0.8x for this form).
    if constexpr (!MASKED && SINGLE) {
#ifdef NMN_I8_KB  // measurement builds (tools/build_variant.sh): steps per batch for every row length
        constexpr int kB = NMN_I8_KB;
#else
        constexpr int kB = CH <= 2 ? 4 : CH <= 4 ? 2 : 1;   // steps per batch: 4-6 loads per lane in each of the two buffers
#endif
        constexpr int kNB = 16 / kB;                        // batches per tile (even)
        v4i xa[kB][CH], xb[kB][CH];
        auto issue = [&](v4i (&x)[kB][CH], uint32_t tile_, int b) __attribute__((always_inline)) {
            const uint64_t r0_ = (uint64_t)tile_ * kTileRows;
#pragma unroll
            for (int s = 0; s < kB; s++) {
                // (rows past the end of the shard: the last tile's padding rows exist in the mirror — cap_pad rows, zeroed)
                const v4i* rowp = reinterpret_cast<const v4i*>(mat + (r0_ + (uint32_t)(b * kB + s) * 4u + grp) * (uint64_t)ld);
                load_row<CH>(rowp, j, x[s]);
            }
        };
        float mydot[NQ];
        auto multiply = [&](const v4i (&x)[kB][CH], int b) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < kB; s++) {
                float acc[NQ];
                dot_row<NQ, CH, QREG>(x[s], j, chunks, qs4, qh, ql, acc);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float t = row16_sum(acc[q]);
                    if (j == (uint32_t)(b * kB + s)) mydot[q] = t;
                }
            }
        };
        issue(xa, t0, 0);
        const uint32_t mybit = j * 4u + grp;
        for (uint32_t tile = t0; tile < t1; tile++) {
            // this lane's row of the tile: its scale (and magnitude) are requested BEFORE the tile's rows — the queue retires in
            // order, so they are there long before the epilogue asks, and the epilogue's wait does not drain the rows in flight
            const uint64_t r0 = (uint64_t)tile * kTileRows;
            const uint64_t myrow = r0 + mybit;
            const bool valid = myrow < p.n_rows;
            const uint64_t srow = valid ? myrow : 0;  // (padding rows of the last tile: any valid address)
            const float sr_raw = p.i8_scale[srow];
            float vn_raw = 1.f;
            if constexpr (METRIC == NMN_METRIC_COSINE) vn_raw = p.norms[srow];
            if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vn_raw = p.i8_vv[srow];
            float vb = 0.f;  // (Euclidean, estimator B of some query: the row's exact magnitude)
            if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
                if (any_est_b) vb = p.norms[srow];  // (wave-uniform)
            }
            (void)vb;
#pragma unroll
            for (int q = 0; q < NQ; q++) mydot[q] = 0.f;
            // (the tile after the last re-reads the last tile's first batch: issued unconditionally, so that the compiler can
            //  count the queue — a branch around the prefetch made it end every tile at vmcnt(0))
            const uint32_t next_tile = min(tile + 1u, t1 - 1u);
#pragma unroll
            for (int b = 0; b < kNB; b += 2) {
                // (the fences keep a batch's loads together and ahead of the other buffer's products: left alone the scheduler
                //  strings them out one load per eight products with four in flight)
                issue(xb, tile, b + 1);
                __builtin_amdgcn_sched_barrier(0);
                multiply(xa, b);
                __builtin_amdgcn_sched_barrier(0);
                if (b + 2 < kNB) issue(xa, tile, b + 2);
                else issue(xa, next_tile, 0);
                __builtin_amdgcn_sched_barrier(0);
                multiply(xb, b + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float sr = valid ? sr_raw : 0.f;
            float vn = 1.f;
            if constexpr (METRIC == NMN_METRIC_COSINE) vn = valid ? vn_raw : 1.f;
            if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vn = valid ? vn_raw : 0.f;
            // (the arithmetic runs for every query slot, only the stores are guarded: uses of sr / vn inside a conditional block
            //  let the compiler sink their loads down here, behind all the rows of the tile)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const bool q_ok = q0 + q < p.nq;
                const float dot = mydot[q] * (qsc[q] * sr);
                float sc;
                if constexpr (METRIC == NMN_METRIC_COSINE) sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : dot / (qmag[q] * vn);
                else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) sc = qq8[q] < 0.f ? l2_from_dot(qmag[q] * qmag[q], vb * vb, dot, neg) : l2_from_dot(qq8[q], vn, dot, neg);
                else sc = dot;
                const uint32_t key = valid ? score_to_key(sc) : kKeyMasked;
                const uint32_t m = wave_max_u32(key);
                if (q_ok) {
                    p.scores[score_at(myrow, q0 + q, p.nql)] = valid ? f2u(sc) : kScoreSentinelBits;
                    if (lane == 0) p.tmax[(uint64_t)(q0 + q) * p.tmax_stride + tile] = m;
                    wmax[q] = max(wmax[q], m);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (q0 + q < p.nq) p.wmax[(size_t)(q0 + q) * p.wmax_stride + wave] = wmax[q];
        }
        return;
    }
#endif
    uint64_t mcache = 0;
    const bool walk = MASKED && p.walk != 0;
    // rows of exactly 128 / 384 elements: eight lanes per row (under a bitmap: the tile-by-tile steps only — chunks of tiles dense
    // enough to skip the survivor walk; the walk itself keeps the 16-lane mapping)
    // (384 under a bitmap keeps the 16-lane steps: its three loads per lane cost the kernel a wave per SIMD, which the walk
    //  needs more — measured 0.57 -> 0.36 of peak at selectivity 0.25 against 0.66 -> 0.72 at 0.9)
    const bool narrow = !SINGLE && ((CH == 1 && chunks == 8u) || (!MASKED && CH == 2 && chunks == 24u)) && NMN_I8_NARROW;
    for (uint32_t rel = 0; rel < p.tiles_per_wave; rel++) {
        const uint32_t tile = tile_at(rel);
        if (tile >= p.n_tiles) break;
        const uint64_t r0 = (uint64_t)tile * kTileRows;
        uint64_t mword = ~0ull;
        if constexpr (MASKED) {  // the wave's bitmap words, 64 tiles at a time (as scan_kernel)
            if ((rel & 63u) == 0) {
                const uint32_t tl = tile_at(rel + lane);
                const bool tl_ok = rel + lane < p.tiles_per_wave && tl < p.n_tiles;
                mcache = tl_ok ? p.mask[tl] : 0ull;  // (requesting the first 64 words before the query is staged: measured, slower at 1536)
                if (tl_ok) {
                    const uint64_t left = p.n_rows - (uint64_t)tl * kTileRows;
                    if (left < 64) mcache &= (1ull << left) - 1ull;
                }
                bool walk_here = walk;
                if (walk && CH <= 3) {
                    // short rows: a step is few loads under a fixed cost per row (list entry, factors, score); from ~20 participating
                    // rows per tile on the tile-by-tile steps are cheaper (measured at 768: selectivity 0.5 0.72 vs 0.75 of peak)
                    uint32_t tot = (uint32_t)__builtin_popcountll(mcache);
#pragma unroll
                    for (uint32_t d = 1; d < 64; d <<= 1) tot += (uint32_t)__shfl_xor((int)tot, d);
                    walk_here = tot <= NMN_I8_WALK_DENSE * min(64u, p.tiles_per_wave - rel);
                }
                if (walk_here) {
                    // ---- the survivor walk: these (up to) 64 tiles at once --------------------------------------------------
                    // A sparse tile alone is a chain of dependent round trips (bitmap -> rows -> store) that keeps one row step
                    // of a wave in flight; at selectivity 0.01 the sweep spent 40 of its 66 us waiting on ~18 such chains per
                    // wave.  Here lane L owns tile rel + L: a prefix sum of the popcounts places every participating row of
                    // the sub-range in ONE list (tile << 6 | bit), the rows are read four per step straight down the list
                    // (loads of step s + 1 in flight under the products of step s), each row's score is parked in LDS, and
                    // the tiles' 256-byte score blocks, maxima and the wave maximum are written at the end.
                    uint16_t* wtab = reinterpret_cast<uint16_t*>(qsi + (size_t)NQ * (ld >> 1) + 4 * (NQ * 64) + 4 * 64) + (threadIdx.x >> 6) * kWalkRows;
                    uint32_t* wsc = reinterpret_cast<uint32_t*>(qsi + (size_t)NQ * (ld >> 1) + 4 * (NQ * 64) + 4 * 64 + 4 * (kWalkRows / 2)) +
                                    (threadIdx.x >> 6) * (NQ * kWalkRows);
                    const uint32_t n_here = min(64u, p.tiles_per_wave - rel);
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(mcache);
                    uint32_t incl = cnt;
#pragma unroll
                    for (uint32_t d = 1; d < 64; d <<= 1) {
                        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
                        if (lane >= d) incl += t;
                    }
                    const float* fptr = p.i8_scale;  // lane j of a 16-lane row fetches: 0 the row's scale, 1 its magnitude / |v~|^2, 2 |v| (estimator B)
                    if constexpr (METRIC == NMN_METRIC_COSINE) fptr = j == 1 ? p.norms : fptr;
                    if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) fptr = j == 1 ? p.i8_vv : (j == 2 ? p.norms : fptr);
                    const bool f_lane = j == 0 || (METRIC == NMN_METRIC_COSINE && j == 1) ||
                                        (METRIC == NMN_METRIC_EUCLIDEAN && (j == 1 || (j == 2 && any_est_b)));
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
                        // one step: entry -> row address -> loads
                        auto entry_of = [&](uint32_t s0) -> uint32_t { return s0 + grp < S ? (uint32_t)wtab[s0 + grp] : 0xFFFFu; };
                        auto row_of = [&](uint32_t e) -> uint64_t { return (uint64_t)tile_at(rel + (e >> 6)) * kTileRows + (e & 63u); };
                        auto finish = [&](uint32_t s0, uint32_t e, const float (&acc)[NQ], float f) {
                            const bool active = e != 0xFFFFu;
                            const float sr = row_share<0>(f);
                            float vn = 1.f, vb = 0.f;
                            if constexpr (METRIC != NMN_METRIC_DOT_PRODUCT) vn = row_share<1>(f);
                            if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vb = row_share<2>(f);
                            (void)vb;
#pragma unroll
                            for (int q = 0; q < NQ; q++) {
                                const float dot = row16_sum(acc[q]) * (qsc[q] * sr);
                                float sc;
                                if constexpr (METRIC == NMN_METRIC_COSINE) sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : dot / (qmag[q] * vn);
                                else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) sc = qq8[q] < 0.f ? l2_from_dot(qmag[q] * qmag[q], vb * vb, dot, neg) : l2_from_dot(qq8[q], vn, dot, neg);
                                else sc = dot;
                                if (j == 0 && active) wsc[q * kWalkRows + s0 + grp] = f2u(sc);
                            }
                        };
                        if constexpr (SINGLE && CH <= NMN_I8_WALK_PIPE_CH) {  // (longer rows: a step is >= 4 KiB in flight per wave as it is,
                            v4i xa[CH], xb[CH];                                //  and the second buffer would cost a wave per SIMD)
                            float fa = 0.f, fb = 0.f;
                            uint32_t ea = entry_of(0), eb = 0xFFFFu;
                            auto issue = [&](uint32_t e, v4i (&x)[CH], float& f) {
                                if (e != 0xFFFFu) {
                                    const uint64_t row = row_of(e);
                                    load_row<CH>(reinterpret_cast<const v4i*>(mat + row * (uint64_t)ld), j, x);
                                    f = f_lane ? fptr[row] : 0.f;
                                } else {
#pragma unroll
                                    for (int c = 0; c < CH; c++) x[c] = (v4i){0, 0, 0, 0};
                                    f = 0.f;
                                }
                            };
                            issue(ea, xa, fa);
                            for (uint32_t s0 = 0; s0 < S; s0 += 8u) {
                                eb = entry_of(s0 + 4u);
                                issue(eb, xb, fb);
                                __builtin_amdgcn_sched_barrier(0);
                                {
                                    float acc[NQ];
                                    dot_row<NQ, CH, QREG>(xa, j, chunks, qs4, qh, ql, acc);
                                    finish(s0, ea, acc, fa);
                                }
                                ea = entry_of(s0 + 8u);
                                issue(ea, xa, fa);
                                __builtin_amdgcn_sched_barrier(0);
                                {
                                    float acc[NQ];
                                    dot_row<NQ, CH, QREG>(xb, j, chunks, qs4, qh, ql, acc);
                                    finish(s0 + 4u, eb, acc, fb);
                                }
                            }
                        } else if (!SINGLE && CH == 1 && narrow) {
                            // 128-element rows: EIGHT listed rows per step, eight lanes each (lanes 0 / 1 / 2 of a row's eight fetch its
                            // factors), two steps' loads in flight
                            const uint32_t j8 = lane & 7u, g8 = lane >> 3, base8 = lane & ~7u;
                            const float* fptr8 = p.i8_scale;
                            if constexpr (METRIC == NMN_METRIC_COSINE) fptr8 = j8 == 1 ? p.norms : fptr8;
                            if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) fptr8 = j8 == 1 ? p.i8_vv : (j8 == 2 ? p.norms : fptr8);
                            const bool f_lane8 = j8 == 0 || (METRIC == NMN_METRIC_COSINE && j8 == 1) ||
                                                 (METRIC == NMN_METRIC_EUCLIDEAN && (j8 == 1 || (j8 == 2 && any_est_b)));
                            for (uint32_t s0 = 0; s0 < S; s0 += 16u) {
                                uint32_t e2[2];
                                v4i x2[2];
                                float f2[2];
#pragma unroll
                                for (int h = 0; h < 2; h++) {
                                    const uint32_t idx = s0 + (uint32_t)h * 8u + g8;
                                    e2[h] = idx < S ? (uint32_t)wtab[idx] : 0xFFFFu;
                                    const uint64_t row = e2[h] != 0xFFFFu ? row_of(e2[h]) : 0ull;
                                    x2[h] = (v4i){0, 0, 0, 0};
                                    f2[h] = 0.f;
                                    if (e2[h] != 0xFFFFu) {
                                        x2[h] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(mat + row * (uint64_t)ld) + j8);
                                        if (f_lane8) f2[h] = fptr8[row];
                                    }
                                }
#pragma unroll
                                for (int h = 0; h < 2; h++) {
                                    const bool active = e2[h] != 0xFFFFu;
                                    const float sr = __shfl(f2[h], (int)base8);
                                    float vn = 1.f, vb = 0.f;
                                    if constexpr (METRIC != NMN_METRIC_DOT_PRODUCT) vn = __shfl(f2[h], (int)base8 + 1);
                                    if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vb = __shfl(f2[h], (int)base8 + 2);
                                    (void)vb;
#pragma unroll
                                    for (int q = 0; q < NQ; q++) {
                                        int hi = 0, lo = 0;
                                        dot16(x2[h], qs4[(uint32_t)q * 2u * chunks + j8], qs4[(uint32_t)q * 2u * chunks + chunks + j8], hi, lo);
                                        float t = (float)hi + (float)lo * 0.00390625f;
                                        t += __shfl_xor(t, 1);
                                        t += __shfl_xor(t, 2);
                                        t += __shfl_xor(t, 4);
                                        const float dot = t * (qsc[q] * sr);
                                        float sc;
                                        if constexpr (METRIC == NMN_METRIC_COSINE) sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : dot / (qmag[q] * vn);
                                        else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) sc = qq8[q] < 0.f ? l2_from_dot(qmag[q] * qmag[q], vb * vb, dot, neg) : l2_from_dot(qq8[q], vn, dot, neg);
                                        else sc = dot;
                                        if (j8 == 0 && active) wsc[q * kWalkRows + s0 + (uint32_t)h * 8u + g8] = f2u(sc);
                                    }
                                }
                            }
                        } else {
                            for (uint32_t s0 = 0; s0 < S; s0 += 4u) {
                                const uint32_t e = entry_of(s0);
                                const bool active = e != 0xFFFFu;
                                const uint64_t row = active ? row_of(e) : 0ull;
                                const float f = (active && f_lane) ? fptr[row] : 0.f;
                                float acc[NQ];
                                row_partial_i8<NQ, CH, SINGLE, QREG>(reinterpret_cast<const v4i*>(mat + row * (uint64_t)ld), active, j, chunks, qs4, qh, ql, acc);
                                finish(s0, e, acc, f);
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
            if (mword == 0ull) {  // nobody of this tile takes part: its maximum says so, which is all anybody reads of it
                if (lane < (uint32_t)NQ && q0 + lane < p.nq) p.tmax[(uint64_t)(q0 + lane) * p.tmax_stride + tile] = kKeyMasked;
                continue;
            }
        }
        // lane L finishes tile row (L & 15) * 4 + (L >> 4).  Its scale (and magnitude) are requested HERE, before the tile's rows:
        // a sparse tile is a chain of dependent memory round trips per wave (bitmap -> rows -> these factors -> store), and
        // at selectivity 0.1 a tile is two row steps — the factors' trip alone was a third of the chain.
        // (rows of 128 elements, no bitmap: eight lanes per row, eight rows per step — see `narrow` below; lane L finishes row (L & 7) * 8 + (L >> 3))
        const uint32_t mybit = narrow ? (lane & 7u) * 8u + (lane >> 3) : j * 4u + grp;
        const bool valid = ((mword >> mybit) & 1ull) != 0;
        const uint64_t myrow = r0 + mybit;
        const uint64_t srow = valid ? myrow : r0;  // (rows that do not take part: any valid address of the tile)
        const float sr_raw = p.i8_scale[srow];
        float vn_raw = 1.f;  // cosine: |v| in reference order; Euclidean: |v~|^2 of the row as stored
        if constexpr (METRIC == NMN_METRIC_COSINE) vn_raw = p.norms[srow];
        if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vn_raw = p.i8_vv[srow];
        float vb = 0.f;  // (Euclidean, estimator B of some query: the row's exact magnitude)
        if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) {
            if (any_est_b) vb = p.norms[srow];  // (wave-uniform)
        }
        (void)vb;

        float mydot[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) mydot[q] = 0.f;

        bool compacted = false;
        if constexpr (MASKED) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(mword);
            if (cnt <= kCompactMaxRows && !narrow) {  // wave-uniform: sparse tile, every step reads 4 participating rows
                compacted = true;
                float* stg = reinterpret_cast<float*>(qsi + (size_t)NQ * (ld >> 1)) + (threadIdx.x >> 6) * (NQ * 64);
                uint32_t* tab = reinterpret_cast<uint32_t*>(qsi + (size_t)NQ * (ld >> 1) + 4 * (NQ * 64)) + (threadIdx.x >> 6) * 64;
                const bool myset = ((mword >> lane) & 1ull) != 0;
                const uint32_t myrank = (uint32_t)__builtin_popcountll(mword & ((1ull << lane) - 1ull));
                if (myset) tab[myrank] = lane;
                __builtin_amdgcn_wave_barrier();
                // (two steps' loads in flight together — 8 rows per trip — was built and measured: 140 VGPRs, three waves per SIMD
                //  instead of five, every selectivity 5-20 % slower; the sparse sweep lives on occupancy)
                for (uint32_t s0 = 0; s0 < cnt; s0 += 4u) {
                    const bool active = s0 + grp < cnt;
                    const uint32_t pos = active ? tab[s0 + grp] : 0u;
                    const v4i* rowp = reinterpret_cast<const v4i*>(mat + (r0 + pos) * (uint64_t)ld);
                    float acc[NQ];
                    row_partial_i8<NQ, CH, SINGLE, QREG>(rowp, active, j, chunks, qs4, qh, ql, acc);
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const float t = row16_sum(acc[q]);
                        if (j == 0 && active) stg[q * 64 + (int)pos] = t;
                    }
                }
                if (valid) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) mydot[q] = stg[q * 64 + (int)mybit];
                }
            }
        }
        if (narrow) {
            // 128-byte rows: the 16-lanes-per-row mapping would idle half of every load (0.38 of peak at 10M x 128); 384-byte rows a
            // quarter (one and a half 256-byte groups).  Here a row is 8 lanes x L8 16-byte loads (L8 = 1 / 3), a step is 8 rows, and
            // steps are issued kG at a time (128: the tile's 8 steps, 8 KiB per wave; 384: 4 steps, 12 KiB) before the first product.
            const uint32_t j8 = lane & 7u, g8 = lane >> 3;
            auto narrow_tile = [&](auto l8c) __attribute__((always_inline)) {
                constexpr int L8 = decltype(l8c)::value;
                constexpr int kG = (L8 == 1 ? 8 : 4) / (MASKED ? 2 : 1);  // (under a bitmap half as many: the walk next door lives on occupancy)
#pragma unroll
                for (int s0 = 0; s0 < 8; s0 += kG) {
                    v4i x8[kG][L8];
#pragma unroll
                    for (int st = 0; st < kG; st++) {
                        const uint32_t rbit = (uint32_t)(s0 + st) * 8u + g8;
                        const bool on = !MASKED || ((mword >> rbit) & 1ull) != 0;  // (rows the bitmap excludes are not read)
                        const v4i* rowp = reinterpret_cast<const v4i*>(mat + (r0 + rbit) * (uint64_t)ld);  // (the mirror is allocated in whole tiles)
#pragma unroll
                        for (int c = 0; c < L8; c++) {
                            if (on) x8[st][c] = __builtin_nontemporal_load(rowp + (uint32_t)c * 8u + j8);
                            else x8[st][c] = (v4i){0, 0, 0, 0};
                        }
                    }
#pragma unroll
                    for (int st = 0; st < kG; st++) {
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            int hi = 0, lo = 0;
#pragma unroll
                            for (int c = 0; c < L8; c++)
                                dot16(x8[st][c], qs4[(uint32_t)q * 2u * chunks + (uint32_t)c * 8u + j8],
                                      qs4[(uint32_t)q * 2u * chunks + chunks + (uint32_t)c * 8u + j8], hi, lo);
                            float t = (float)hi + (float)lo * 0.00390625f;
                            t += __shfl_xor(t, 1);
                            t += __shfl_xor(t, 2);
                            t += __shfl_xor(t, 4);
                            if (j8 == (uint32_t)(s0 + st)) mydot[q] = t;
                        }
                    }
                }
            };
            if constexpr (!SINGLE && CH == 1) narrow_tile(std::integral_constant<int, 1>{});  // (the kernel of 128-element rows)
            else if constexpr (!SINGLE && CH == 2 && !MASKED) narrow_tile(std::integral_constant<int, 3>{});  // (... of 384-element rows)
        } else if (!compacted) {
            constexpr int kSteps = CH <= 3 ? 4 : 2;  // row steps whose loads are in flight together (>= 6 x 16 B per lane)
#pragma unroll kSteps
            for (uint32_t s = 0; s < 16; s++) {
                if constexpr (MASKED) {
                    if (((mword >> (s * 4)) & 0xFull) == 0) continue;  // wave-uniform: 4 rows all excluded
                }
                const uint32_t rbit = s * 4 + grp;
                const bool active = MASKED ? ((mword >> rbit) & 1ull) != 0 : true;
                const v4i* rowp = reinterpret_cast<const v4i*>(mat + (r0 + rbit) * (uint64_t)ld);
                float acc[NQ];
                row_partial_i8<NQ, CH, SINGLE, QREG>(rowp, active, j, chunks, qs4, qh, ql, acc);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const float t = row16_sum(acc[q]);
                    if (j == s) mydot[q] = t;
                }
            }
        }

        const float sr = valid ? sr_raw : 0.f;
        float vn = 1.f;
        if constexpr (METRIC == NMN_METRIC_COSINE) vn = valid ? vn_raw : 1.f;
        if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) vn = valid ? vn_raw : 0.f;
        // (the arithmetic runs for every query slot, only the stores are guarded: see the pipelined loop above)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const bool q_ok = q0 + q < p.nq;
            const float dot = mydot[q] * (qsc[q] * sr);
            float sc;
            if constexpr (METRIC == NMN_METRIC_COSINE) sc = (vn == 0.f || qmag[q] == 0.f) ? 0.f : dot / (qmag[q] * vn);
            else if constexpr (METRIC == NMN_METRIC_EUCLIDEAN) sc = qq8[q] < 0.f ? l2_from_dot(qmag[q] * qmag[q], vb * vb, dot, neg) : l2_from_dot(qq8[q], vn, dot, neg);
            else sc = dot;
            const uint32_t key = valid ? score_to_key(sc) : kKeyMasked;
            const uint32_t m = wave_max_u32(key);
            if (q_ok) {
                p.scores[score_at(myrow, q0 + q, p.nql)] = valid ? f2u(sc) : kScoreSentinelBits;
                if (lane == 0) p.tmax[(uint64_t)(q0 + q) * p.tmax_stride + tile] = m;
                wmax[q] = max(wmax[q], m);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
            if (q0 + q < p.nq) p.wmax[(size_t)(q0 + q) * p.wmax_stride + wave] = wmax[q];
    }
}

template <int METRIC, bool MASKED, int NQ, int CH, bool SINGLE>
hipError_t launch_one(const ScanParams& p, hipStream_t s) {
    const uint32_t waves = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    dim3 grid((waves + 3) / 4, (p.nq + NQ - 1) / NQ);
    // query planes (2 bytes per element), plus (masked) the per-wave staging arrays and rank tables of the compacted tiles
    // and (survivor walk) a row list of kWalkRows 16-bit entries and kWalkRows parked scores per query, per wave
    const size_t lds = (size_t)NQ * p.ld * 2 + (MASKED ? 4 * NQ * 64 * sizeof(float) + 4 * 64 * sizeof(uint32_t) +
                                                             4 * kWalkRows * sizeof(uint16_t) + 4 * NQ * kWalkRows * sizeof(uint32_t)
                                                       : 0);
    auto kern = scan_i8_kernel<METRIC, MASKED, NQ, CH, SINGLE>;
    // (workgroups resident per CU, by way of the LDS request — nmn_scan.hip's launch_one has the story; here a knob only)
    static const int wgs_per_cu = [] { const char* e = getenv("NMN_SCAN_I8_WGS_PER_CU"); return e ? atoi(e) : 0; }();
    size_t lds_total = lds;
    if (wgs_per_cu > 0 && grid.x > 256u * (unsigned)wgs_per_cu)  // (masked or not)
        lds_total = std::max<size_t>(lds, ((size_t)160 * 1024 / (size_t)(wgs_per_cu + 1) + 4096) & ~(size_t)1023);
    if (lds_total > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_total, s, p);
    return hipGetLastError();
}

template <int METRIC, bool MASKED, int NQ>
hipError_t launch_layout(const ScanParams& p, hipStream_t s) {
    if (p.ld & 255u) {  // 384, 640, 896 ...: the generic loop with a guarded last group — up to 1408 in ONE pass of ceil(chunks / 16) loads per lane
        switch ((p.ld + 255u) >> 8) {
            case 2: return launch_one<METRIC, MASKED, NQ, 2, false>(p, s);
            case 3: return launch_one<METRIC, MASKED, NQ, 3, false>(p, s);
            case 4: return launch_one<METRIC, MASKED, NQ, 4, false>(p, s);
            case 5: return launch_one<METRIC, MASKED, NQ, 5, false>(p, s);
            case 6: return launch_one<METRIC, MASKED, NQ, 6, false>(p, s);
            default: return launch_one<METRIC, MASKED, NQ, 1, false>(p, s);
        }
    }
    switch (p.ld >> 8) {  // chunks / 16 = 256-element groups per row
        case 1: return launch_one<METRIC, MASKED, NQ, 1, true>(p, s);
        case 2: return launch_one<METRIC, MASKED, NQ, 2, true>(p, s);
        case 3: return launch_one<METRIC, MASKED, NQ, 3, true>(p, s);
        case 4: return launch_one<METRIC, MASKED, NQ, 4, true>(p, s);
        case 5: return launch_one<METRIC, MASKED, NQ, 5, true>(p, s);
        case 6: return launch_one<METRIC, MASKED, NQ, 6, true>(p, s);
        default: break;
    }
    if ((p.ld >> 8) % 6 == 0) return launch_one<METRIC, MASKED, NQ, 6, false>(p, s);
    if ((p.ld >> 8) % 4 == 0) return launch_one<METRIC, MASKED, NQ, 4, false>(p, s);
    return launch_one<METRIC, MASKED, NQ, 1, false>(p, s);
}

template <int METRIC>
hipError_t launch_metric(const ScanParams& p, hipStream_t s) {
    if (p.mask) return p.nq >= 2 ? launch_layout<METRIC, true, 2>(p, s) : launch_layout<METRIC, true, 1>(p, s);
    return p.nq >= 2 ? launch_layout<METRIC, false, 2>(p, s) : launch_layout<METRIC, false, 1>(p, s);
}

// ---- writing the mirror -----------------------------------------------------------------------------------------------
// `lpr` lanes (a power of two <= 64) share one row: pass 1 finds max |v| (the row's scale), pass 2 re-reads the row (it is
// in L2), rounds v / s to the nearest integer in [-127, 127], accumulates |v - s c|^2 and stores 8 codes per 8 elements.
// A row with a non-finite element gets scale 0 / codes 0 and an infinite error norm: its shard's margin becomes useless,
// every query overflows into the f32 retry and the shard leaves the mirror alone (nmn_api.hip) — never a wrong answer.
__global__ void __launch_bounds__(256) q8_rows_kernel(const float* __restrict__ corpus, int8_t* __restrict__ q8, float* __restrict__ scale,
                                                      float* __restrict__ vv, float* __restrict__ cosf, const float* __restrict__ norms,
                                                      uint32_t ld, uint64_t row0, uint64_t n, float* __restrict__ row_err2, uint32_t lpr) {
    const uint32_t per_row = ld >> 3;  // 8-element groups per row
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & (lpr - 1u), slot = lane / lpr, rows_per_wave = 64u / lpr;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t rb = wave * rows_per_wave; rb < n; rb += n_waves * rows_per_wave) {
        const uint64_t ri = rb + slot;
        const bool live = ri < n;
        const uint64_t r = row0 + (live ? ri : 0);
        float mx = 0.f;
        bool bad = false;
        for (uint32_t g = sub; live && g < per_row; g += lpr) {
            const v4f a = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u);
            const v4f b = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u + 4u);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const float ax = __builtin_fabsf(x[t]);
                bad = bad || !(ax <= 3.0e38f);  // inf or NaN
                mx = __builtin_fmaxf(mx, ax);
            }
        }
        for (uint32_t off = lpr >> 1; off > 0; off >>= 1) {
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, (int)off));
            bad = bad || (__shfl_xor((int)bad, (int)off) != 0);
        }
        const float s = (bad || mx == 0.f) ? 0.f : mx / 127.0f;
        const float inv = s > 0.f ? 127.0f / mx : 0.f;
        float err2 = 0.f;
        int cc = 0;  // c.c of this lane's codes: an exact integer (<= 4096 * 127^2 over the whole row)
        for (uint32_t g = sub; live && g < per_row; g += lpr) {
            const v4f a = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u);
            const v4f b = *reinterpret_cast<const v4f*>(corpus + r * ld + g * 8u + 4u);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t pk[2] = {0u, 0u};
#pragma unroll
            for (int t = 0; t < 8; t++) {
                float c = __builtin_rintf(x[t] * inv);
                c = __builtin_fminf(__builtin_fmaxf(c, -127.0f), 127.0f);
                if (!(c == c)) c = 0.f;
                const float e = x[t] - s * c;
                err2 = __builtin_fmaf(e, e, err2);
                cc += (int)c * (int)c;
                pk[t >> 2] |= ((uint32_t)(int)c & 0xFFu) << (8 * (t & 3));
            }
            *reinterpret_cast<uint2*>(q8 + r * (uint64_t)ld + g * 8u) = make_uint2(pk[0], pk[1]);
        }
        for (uint32_t off = lpr >> 1; off > 0; off >>= 1) {
            err2 += __shfl_xor(err2, (int)off);
            cc += __shfl_xor(cc, (int)off);
        }
        if (live && sub == 0) {
            scale[r] = s;
            vv[r] = (s * s) * (float)cc;
            const float vn = norms[r];
            cosf[r] = (vn > 0.f && vn <= 3.0e38f) ? s / vn : 0.f;  // cosine factor of the batched sweep (a zero row scores 0)
            row_err2[ri] = bad ? __builtin_inff() : err2;
        }
    }
}

// err_bits[0] = max_r |e_r|, err_bits[1] = max_r |e_r| / |v_r| (with slack for the order of the f32 sums that produced them)
__global__ void __launch_bounds__(256) q8_err_kernel(const float* __restrict__ row_err2, const float* __restrict__ norms, uint64_t row0,
                                                     uint64_t n, uint32_t* __restrict__ err_bits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float e = __builtin_sqrtf(row_err2[i]) * 1.0005f;
        const float vn = norms[row0 + i];
        if (e > 0.f) {  // (+inf included; e >= 0: bit order == value order)
            atomicMax(err_bits, __float_as_uint(e));
            if (vn > 0.f || !(vn == vn)) {
                float rel = e / vn * 1.0005f;
                if (!(rel == rel)) rel = __builtin_inff();  // a non-finite row (inf / inf): no margin vouches for it
                atomicMax(err_bits + 1, __float_as_uint(rel));
            }
        }
    }
}

}  // namespace

bool scan_i8_supported(uint32_t ld, uint32_t dim, int metric) {
    (void)dim;
    // rows of whole 256-element groups (one 16-byte chunk per lane of a 16-lane row group), up to the 4096 of the widest sweep;
    // also rows of an odd number of 128-element halves (half of the last group's lanes idle: 384 reads 1.5 load
    // slots per row for 384 bytes — still fewer bytes than the 768 of the bf16 mirror)
    static const uint32_t min_odd = [] { const char* e = getenv("NMN_I8_MIN_ODD_LD"); return e ? (uint32_t)atol(e) : 128u; }();
    if (ld == 0 || ld % 128u != 0 || ld > 4096u) return false;
    if (ld % 256u != 0 && ld < min_odd) return false;
    return metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2;
}

hipError_t launch_scan_i8(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_metric<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN:
        case kMetricNegL2: return launch_metric<NMN_METRIC_EUCLIDEAN>(p, s);
        default: return launch_metric<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

hipError_t launch_q8_rows(const float* corpus, int8_t* q8, float* scale, float* vv, float* cosf, uint32_t ld, uint64_t row0, uint64_t n, const float* norms,
                          float* row_err2_scratch, uint32_t* err_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t lpr = 1;
    while (lpr < 64u && lpr < (ld >> 3)) lpr <<= 1;
    hipLaunchKernelGGL(q8_rows_kernel, dim3(256 * 16), dim3(256), 0, s, corpus, q8, scale, vv, cosf, norms, ld, row0, n, row_err2_scratch, lpr);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(q8_err_kernel, dim3(blocks), dim3(256), 0, s, row_err2_scratch, norms, row0, n, err_bits);
    return hipGetLastError();
}

}  // namespace nmn
