// nmn_scan_i8b.hip — batched queries (3..64 per sweep) over the 8-bit mirror: queries in LDS, rows straight into registers.
// OPT-IN (NMN_I8B=1): exact on the whole batched suite, measured SLOWER than the LDS-ring kernel it was built to beat
// (1.65-1.73 vs 1.46-1.50 ms per 64-query sweep of 10M x 768; no batch size at which it wins) — docs/kernel-scan-i8b.md has the
// account, profiles/r04q_*, r04x_* the numbers.  Kept because it is the other point in the design space, and measured.
//
// The matrix-core sweep of nmn_scan_mfma.hip keeps the QUERIES in registers (a wave owns 16 of them for the whole row)
// and streams the ROWS through an LDS ring that all four waves of a workgroup read: every stage is a counted wait and a
// workgroup barrier, every wave reads every stage (4x the stage bytes of LDS traffic), and a tile's epilogue in any wave
// holds up the other three at the next barrier.
//
// This kernel turns the two operands round:
//   * the QUERIES of the pass (int8 planes h, l of q = s_q (h + l / 256) + e_q, qprep_kernel; NG = ceil(nq / 16) groups of 16)
//     sit in LDS in fragment order, [k-step][query group, plane][lane] x 16 B — 96 KiB at 768 elements and 64 queries, written
//     once per workgroup, read-only afterwards: ds_read_b128 at lane * 16, conflict-free;
//   * the ROWS go from HBM into registers as MFMA A-fragments (buffer_load_dwordx4 with a per-tile descriptor: lane (n = lane & 15,
//     g = lane >> 4) holds bytes [64 ks + 16 g, +16) of row n of a 16-row block — 16 rows x 64 B per instruction; the two halves
//     of a row's 128-byte line are asked for back to back, as a pair of k-steps).  A wave owns WHOLE tiles (64 rows x all the
//     queries), worked as two half-tiles of 32 rows with an accumulator set each; the loads of the NEXT tile's k-step go out
//     right behind the MFMAs of this tile's, into the registers those just freed: a tile (48 KiB per wave, 192 KiB per CU) is
//     always asked for, and nothing in the loop is shared between waves — no barrier, no counted hand-over;
//   * the EPILOGUE of half-tile X - 1 (int32 -> f32, row factors, running maxima, keys, score writes) is issued in slices between
//     the MFMAs of half-tile X (see the kernel's comment); the MFMAs are inline asm so that their accumulators stay in
//     architectural VGPRs, where VALU instructions can read them, and the rows in AGPRs, where the loads put them;
//   * a wave is a "scan wave" of the selection (select_kernel): contiguous tile range, one wmax entry — 4096 of them, as in
//     the VALU sweeps.  Workgroups exist only to share the LDS copy of the queries.
// Outputs (scores / tmax / wmax, the sampling pass, skip_key, launch ranges) are those of scan_mfma_kernel<.., I8 = true>:
// the rest of the chain does not know which of the two ran.  Margins: qprep_kernel, approx_pass 1 | 2 | 4 — the products
// are exact integers either way, the epilogue's float arithmetic is the ring kernel's.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "nmn_internal.h"

#ifdef NMN_I8B_TIMING  // measurement build: per scan wave, cycles in the k-loop / in the epilogue / in all, tiles (s_memtime)
__device__ unsigned long long nmn_i8b_dbg[4096 * 8];
extern "C" int nmn_i8b_debug_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(nmn_i8b_dbg), sizeof(nmn_i8b_dbg));
}
#endif

namespace nmn {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kI8bWaves = 4;  // waves per workgroup: one per SIMD, up to 512 registers each
#ifndef NMN_I8B_RING   // registers of query fragments in flight between LDS and the matrix cores (x 4 VGPRs)
#define NMN_I8B_RING 4
#endif
#ifndef NMN_I8B_AHEAD  // fragments a ds_read_b128 is issued ahead of its MFMAs (< NMN_I8B_RING)
#define NMN_I8B_AHEAD 3
#endif

template <bool NEG>
__device__ __forceinline__ float l2_score_q(float qq8, float vv, float dot) {  // |q~ - v~|^2 = |q~|^2 + |v~|^2 - 2 q~.v~
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, dot, qq8 + vv), 0.0f);
    const float d = __builtin_amdgcn_sqrtf(d2);
    return NEG ? -d : __builtin_amdgcn_rcpf(1.0f + d);
}

// Row loads are BUFFER loads: a 128-bit descriptor in SGPRs (rebuilt per tile by scalar adds), ONE 32-bit lane offset for all
// of them, the row block in the scalar offset and the k-step in the instruction's immediate — no 64-bit address arithmetic
// in the vector unit, no address registers to spill; a descriptor of zero records makes the loads behind a wave's last tile
// return zeros without touching memory.
constexpr uint32_t kRsrcFlags = 0x00020000u;  // gfx9-family raw buffer: DATA_FORMAT = 32 bit
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, kRsrcFlags);
}
template <int POLICY>
__device__ __forceinline__ u4 load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, POLICY ? 2 : 0);  // aux 2: non-temporal
}

// KSTEPS = row bytes / 64 (k-steps of a row); POLICY: cache policy of the row loads (0 default, 1 non-temporal)
//
// One wave per SIMD, and the epilogue INSIDE the matrix-core stream.  A tile is two half-tiles of 32 rows (two 16-row
// blocks); each half has its own accumulator set.  While the MFMAs of half X fill set X & 1, the VALU work of half X - 1 —
// int32 -> f32, the row factors, the running maxima, the keys — is issued in slices between them (a v_mfma_i32_16x16x64_i8
// holds the matrix pipe for 16 cycles and the issue port for 4: one wave has room for two or three other instructions per
// MFMA).  The wave's time per tile is then the MFMAs' alone (~6 100 cycles against a memory period of ~16 000 per tile and
// wave), and it is never away from its load stream for longer than a fragment.  (Measured on the forms this replaces —
// epilogue behind the k-loop: one wave per SIMD 1.70 ms per 64-query sweep of 10M x 768, two waves per SIMD 1.60-1.69, without
// the epilogue 1.32-1.38; the LDS-ring kernel 1.46-1.49.)
// NG: query groups of 16 the pass holds (1 .. 4): everything per query group — LDS fragments, MFMAs, accumulators, epilogue slices —
// scales with it; a pass of <= 16 queries does a quarter of the matrix-core and VALU work of a pass of 64 for the same bytes.
template <int KSTEPS, int METRIC, bool MASKED, int POLICY, int NG>
__global__ void __launch_bounds__(kI8bWaves * 64, 1) scan_i8b_kernel(ScanParams p) {
    constexpr int NF = 2 * NG;  // query fragments per k-step: [group][plane h, l]
    constexpr bool kL2 = METRIC == NMN_METRIC_EUCLIDEAN || METRIC == kMetricNegL2;
    constexpr bool kScaled = METRIC == NMN_METRIC_COSINE || METRIC == NMN_METRIC_DOT_PRODUCT;
    constexpr int kFragsRow = KSTEPS * NF;  // query fragments per row: [k-step][group][plane h, l]
    constexpr int kIdx = 2 * KSTEPS;       // k-steps of a tile: two half-tiles of 32 rows, KSTEPS each = the rows in flight per wave
    constexpr int R = NMN_I8B_RING, LA = NMN_I8B_AHEAD;
    static_assert(LA < R && kFragsRow % R == 0 && KSTEPS % 2 == 0 && kFragsRow >= 24 * NG, "rings; the epilogue's 5 NG slices need 24 NG fragment slots");
    constexpr uint32_t ld = KSTEPS * 64;   // bytes per row of the mirror (= p.ld elements)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    u4* const bq = reinterpret_cast<u4*>(lds_u);                       // [kFragsRow][64 lanes] x 16 B
    uint32_t* const tk_pend = lds_u + (uint32_t)kFragsRow * 64u * 4u;  // pending tile maxima: [wave][group][16 queries][4 tiles]

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g = lane >> 4, n = lane & 15u;
#ifdef NMN_I8B_TIMING
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif

    // ---- the queries: fragment order, zero for queries beyond nq --------------------------------------------------
    {
        const char* qb = reinterpret_cast<const char*>(p.qi8);
        for (uint32_t e = threadIdx.x; e < (uint32_t)kFragsRow * 64u; e += (uint32_t)kI8bWaves * 64u) {
            const uint32_t ks = (e >> 6) / (uint32_t)NF, f = (e >> 6) % (uint32_t)NF, l = e & 63u;
            const uint32_t q = (f >> 1) * 16u + (l & 15u), pl = f & 1u;
            u4 v = {0u, 0u, 0u, 0u};
            if (q < p.nq) v = *reinterpret_cast<const u4*>(qb + ((size_t)q * 2u + pl) * ld + ks * 64u + (l >> 4) * 16u);
            bq[e] = v;
        }
    }
    __syncthreads();

    // ---- this wave's tiles ---------------------------------------------------------------------------------------
    const uint32_t wv = blockIdx.x * (uint32_t)kI8bWaves + wave;  // wave of this launch
    if (p.bx_count && wv >= p.bx_count) return;
    const uint32_t sw = p.bx_base + wv;                  // scan wave: index into wmax, owner of tiles [t0, t1)
    const uint32_t tstep = p.tile_step;                  // 1, or S on the sampling pass (tile index i -> tile i * S)
    const bool sampling = tstep > 1;
    const uint32_t t0 = sw * p.tiles_per_wave;
    if (t0 >= p.n_tiles) return;
    const uint32_t t1 = min(t0 + p.tiles_per_wave, p.n_tiles);

    const char* const mirror = reinterpret_cast<const char*>(p.corpus_i8);
    constexpr uint32_t kTileBytes = kTileRows * ld;
    auto tile_rsrc = [&](uint32_t tile_, bool on) -> __amdgpu_buffer_rsrc_t {
        return make_rsrc(mirror + (uint64_t)tile_ * tstep * kTileBytes, on ? kTileBytes : 0u);
    };
    const uint32_t voff = n * ld + g * 16u;  // row n of a 16-row block, 16-byte chunk g of a k-step

    // per-lane constants of the four query groups: C column n of group H is query H * 16 + n
    // (kept small on purpose: everything else about a query — its number, its addresses — is re-derived from n where it is used)
    uint32_t skip_h[NG], wmax_h[NG];
    float invq_h[NG], qq_h[kL2 ? NG : 1];
#pragma unroll
    for (int h = 0; h < NG; h++) {
        const uint32_t qn = (uint32_t)h * 16u + n;
        const bool ok = qn < p.nq;
        const float qmag = ok ? p.qinfo[qn].qmag : 0.f;
        const float qsc = ok ? p.qinfo[qn].qscale : 0.f;
        // per-query factor: s_q / |q| (cosine), s_q (dot product; also what scales the Euclidean dot)
        invq_h[h] = METRIC == NMN_METRIC_COSINE ? (qmag == 0.f ? 0.f : qsc * __builtin_amdgcn_rcpf(qmag)) : qsc;
        if constexpr (kL2) qq_h[h] = ok ? p.qinfo[qn].qq8 : 0.f;
        skip_h[h] = (ok && p.skip_key) ? p.skip_key[qn] : kKeyNaN;  // kKeyNaN: write every tile
        wmax_h[h] = kKeyMasked;
    }
    // per-row factor of the epilogue: s_r / |v| (cosine) or s_r (dot, Euclidean); Euclidean also |v~|^2 of the row as stored
    // (buffer loads like the rows: descriptor over the whole array, the half-tile in the scalar offset, g * 16 bytes per lane)
    const uint32_t fbytes = (uint32_t)min((uint64_t)p.n_tiles * tstep * kTileRows * 4ull, 0xFFFFFFFFull);
    const __amdgpu_buffer_rsrc_t rf_rs = make_rsrc(METRIC == NMN_METRIC_COSINE ? p.i8_cos : p.i8_scale, fbytes);
    const __amdgpu_buffer_rsrc_t rv_rs = make_rsrc(p.i8_vv, fbytes);
    (void)rv_rs;

    // Row factors of a half-tile: asked for a whole TILE before the half begins (loads return in issue order and a tile of rows
    // is always in flight: a factor load lands right before the rows issued behind it), used while the NEXT half multiplies.
    // Per half-parity two sets: the one in use / arrived (fc) and the one in flight (fn).
    struct Factors {
        f4 rf[2];
        f4 rv[kL2 ? 2 : 1];
    };
    auto load_factors = [&](uint32_t tile_, uint32_t half_, Factors& F_) __attribute__((always_inline)) {
        // (scalar offset: real tile index — on the sampling pass every tstep-th — x 256 bytes of factors, + the half)
        const uint32_t so = __builtin_amdgcn_readfirstlane(tile_ * tstep * 256u + half_ * 128u);
#ifdef NMN_I8B_NO_FACTOR_LOADS  // measurement only (answers are wrong)
        F_.rf[0] = F_.rf[1] = (f4){1.f, 1.f, 1.f, 1.f};
        if constexpr (kL2) F_.rv[0] = F_.rv[1] = (f4){1.f, 1.f, 1.f, 1.f};
        (void)so;
        return;
#endif
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            F_.rf[rb] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rf_rs, g * 16u + (uint32_t)rb * 64u, so, 0));
            if constexpr (kL2) F_.rv[rb] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rv_rs, g * 16u + (uint32_t)rb * 64u, so, 0));
        }
    };
    // this lane's 8 rows of half `half_` of a tile: bit (rb * 4 + e) of the result <- bit (half * 32 + rb * 16 + g * 4 + e) of the word
    auto lane_rows = [&](uint64_t word, int half_) __attribute__((always_inline)) -> uint32_t {
        const uint32_t hw = (uint32_t)(word >> (half_ * 32));
        return ((hw >> (g * 4u)) & 0xFu) | (((hw >> (16u + g * 4u)) & 0xFu) << 4);
    };
    // the bitmap word of (tile, group): one bitmap for the batch, or one per query (lanes with the same n = the same query)
    auto mask_word = [&](uint32_t tile_, int h) __attribute__((always_inline)) -> uint64_t {
        if constexpr (!MASKED) return ~0ull;
        const uint32_t qn = (uint32_t)h * 16u + n;
        const uint64_t* mq = p.qmasks ? (qn < p.nq ? p.qmasks[qn] : nullptr) : p.mask;
        return (mq && tile_ < t1) ? mq[(uint64_t)tile_ * tstep] : ~0ull;
    };
    Factors fc[2], fn[2];  // [half parity]
    uint64_t mwc[MASKED ? NG : 1], mwn[MASKED ? NG : 1], mwe[MASKED ? NG : 1];  // bitmap words per group: the tile being multiplied, the next one, the previous one
    (void)mwc; (void)mwe;
    load_factors(t0, 0u, fn[0]);
    load_factors(t0, 1u, fn[1]);
    if constexpr (MASKED) {
#pragma unroll
        for (int h = 0; h < NG; h++) mwn[h] = mask_word(t0, h);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: the first tile's rows, the first query fragments ------------------------------------------------
    // a[idx]: k-step idx of the tile (idx = half * KSTEPS + ks), two 16-row blocks each
    u4 a[kIdx][2];
    auto issue_pair = [&](__amdgpu_buffer_rsrc_t rs, int idx0) __attribute__((always_inline)) {  // k-steps idx0, idx0 + 1 (the two halves of the rows' 128-byte lines)
        const int hf = idx0 / KSTEPS, ks0 = idx0 % KSTEPS;
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            a[idx0][rb] = load16<POLICY>(rs, voff + (uint32_t)ks0 * 64u, (uint32_t)(hf * 32 + rb * 16) * ld);
            a[idx0 + 1][rb] = load16<POLICY>(rs, voff + (uint32_t)(ks0 + 1) * 64u, (uint32_t)(hf * 32 + rb * 16) * ld);
        }
    };
    {
        const __amdgpu_buffer_rsrc_t r0 = tile_rsrc(t0, true);
#pragma unroll
        for (int i = 0; i < kIdx; i += 2) {
            issue_pair(r0, i);
            __builtin_amdgcn_sched_barrier(0);  // (in THIS order: loads return in issue order, and k-step 0 is wanted first)
        }
    }
    u4 b[R];
#pragma unroll
    for (int F = 0; F < LA; F++) b[F % R] = bq[(uint32_t)F * 64u + lane];

    // accumulators: [set = half parity][row block][query group], int32 sums of the h plane / the l plane
    v4i ach[2][2][NG], acl[2][2][NG];
    const v4i zero = {0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int h = 0; h < NG; h++) ach[1][rb][h] = acl[1][rb][h] = zero;  // (the first half's epilogue slices run on "half -1")

    // state of the epilogue in progress
    float m_h[NG];          // running maximum of the half being finished, per group (before the query's factor)
    uint32_t keyA[NG];      // first half's maximum key per group
    bool wroteA[NG];        // ... and whether its scores were written
#pragma unroll
    for (int h = 0; h < NG; h++) { m_h[h] = -__builtin_inff(); keyA[h] = kKeyMasked; wroteA[h] = false; }

#ifdef NMN_I8B_TIMING
    unsigned long long tk_sum = 0;
    const unsigned long long te_sum = 0;  // (the epilogue has no time of its own in this form: it is inside the k-loop)
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // One iteration = one tile: half 0 (with the slices of the PREVIOUS tile's half 1), half 1 (with those of this tile's half 0).
    // The range is walked one tile past its end: that last half 0 multiplies zeros (a descriptor of zero records) and carries the
    // slices of the last real half.
    for (uint32_t tile = t0; tile <= t1; tile++) {
        const bool real = tile < t1;
        // the next tile's descriptor (past the wave's range: zero records — the loads return zeros, no memory traffic)
        const __amdgpu_buffer_rsrc_t nrs = tile_rsrc(tile + 1u, tile + 1u < t1);

        // `half` multiplies into set `half`; its fragment slots carry the slices of half `1 - half` of tile `et` (set 1 - half)
        auto do_half = [&](auto half_c) __attribute__((always_inline)) {
            constexpr int half = decltype(half_c)::value;
            constexpr int ps = 1 - half;                       // the set / half-parity being finished
            const uint32_t et = half == 0 ? tile - 1u : tile;  // the tile being finished
            const bool e_on = half == 0 ? tile > t0 : true;    // (wave-uniform) false: nothing to finish yet
            const uint64_t ert = (uint64_t)et * tstep;
            const uint64_t eleft = p.n_rows - ert * kTileRows;  // rows of that tile that exist
            // this half's factors landed with its rows; the ones of the same half of the next tile go out now
            fc[half] = fn[half];
            load_factors(tile + 1u, (uint32_t)half, fn[half]);
            if constexpr (MASKED) {
                if (half == 0) {
#pragma unroll
                    for (int h = 0; h < NG; h++) {
                        mwc[h] = mwn[h];
                        mwn[h] = mask_word(tile + 1u, h);
                    }
                }
            }
            // rows of the tile being finished that take part, as this lane sees them (8 bits: [rb][e]); the bitmap word of a tile is
            // current until the next tile's half 0 has replaced it, so half 1 of tile - 1 is finished from the copy `mwe`
            uint32_t rows_h[NG];
#pragma unroll
            for (int h = 0; h < NG; h++) {
                uint64_t w = ~0ull;
                if constexpr (MASKED) w = half == 0 ? mwe[h] : mwc[h];
                if (eleft < 64) w &= (1ull << eleft) - 1ull;
                rows_h[h] = lane_rows(w, ps);
            }
            if constexpr (MASKED) {
                if (half == 0) {
#pragma unroll
                    for (int h = 0; h < NG; h++) mwe[h] = mwc[h];  // (this tile's word, for its half 1's slices in the next iteration)
                }
            }
            bool ragged = false;
            if constexpr (!MASKED) {
                // a tile that ends inside its 64 rows (the shard's last): the missing rows' factors become NaN, so their scores are NaNs
                // and v_max_f32 passes over them; the write path replaces them by the sentinel
                ragged = eleft < 64;
                if (ragged) {
#pragma unroll
                    for (int rb = 0; rb < 2; rb++)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (!((rows_h[0] >> (rb * 4 + e)) & 1u)) fc[ps].rf[rb][e] = __builtin_nanf("");
                }
            }
            (void)ragged;
            const __amdgpu_buffer_rsrc_t sc_rs = make_rsrc(p.scores + ert * p.nql * 64ull, p.nql * 256u);

            // score of element (rb, e) of group H of the half being finished, without / with the query's factor
            auto raw = [&](int H, int rb, int e) __attribute__((always_inline)) -> float {
                // h.c + (l.c) / 256: exact integers well below 2^24 * 256, one rounding of 2^-24 relative each
#ifdef NMN_I8B_HALF_VALU  // measurement only: one conversion instead of two and a multiply-add (answers are wrong)
                return (float)ach[ps][rb][H][e];
#endif
                return __builtin_fmaf((float)acl[ps][rb][H][e], 0.00390625f, (float)ach[ps][rb][H][e]);
            };
            auto pre = [&](int H, int rb, int e) __attribute__((always_inline)) -> float {  // what the running maximum takes
                const float x = raw(H, rb, e);
                if constexpr (kScaled) return x * fc[ps].rf[rb][e];  // (the query's factor, >= 0, is applied to the maximum)
                else return l2_score_q<METRIC == kMetricNegL2>(qq_h[kL2 ? H : 0], fc[ps].rv[kL2 ? rb : 0][e], x * (invq_h[H] * fc[ps].rf[rb][e]));
            };
            auto word = [&](int H, int rb, int e) __attribute__((always_inline)) -> uint32_t {  // the score as written
                const float sc = kScaled ? pre(H, rb, e) * invq_h[H] : pre(H, rb, e);
                return ((rows_h[H] >> (rb * 4 + e)) & 1u) ? f2u(sc) : kScoreSentinelBits;
            };
            // slice v (0..15): two elements of one (group, row block) into the group's running maximum
            auto value_slice = [&](int v) __attribute__((always_inline)) {
                const int H = v >> 2, rb = (v >> 1) & 1, e0 = (v & 1) * 2;
                float s0 = pre(H, rb, e0), s1 = pre(H, rb, e0 + 1);
                if constexpr (MASKED) {
                    if (!((rows_h[H] >> (rb * 4 + e0)) & 1u)) s0 = -__builtin_inff();
                    if (!((rows_h[H] >> (rb * 4 + e0 + 1)) & 1u)) s1 = -__builtin_inff();
                }
                m_h[H] = __builtin_fmaxf(m_h[H], __builtin_fmaxf(s0, s1));  // v_max_f32 skips NaNs; an all-NaN lane reports -inf, an upper bound of its key
                asm volatile("" : "+v"(m_h[H]));  // (pins the slice HERE: left alone the compiler sinks all sixteen down to the group's finish)
            };
            // slice "finish group H": the half's key, and after the second half the tile's key, maxima and score writes
            auto finish_slice = [&](int H) __attribute__((always_inline)) {
                const float mm = kScaled ? m_h[H] * invq_h[H] : m_h[H];  // (>= 0: monotone, rounding included)
                m_h[H] = -__builtin_inff();
                uint32_t hkey = rows_h[H] ? score_to_key(mm) : kKeyMasked;
                // the maximum over the four lane groups of a query (v_permlane32_swap / v_permlane16_swap: no LDS round trip)
                {
                    const auto r32 = __builtin_amdgcn_permlane32_swap(hkey, hkey, false, false);
                    hkey = max((uint32_t)r32[0], (uint32_t)r32[1]);
                    const auto r16 = __builtin_amdgcn_permlane16_swap(hkey, hkey, false, false);
                    hkey = max((uint32_t)r16[0], (uint32_t)r16[1]);
                }
                if (!e_on) return;
                uint32_t n_v = n;
                asm volatile("" : "+v"(n_v));  // (opaque here: what is derived from it — query numbers, store addresses — is not kept across the loop)
                const uint32_t qn = (uint32_t)H * 16u + n_v;
                const bool q_ok = qn < p.nq;
                const uint32_t skip = skip_h[H];
#ifdef NMN_I8B_NO_SCORE_WRITES
                const bool may_write = false;
#else
                const bool may_write = q_ok && !sampling;
#endif
                auto store_half = [&](int hf, bool sentinel) __attribute__((always_inline)) {
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) {
                        u4 w = {kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits, kScoreSentinelBits};
                        if (!sentinel) {
#pragma unroll
                            for (int e = 0; e < 4; e++) w[e] = word(H, rb, e);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(w, sc_rs, qn * 256u + g * 16u + (uint32_t)(hf * 32 + rb * 16) * 4u, 0, 0);
                    }
                };
                if (ps == 0) {
                    // Scores are only worth their HBM write when the tile can still hold a candidate (skip_key: the sampled bound).
                    // The first half does not know the tile's maximum yet: it writes if ITS maximum qualifies; if only the second half
                    // qualifies the first half's rows (all below skip_key: no candidates) are written as sentinels then.
                    keyA[H] = hkey;
                    wroteA[H] = may_write && hkey != kKeyMasked && hkey >= skip;
                    if (wroteA[H]) store_half(0, false);
                } else {
                    const uint32_t tkey = max(keyA[H], hkey);
                    // tile maxima leave in groups of four tiles (one 16-byte store per query); ragged ends one by one
                    {
                        const uint32_t slot = et & 3u;  // (wave-uniform)
                        uint32_t* mine = tk_pend + ((wave * 4u + (uint32_t)H) * 16u + n_v) * 4u;
                        if (g == 0) mine[slot] = tkey;
                        if (slot == 3u || et + 1u == t1) {
                            const uint32_t g0 = et & ~3u, first = max(g0, t0);
                            if (q_ok && g == 0) {
                                uint32_t* dst = p.tmax + (uint64_t)qn * p.tmax_stride + g0;
                                const u4 v = *reinterpret_cast<const u4*>(mine);
                                if (first == g0 && slot == 3u && (p.tmax_stride & 3ull) == 0ull) {
                                    *reinterpret_cast<u4*>(dst) = v;
                                } else {
                                    if (first <= g0 + 0u) dst[0] = v[0];
                                    if (first <= g0 + 1u && slot >= 1u) dst[1] = v[1];
                                    if (first <= g0 + 2u && slot >= 2u) dst[2] = v[2];
                                    if (first <= g0 + 3u && slot >= 3u) dst[3] = v[3];
                                }
                            }
                        }
                    }
                    wmax_h[H] = max(wmax_h[H], tkey);
                    if (may_write && tkey != kKeyMasked && tkey >= skip) {
                        store_half(1, false);
                        if (!wroteA[H]) store_half(0, true);
                    }
                }
            };

            const __amdgpu_buffer_rsrc_t lrs = nrs;
            __builtin_amdgcn_sched_barrier(0);
#ifdef NMN_I8B_TIMING
            const unsigned long long t_a = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                const int idx = half * KSTEPS + ks;
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const int F = ks * NF + f;  // fragment slot of this half: 0 .. kFragsRow - 1
                    b[(F + LA) % R] = bq[(uint32_t)((F + LA) % kFragsRow) * 64u + lane];
                    const v4i bv = __builtin_bit_cast(v4i, b[F % R]);
#pragma unroll
                    for (int rb = 0; rb < 2; rb++) {
                        const v4i av = __builtin_bit_cast(v4i, a[idx][rb]);
#ifdef NMN_I8B_NO_MFMA  // measurement only: the loads without the products (answers are wrong)
                        if (ks == 0) { ach[half][rb][f >> 1] = zero; acl[half][rb][f >> 1] = zero; }
                        if (f == 0) ach[half][rb][0] += av + bv;
#else
                        // (inline asm: the accumulators are pinned to architectural VGPRs — the slices read them with VALU instructions, and
                        //  out of AGPRs every element would cost a v_accvgpr_read first, 256 per tile, which the register allocator also
                        //  gathers at the top of the half, outside the matrix-core stream — and the rows to AGPRs, where the loads put
                        //  them.  The compiler does not know these are MFMAs: a set is read by the slices >= 8 MFMAs after its last
                        //  write and re-used as an accumulator every 8th MFMA, beyond every wait state the ISA asks for.)
                        v4i& acc = (f & 1) == 0 ? ach[half][rb][f >> 1] : acl[half][rb][f >> 1];
                        if (ks == 0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=v"(acc) : "a"(av), "v"(bv));
                        else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "a"(av), "v"(bv));
#endif
                    }
#ifndef NMN_I8B_NO_EPILOGUE  // (measurement only: the sweep without its epilogue — answers are wrong)
                    // the slices of the half being finished: 4 NG value slices in slots 4, 8, .., 16 NG, the groups' finish in 16 NG + 4, + 8, ..
                    if (F >= 4 && F <= 16 * NG && F % 4 == 0) value_slice(F / 4 - 1);
                    if (F >= 16 * NG + 4 && F <= 16 * NG + 4 * NG && (F - 16 * NG) % 4 == 0) finish_slice((F - 16 * NG) / 4 - 1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifndef NMN_I8B_NO_LOADS  // (measurement only: the products without the loads — every tile multiplies the first one's rows)
                // the same k-steps of the next tile into the registers just consumed — by PAIRS: the two 64-byte halves of a row's
                // 128-byte line are asked for back to back
                if (idx & 1) {
                    issue_pair(lrs, idx - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
#ifdef NMN_I8B_TIMING
            tk_sum += __builtin_amdgcn_s_memtime() - t_a;
#endif
        };
        do_half(std::integral_constant<int, 0>{});
        if (!real) break;
        do_half(std::integral_constant<int, 1>{});
    }
#ifdef NMN_I8B_TIMING
    if (!sampling && lane == 0) {
        nmn_i8b_dbg[sw * 8 + 0] = tk_sum;
        nmn_i8b_dbg[sw * 8 + 1] = te_sum;
        nmn_i8b_dbg[sw * 8 + 2] = __builtin_amdgcn_s_memtime() - t_begin;
        nmn_i8b_dbg[sw * 8 + 3] = t1 - t0;
        nmn_i8b_dbg[sw * 8 + 4] = t_entry;
        nmn_i8b_dbg[sw * 8 + 5] = t_begin;
        nmn_i8b_dbg[sw * 8 + 6] = __builtin_amdgcn_s_memtime();
        nmn_i8b_dbg[sw * 8 + 7] = (unsigned long long)blockIdx.x << 32;
    }
#endif
    if (sampling) return;  // the sampling pass leaves only tmax
#pragma unroll
    for (int h = 0; h < NG; h++) {
        const uint32_t qn = (uint32_t)h * 16u + n;
        if (qn < p.nq && g == 0) p.wmax[(size_t)qn * p.wmax_stride + sw] = wmax_h[h];
    }
}

template <int KSTEPS, int METRIC, bool MASKED, int NG>
hipError_t launch_one(const ScanParams& p, hipStream_t s) {
    const uint32_t waves_all = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    if (p.bx_base >= waves_all) return hipSuccess;
    const uint32_t waves = p.bx_count ? std::min(p.bx_count, waves_all - p.bx_base) : waves_all - p.bx_base;
    ScanParams pf = p;
    pf.bx_count = waves;
    const size_t lds = (size_t)KSTEPS * 2 * NG * 64 * 16 + (size_t)kI8bWaves * 4 * 16 * 16;  // the queries + the pending tile maxima
    // cache policy of the row loads: a 128-byte line is read as two 64-byte halves by two instructions, so the line must survive in
    // the vector cache between them — non-temporal loads (what every other sweep uses) re-fetch it
    // (tools/micro/read_bw.hip "fragment": 5.4 vs 6.4 TB/s).  NMN_I8B_NT=1: the A/B.
    static const bool nt = getenv("NMN_I8B_NT") != nullptr;
    auto kern = nt ? scan_i8b_kernel<KSTEPS, METRIC, MASKED, 1, NG> : scan_i8b_kernel<KSTEPS, METRIC, MASKED, 0, NG>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((waves + (uint32_t)kI8bWaves - 1u) / (uint32_t)kI8bWaves), dim3(kI8bWaves * 64), lds, s, pf);
    return hipGetLastError();
}

template <int METRIC, bool MASKED>
hipError_t launch_groups(const ScanParams& p, hipStream_t s) {
    if (p.ld != 768u) return hipErrorInvalidValue;
    switch ((p.nq + 15u) / 16u) {
        case 1: return launch_one<12, METRIC, MASKED, 1>(p, s);
        case 2: return launch_one<12, METRIC, MASKED, 2>(p, s);
        case 3: return launch_one<12, METRIC, MASKED, 3>(p, s);
        case 4: return launch_one<12, METRIC, MASKED, 4>(p, s);
        default: return hipErrorInvalidValue;
    }
}

template <int METRIC>
hipError_t launch_metric(const ScanParams& p, hipStream_t s) {
    return (p.mask || p.qmasks) ? launch_groups<METRIC, true>(p, s) : launch_groups<METRIC, false>(p, s);
}

}  // namespace

// rows of 768 elements, up to 64 queries per pass (the queries' two planes take ld * 128 bytes of LDS)
bool scan_i8b_supported(uint32_t ld, uint32_t dim, int metric, uint32_t nq) {
    if (!(metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2))
        return false;
    // Opt-in (NMN_I8B=1): parity green on the whole batched suite, but on 10M x 768, 64 queries it measures 1.65-1.73 ms per
    // sweep against the LDS-ring kernel's 1.46-1.50 (docs/kernel-scan-i8b.md: where the time goes).  The ring kernel serves by default.
    static const bool on = getenv("NMN_I8B") != nullptr;
    return on && dim <= ld && ld == 768u && nq <= 64u;
}

// p.tiles_per_wave = tiles per WAVE (a scan wave of the selection), p.bx_base / bx_count in waves
hipError_t launch_scan_i8b(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_metric<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN: return launch_metric<NMN_METRIC_EUCLIDEAN>(p, s);
        case kMetricNegL2: return launch_metric<kMetricNegL2>(p, s);
        default: return launch_metric<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

}  // namespace nmn
