// nmn_scan_ring.hip — ONE query over the row-major f32 corpus, f32 arithmetic, the rows streamed through the LDS-DMA ring.
//
// The headline sweep of the path (rows * dim * 4 bytes per query; vector_engine/src/lib.rs:2115-2228 is the loop it
// replaces).  nmn_scan.hip's scan_kernel does the same job with register loads (16 lanes per row, 12 x 16 bytes per lane in
// flight) and stands at 0.81-0.82 of the 8 TB/s HBM peak; round 5 measured the matrix-core sweep over the SAME f32 rows
// (nmn_scan_mfma_f32.hip) at 0.85 with three queries (profiles/r05z7_*): what streams faster there is not the matrix core but the
// data movement — global_load_lds pieces of 1 KiB into a ring of four 32-KiB stages, three always in flight per CU, no VGPRs, no
// address arithmetic per load in the loop.  This kernel keeps that movement and does the arithmetic the headline must do in f32:
//   * workgroup = 4 waves = one 64-row tile at a time, a contiguous range of tiles per workgroup; a stage is [64 rows][128 f32];
//     wave w owns rows 16 w .. 16 w + 15 of every tile and takes them four at a time, SIXTEEN LANES PER ROW (scan_kernel's shape):
//     lane (r4 = lane >> 4, j = lane & 15) reads the eight f32 at 8 j of row 16 w + 4 sub + r4 of the stage (two ds_read_b128 at the
//     swizzled chunks 2 j, 2 j + 1; a quarter-wave reads one row's 512 contiguous bytes: no bank conflicts) and multiplies them
//     into two accumulators per sub-step against the query's eight values for that column slice — the query sits in LDS behind
//     the ring (two more 16-byte reads per stage and lane, the same 512 bytes for all four quarter-waves), so the stage loop is a
//     plain loop over the row's ld / 128 stages: one kernel per metric for every row length, ~90 VGPRs.  (Query slices in
//     registers with the stage loop unrolled measured the same at 768 elements and spilled from 1536 on.)
//   * per tile: the sixteen lanes of a row meet (four DPP rotations), the row's score is formed as scan_kernel forms it (the same
//     expressions: the candidate margins of qprep_kernel's plain-f32 case apply unchanged), 16 scores per wave are written, the
//     tile maximum meets through LDS behind the next stage's barrier, the workgroup maximum at the end — the three-level
//     hierarchy select_kernel reads, with `tiles_per_wave` = tiles per WORKGROUP as on the matrix-core path.
// Approximate scores only (any summation order, FMA): exactness is restored by the rescore in the reference's order, as always.
// Unmasked single queries on shards of >= 4096 tiles and row strides of whole 128-element stages up to 1536; everything else
// stays on scan_kernel (bitmaps — it reads only the kept rows —, two queries, short shards, the f32 retry, f64 artifact scores).
#include <algorithm>
#include <cstdlib>

#include "nmn_internal.h"

namespace nmn {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int kRingStageBytes = 64 * 128 * 4;  // [64 rows][128 f32] = 32 KiB
constexpr int kRingStages = 4;                 // 128 KiB, three stages in flight
constexpr int kRingPieces = 8;                 // 1-KiB LDS-DMA instructions per wave and stage (2 rows x 512 B each)
constexpr int kRingRowPitch = 128;             // floats between the rows of a stage

template <int N>
__device__ __forceinline__ void ring_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// METRIC: NMN_METRIC_COSINE | NMN_METRIC_DOT_PRODUCT (dot products) or NMN_METRIC_EUCLIDEAN (sum of squared differences; kMetricNegL2
// picks -d over 1 / (1 + d) in the epilogue).  KC = ld / 128 stages per row (runtime).
template <int METRIC>
__global__ void __launch_bounds__(256, 1) scan_ring_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // ring | [8 tiles][64] row magnitudes | [2][4] tile-maximum parts | query [ld]
    float* const nrm = lds + kRingStages * (kRingStageBytes / 4);
    uint32_t* const tpart = reinterpret_cast<uint32_t*>(nrm + 8 * 64);  // (8 slots: with one stage per row the ring runs 4 tiles ahead of the epilogue)
    float* const qlds = nrm + 8 * 64 + 8;
    const uint32_t KC = p.ld / 128u;
    constexpr bool kL2 = METRIC == NMN_METRIC_EUCLIDEAN;
    constexpr bool kCos = METRIC == NMN_METRIC_COSINE;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r4 = lane >> 4, j = lane & 15u;
    const uint32_t ld = p.ld;
    const uint32_t row_bytes = ld * 4u;
    const uint32_t bx = blockIdx.x;
    const uint32_t t0 = bx * p.tiles_per_wave;  // tiles per WORKGROUP on this path
    if (t0 >= p.n_tiles) return;
    const uint32_t t1 = min(t0 + p.tiles_per_wave, p.n_tiles);
    const uint32_t n_stage = (t1 - t0) * KC;

    // ---- the query into LDS (read back per stage: elements 128 kc + 8 j .. + 7 for this lane), by LDS-DMA as well: 1 KiB per
    // instruction, wave w takes the KiBs w, w + 4, ...  No register round trip in front of the ring's first pieces — a workgroup's
    // start is one memory latency, not two, sixteen times per CU and sweep — and these are the OLDEST entries of the wave's in-order
    // queue: the first stage's counted wait covers them, its barrier makes them visible.
    for (uint32_t c = wave; c * 256u < ld; c += 4u) {
        if (c * 256u + lane * 4u < ld)  // (row strides are multiples of 128 floats: the last KiB may be half)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.qpad + c * 256u + lane * 4u),
                                             (__attribute__((address_space(3))) void*)(qlds + c * 256u), 16, 0, 0);
    }
    const float qmag = p.qinfo[0].qmag;

    // DMA source offsets of this wave's pieces: piece pp = rows 2 (8 wave + pp) + lane / 32, LDS chunk lane % 32, source chunk
    // (lane % 32) ^ (row & 15) (the swizzle lives on the source side: the LDS side of an LDS-DMA is wave base + lane * 16)
    uint32_t loff[kRingPieces];
#pragma unroll
    for (int pp = 0; pp < kRingPieces; pp++) {
        const uint32_t r = 2u * (wave * kRingPieces + (uint32_t)pp) + lane / 32u;
        loff[pp] = r * row_bytes + (((lane % 32u) ^ (r & 15u)) * 16u);
    }
    const char* const mat = reinterpret_cast<const char*>(p.corpus);
    auto stage_src = [&](uint32_t tile_, uint32_t kc_) -> const char* {
        return mat + (uint64_t)tile_ * kTileRows * row_bytes + (uint64_t)kc_ * 512u;
    };
    auto issue_stage = [&](const char* src, uint32_t lmask, uint32_t slot) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < kRingPieces; pp++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (loff[pp] & lmask)),
                                             (__attribute__((address_space(3))) void*)(lds + slot * (kRingStageBytes / 4) +
                                                                                        (wave * kRingPieces + (uint32_t)pp) * 256u),
                                             16, 0, 2);  // non-temporal: the rows are read once
    };
    auto norms_dma = [&](uint32_t tile_, uint32_t rel) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.norms + (uint64_t)tile_ * kTileRows + lane),
                                         (__attribute__((address_space(3))) void*)(nrm + (rel & 7u) * 64u), 4, 0, 0);
    };
    // prologue: kRing - 1 stages in flight (dummy pieces where the range is shorter: the counted waits below count on them) — and the
    // row magnitudes of the tile of stage kRing - 1 as well: its pieces go out in the first iteration, which asks for the magnitudes
    // of the stage AFTER it only (rows of <= 384 elements start a new tile there)
#pragma unroll
    for (uint32_t s0 = 0; s0 < kRingStages; s0++) {
        if (s0 < n_stage) {
            if (kCos && wave == 0 && s0 % KC == 0) norms_dma(t0 + s0 / KC, s0 / KC);
            if (s0 < kRingStages - 1) issue_stage(stage_src(t0 + s0 / KC, s0 % KC), 0xFFFFFFFFu, s0 % kRingStages);
        } else if (s0 < kRingStages - 1) {
            issue_stage(mat, 0u, s0 % kRingStages);
        }
    }
    // LDS read offsets (floats) of the four sub-steps: row 16 wave + 4 sub + r4, chunk (2 j) ^ (row & 15) and its partner (^ 4 floats)
    constexpr int kSub = 4;
    uint32_t off[kSub];
#pragma unroll
    for (int sub = 0; sub < kSub; sub++) {
        const uint32_t rr = (uint32_t)sub * 4u + r4;  // row within the wave's sixteen (= row & 15 of the tile row 16 wave + rr)
        off[sub] = (wave * 16u + rr) * kRingRowPitch + (((j * 2u) ^ rr) * 4u);
    }

    uint32_t wmax = kKeyMasked;   // (wave 0: over the finished tiles of the workgroup)
    uint32_t sidx = 0;
    // the stage the loop issues next (stage index sidx + kRingStages - 1), advanced incrementally: a division per stage costs more
    // scalar instructions than the stage's arithmetic
    uint32_t nt = t0 + (kRingStages - 1) / KC, nkc = (kRingStages - 1) % KC;
    for (uint32_t tile = t0; tile < t1; tile++) {
        float acc[kSub][2];
#pragma unroll
        for (int sub = 0; sub < kSub; sub++) acc[sub][0] = acc[sub][1] = 0.f;
        for (uint32_t kc = 0; kc < KC; kc++, sidx++) {
            const float* buf = lds + (sidx % kRingStages) * (kRingStageBytes / 4);
            ring_wait_vm<(kRingStages - 2) * kRingPieces>();  // stage sidx has landed (pieces are issued for every stage, real or dummy)
            // A wave reads only the rows its OWN pieces brought (rows 16 w .. 16 w + 15 of every stage): its counted wait is all the
            // hand-over a stage needs.  The workgroup meets once per TILE — for the query (first tile) and the tile maxima's parts.
            if (kc == 0) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kc == 0 && wave == 0 && tile > t0) {
                // the previous tile's maximum: its four parts were written before this barrier
                const uint32_t* tp = tpart + ((tile - 1u - t0) & 1u) * 4u;
                const uint32_t m = max(max(tp[0], tp[1]), max(tp[2], tp[3]));
                if (lane == 0) p.tmax[tile - 1u] = m;
                wmax = max(wmax, m);
            }
            const uint32_t ns = sidx + (kRingStages - 1);
            const bool issue = ns < n_stage;
            const char* const nsrc = issue ? stage_src(nt, nkc) : mat;
            const uint32_t lmask = issue ? 0xFFFFFFFFu : 0u;
            float* const nbuf = lds + (ns % kRingStages) * (kRingStageBytes / 4);
            f4 a[kSub][2];
#pragma unroll
            for (int sub = 0; sub < kSub; sub++) {
                a[sub][0] = *reinterpret_cast<const f4*>(buf + off[sub]);
                a[sub][1] = *reinterpret_cast<const f4*>(buf + (off[sub] ^ 4u));
            }
            const f4 q0 = *reinterpret_cast<const f4*>(qlds + kc * 128u + j * 8u), q1 = *reinterpret_cast<const f4*>(qlds + kc * 128u + j * 8u + 4u);
#pragma unroll
            for (int sub = 0; sub < kSub; sub++) {
                // two pieces of the stage ahead per sub-step, in the shadow of the arithmetic
#pragma unroll
                for (int pp = sub * 2; pp < sub * 2 + 2; pp++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc + (loff[pp] & lmask)),
                                                     (__attribute__((address_space(3))) void*)(nbuf + (wave * kRingPieces + (uint32_t)pp) * 256u), 16, 0, 2);
                const f4 x0 = a[sub][0], x1 = a[sub][1];
                if constexpr (kL2) {
                    const f4 d0 = x0 - q0, d1 = x1 - q1;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        acc[sub][0] = __builtin_fmaf(d0[e], d0[e], acc[sub][0]);
                        acc[sub][1] = __builtin_fmaf(d1[e], d1[e], acc[sub][1]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        acc[sub][0] = __builtin_fmaf(x0[e], q0[e], acc[sub][0]);
                        acc[sub][1] = __builtin_fmaf(x1[e], q1[e], acc[sub][1]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // row magnitudes of the tile whose first stage goes out in the next iteration
            if (++nkc == KC) {
                nkc = 0;
                nt++;
            }
            if (kCos && wave == 0 && ns + 1u < n_stage && nkc == 0) norms_dma(nt, nt - t0);
        }
        // ---- the tile's 16 rows of this wave: the sixteen lanes of a row meet (row_ror 8, 4, 2, 1: every lane of the DPP row holds the
        // sum), then lane (r4, j) finishes row 4 (j & 3) + r4 of the wave's sixteen (four lanes per row: the write below takes j < 4)
        float v = 0.f;
#pragma unroll
        for (int sub = 0; sub < kSub; sub++) {
            float t = acc[sub][0] + acc[sub][1];
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xF, 0xF, false));
            if ((j & 3u) == (uint32_t)sub) v = t;
        }
        const uint32_t wrow = (j & 3u) * 4u + r4;  // this lane's row among the wave's sixteen
        const uint64_t row = (uint64_t)tile * kTileRows + wave * 16u + wrow;
        const bool valid = row < p.n_rows;
        float sc;
        if constexpr (kCos) {
            const float vn = nrm[((tile - t0) & 7u) * 64u + wave * 16u + wrow];
            sc = (vn == 0.f || qmag == 0.f) ? 0.f : v / (qmag * vn);
        } else if constexpr (kL2) {
            const float dist = sqrtf(fmaxf(v, 0.f));
            sc = p.metric == kMetricNegL2 ? -dist : 1.0f / (1.0f + dist);
        } else {
            sc = v;
        }
        uint32_t key = valid ? score_to_key(sc) : kKeyMasked;
        if (j < 4u) p.scores[row] = valid ? f2u(sc) : kScoreSentinelBits;  // (nql == 1: score_at(row, 0, 1) == row)
        // the wave's maximum (every row's key is held by four lanes)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o));
        if (lane == 0) tpart[((tile - t0) & 1u) * 4u + wave] = key;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    ring_wait_vm<0>();  // the dummy pieces of the tail have landed before this workgroup's LDS is handed on
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave == 0) {
        const uint32_t* tp = tpart + ((t1 - 1u - t0) & 1u) * 4u;
        const uint32_t m = max(max(tp[0], tp[1]), max(tp[2], tp[3]));
        wmax = max(wmax, m);
        if (lane == 0) {
            p.tmax[t1 - 1u] = m;
            p.wmax[bx] = wmax;
        }
    }
}

// ---- the same sweep over the shard's 8-BIT mirror --------------------------------------------------------------------------------
// nmn_scan_i8.hip's single-query sweep (one byte per element, v_dot4_i32_i8 against the query's h / l planes, the margin carries the
// measured quantisation error) with the rows through the ring instead of 16-byte register loads.  A stage is [64 rows][256 B]
// (16 KiB), eight of them in the ring, seven in flight; a wave's four 1-KiB pieces of a stage are ITS OWN sixteen rows (4 rows x
// 256 B each), so a stage needs no workgroup barrier: the wave's counted wait hands it over.  Sixteen lanes per row as everywhere:
// lane j holds the 16 bytes at 16 j of the row's 256-byte segment — chunk j + 16 kc of the row, the chunk assignment of
// scan_i8_kernel, so the per-lane integer sums, their conversion and the row's sixteen-lane sum are that kernel's to the bit —
// and reads the query's two planes for that chunk from LDS behind the ring.  Per tile: scale / magnitude / |v~|^2 of the 64 rows
// come by LDS-DMA as well (wave 0: scales, wave 1: magnitudes or |v~|^2, wave 2: magnitudes for the second Euclidean estimator),
// the score is formed with scan_i8_kernel's expressions, tile maxima meet through LDS once per tile.
constexpr int kI8StageBytes = 64 * 256;  // 16 KiB
constexpr int kI8Stages = 8;             // 128 KiB
constexpr int kI8Pieces = 4;             // 1-KiB LDS-DMA instructions per wave and stage
constexpr int kI8FacSlots = 16;          // tiles whose per-row factors live in LDS at once (rows of 256 elements: the ring runs 8 tiles ahead)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int METRIC>
__global__ void __launch_bounds__(256, 1) scan_ring_i8_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // ring | [16 tiles][3][64] factors | [2][4] tile-maximum parts | query planes [2][ld] int8
    float* const fac = lds + kI8Stages * (kI8StageBytes / 4);
    uint32_t* const tpart = reinterpret_cast<uint32_t*>(fac + kI8FacSlots * 3 * 64);
    float* const qlds = fac + kI8FacSlots * 3 * 64 + 8;
    const uint32_t ld = p.ld, KC = ld / 256u;
    constexpr bool kL2 = METRIC == NMN_METRIC_EUCLIDEAN;
    constexpr bool kCos = METRIC == NMN_METRIC_COSINE;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r4 = lane >> 4, j = lane & 15u;
    const uint32_t bx = blockIdx.x;
    const uint32_t t0 = bx * p.tiles_per_wave;  // tiles per WORKGROUP on this path
    if (t0 >= p.n_tiles) return;
    const uint32_t t1 = min(t0 + p.tiles_per_wave, p.n_tiles);
    const uint32_t n_stage = (t1 - t0) * KC;

    // the query's planes (h: bytes [0, ld), l: [ld, 2 ld)) by LDS-DMA, the oldest entries of every wave's queue
    for (uint32_t c = wave; c * 1024u < 2u * ld; c += 4u) {
        if (c * 1024u + lane * 16u < 2u * ld)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(p.qi8) + c * 1024u + lane * 16u),
                                             (__attribute__((address_space(3))) void*)(qlds + c * 256u), 16, 0, 0);
    }
    const float qmag = p.qinfo[0].qmag, qsc = p.qinfo[0].qscale, qq8 = p.qinfo[0].qq8;
    const bool neg = p.metric == kMetricNegL2;
    const bool est_b = kL2 && qq8 < 0.f;  // (qprep_kernel: which Euclidean estimator this query takes)

    uint32_t loff[kI8Pieces];
#pragma unroll
    for (int pp = 0; pp < kI8Pieces; pp++) loff[pp] = (wave * 16u + 4u * (uint32_t)pp + lane / 16u) * ld + (lane % 16u) * 16u;
    const char* const mat = reinterpret_cast<const char*>(p.corpus_i8);
    auto stage_src = [&](uint32_t tile_, uint32_t kc_) -> const char* { return mat + (uint64_t)tile_ * kTileRows * ld + (uint64_t)kc_ * 256u; };
    auto issue_stage = [&](const char* src, uint32_t lmask, uint32_t slot) __attribute__((always_inline)) {
#pragma unroll
        for (int pp = 0; pp < kI8Pieces; pp++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (loff[pp] & lmask)),
                                             (__attribute__((address_space(3))) void*)(lds + slot * (kI8StageBytes / 4) + (wave * kI8Pieces + (uint32_t)pp) * 256u),
                                             16, 0, 2);
    };
    // per-row factors of a tile: wave 0 the scales, wave 1 magnitudes (cosine) / |v~|^2 (Euclidean), wave 2 magnitudes (estimator B)
    auto factors_dma = [&](uint32_t tile_, uint32_t rel) __attribute__((always_inline)) {
        const float* src = wave == 0 ? p.i8_scale : wave == 1 ? (kCos ? p.norms : p.i8_vv) : p.norms;
        const bool mine = wave == 0 || (wave == 1 && (kCos || kL2)) || (wave == 2 && est_b);
        if (mine)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint64_t)tile_ * kTileRows + lane),
                                             (__attribute__((address_space(3))) void*)(fac + ((rel % kI8FacSlots) * 3u + wave) * 64u), 4, 0, 0);
    };
    // prologue: kI8Stages - 1 stages in flight, and the factors of every tile that has a stage among them or right behind them
#pragma unroll
    for (uint32_t s0 = 0; s0 < kI8Stages; s0++) {
        if (s0 < n_stage) {
            if (s0 % KC == 0) factors_dma(t0 + s0 / KC, s0 / KC);
            if (s0 < kI8Stages - 1) issue_stage(stage_src(t0 + s0 / KC, s0 % KC), 0xFFFFFFFFu, s0);
        } else if (s0 < kI8Stages - 1) {
            issue_stage(mat, 0u, s0);
        }
    }
    constexpr int kSub = 4;
    uint32_t off[kSub];  // (floats) row 16 wave + 4 sub + r4 of the stage, chunk j
#pragma unroll
    for (int sub = 0; sub < kSub; sub++) off[sub] = (wave * 16u + (uint32_t)sub * 4u + r4) * 64u + j * 4u;

    uint32_t wmax = kKeyMasked;
    uint32_t sidx = 0;
    uint32_t nt = t0 + (kI8Stages - 1) / KC, nkc = (kI8Stages - 1) % KC;
    for (uint32_t tile = t0; tile < t1; tile++) {
        int hi[kSub], lo[kSub];
#pragma unroll
        for (int sub = 0; sub < kSub; sub++) hi[sub] = lo[sub] = 0;
        for (uint32_t kc = 0; kc < KC; kc++, sidx++) {
            const float* buf = lds + (sidx % kI8Stages) * (kI8StageBytes / 4);
            ring_wait_vm<(kI8Stages - 2) * kI8Pieces>();  // this wave's pieces of stage sidx have landed
            if (kc == 0) {
                __builtin_amdgcn_s_barrier();  // once per tile: the query (first tile), the factors other waves fetched, the maxima's parts
                asm volatile("" ::: "memory");
                if (wave == 0 && tile > t0) {
                    const uint32_t* tp = tpart + ((tile - 1u - t0) & 1u) * 4u;
                    const uint32_t m = max(max(tp[0], tp[1]), max(tp[2], tp[3]));
                    if (lane == 0) p.tmax[tile - 1u] = m;
                    wmax = max(wmax, m);
                }
            }
            const uint32_t ns = sidx + (kI8Stages - 1);
            const bool issue = ns < n_stage;
            const char* const nsrc = issue ? stage_src(nt, nkc) : mat;
            const uint32_t lmask = issue ? 0xFFFFFFFFu : 0u;
            float* const nbuf = lds + (ns % kI8Stages) * (kI8StageBytes / 4);
            v4i x[kSub];
#pragma unroll
            for (int sub = 0; sub < kSub; sub++) x[sub] = *reinterpret_cast<const v4i*>(buf + off[sub]);
            const v4i qh = *reinterpret_cast<const v4i*>(qlds + (kc * 16u + j) * 4u), ql = *reinterpret_cast<const v4i*>(qlds + (ld >> 2) + (kc * 16u + j) * 4u);
#pragma unroll
            for (int sub = 0; sub < kSub; sub++) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc + (loff[sub] & lmask)),
                                                 (__attribute__((address_space(3))) void*)(nbuf + (wave * kI8Pieces + (uint32_t)sub) * 256u), 16, 0, 2);
                hi[sub] = __builtin_amdgcn_sdot4(x[sub].x, qh.x, hi[sub], false);
                lo[sub] = __builtin_amdgcn_sdot4(x[sub].x, ql.x, lo[sub], false);
                hi[sub] = __builtin_amdgcn_sdot4(x[sub].y, qh.y, hi[sub], false);
                lo[sub] = __builtin_amdgcn_sdot4(x[sub].y, ql.y, lo[sub], false);
                hi[sub] = __builtin_amdgcn_sdot4(x[sub].z, qh.z, hi[sub], false);
                lo[sub] = __builtin_amdgcn_sdot4(x[sub].z, ql.z, lo[sub], false);
                hi[sub] = __builtin_amdgcn_sdot4(x[sub].w, qh.w, hi[sub], false);
                lo[sub] = __builtin_amdgcn_sdot4(x[sub].w, ql.w, lo[sub], false);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (++nkc == KC) {
                nkc = 0;
                nt++;
            }
            if (ns + 1u < n_stage && nkc == 0) factors_dma(nt, nt - t0);  // the tile whose first stage goes out in the next iteration
        }
        // the wave's sixteen rows: h.c + (l.c) / 256 per lane (exact conversions, one rounding), the sixteen lanes of a row meet, lane
        // (r4, j) finishes row 4 (j & 3) + r4
        float v = 0.f;
#pragma unroll
        for (int sub = 0; sub < kSub; sub++) {
            float t = (float)hi[sub] + (float)lo[sub] * 0.00390625f;
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xF, 0xF, false));
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xF, 0xF, false));
            if ((j & 3u) == (uint32_t)sub) v = t;
        }
        const uint32_t wrow = wave * 16u + (j & 3u) * 4u + r4;  // this lane's row of the tile
        const uint64_t row = (uint64_t)tile * kTileRows + wrow;
        const bool valid = row < p.n_rows;
        const float* fs = fac + ((tile - t0) % kI8FacSlots) * 3u * 64u;
        const float sr = valid ? fs[wrow] : 0.f;
        float vn = 1.f, vb = 0.f;
        if constexpr (kCos) vn = valid ? fs[64u + wrow] : 1.f;
        if constexpr (kL2) {
            vn = valid ? fs[64u + wrow] : 0.f;
            if (est_b) vb = fs[128u + wrow];
        }
        const float dot = v * (qsc * sr);
        float sc;
        if constexpr (kCos) sc = (vn == 0.f || qmag == 0.f) ? 0.f : dot / (qmag * vn);
        else if constexpr (kL2) {
            const float qq = est_b ? qmag * qmag : qq8, vv = est_b ? vb * vb : vn;
            const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, dot, qq + vv), 0.0f);
            const float d = __builtin_amdgcn_sqrtf(d2);
            sc = neg ? -d : __builtin_amdgcn_rcpf(1.0f + d);
        } else sc = dot;
        uint32_t key = valid ? score_to_key(sc) : kKeyMasked;
        if (j < 4u) p.scores[row] = valid ? f2u(sc) : kScoreSentinelBits;  // (nql == 1)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o));
        if (lane == 0) tpart[((tile - t0) & 1u) * 4u + wave] = key;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    ring_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave == 0) {
        const uint32_t* tp = tpart + ((t1 - 1u - t0) & 1u) * 4u;
        const uint32_t m = max(max(tp[0], tp[1]), max(tp[2], tp[3]));
        wmax = max(wmax, m);
        if (lane == 0) {
            p.tmax[t1 - 1u] = m;
            p.wmax[bx] = wmax;
        }
    }
}

template <int METRIC>
hipError_t launch_ring_i8_metric(const ScanParams& p, hipStream_t s) {
    const uint32_t blocks = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    const size_t lds = (size_t)kI8Stages * kI8StageBytes + kI8FacSlots * 3 * 64 * 4 + 8 * 4 + (size_t)p.ld * 2;
    auto kern = scan_ring_i8_kernel<METRIC>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int METRIC>
hipError_t launch_ring_metric(const ScanParams& p, hipStream_t s) {
    const uint32_t blocks = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    const size_t lds = (size_t)kRingStages * kRingStageBytes + 8 * 64 * 4 + 8 * 4 + (size_t)p.ld * 4;
    auto kern = scan_ring_kernel<METRIC>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, p);
    return hipGetLastError();
}

}  // namespace

// one unmasked f32 query, row strides of whole 128-element stages this kernel is built for
bool scan_ring_supported(uint32_t ld, uint32_t dim, int metric) {
    if (!(metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2)) return false;
    return dim <= ld && ld % 128u == 0 && ld >= 128u && ld <= 4096u;  // (the query behind the ring: 16 KiB at 4096 elements)
}

// p.nq == 1, p.nql == 1, no bitmap, p.tiles_per_wave = tiles per WORKGROUP, tmax / wmax / scores of query 0
hipError_t launch_scan_ring(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_ring_metric<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN:
        case kMetricNegL2: return launch_ring_metric<NMN_METRIC_EUCLIDEAN>(p, s);
        default: return launch_ring_metric<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

// one unmasked query over the 8-bit mirror: rows of whole 256-byte segments
bool scan_ring_i8_supported(uint32_t ld, uint32_t dim, int metric) {
    if (!(metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2)) return false;
    return dim <= ld && ld % 256u == 0 && ld >= 256u && ld <= 4096u;
}

// p.nq == 1, p.nql == 1, no bitmap, p.corpus_i8 / i8_scale / i8_vv / qi8 set, p.tiles_per_wave = tiles per WORKGROUP
hipError_t launch_scan_ring_i8(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_ring_i8_metric<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN:
        case kMetricNegL2: return launch_ring_i8_metric<NMN_METRIC_EUCLIDEAN>(p, s);
        default: return launch_ring_i8_metric<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

}  // namespace nmn
