// nmn_scan_mfma_kernel.h — (shared by nmn_scan_mfma.hip: bf16 / 8-bit mirrors, and nmn_scan_mfma_f32.hip: the f32 rows)
// batched-query scan (3..128 queries per corpus sweep) on the CDNA4 matrix cores.
//
// With nq queries the scan is a [rows x dim] x [dim x nq] product.  At nq = 64 the f32 VALU / f32-MFMA
// rate (157 TFLOP/s) would bound it at 6.25 ms per 10M x 768 sweep, above the HBM floor (SURVEY.md §7 hard
// part b), so the APPROXIMATE pass runs on bf16 MFMA.
//
// THREE streamed matrices share this kernel (template parameters I8 / F32): the shard's bf16 mirror (described first), its 8-bit
// mirror (see `I8` below) and — round 5 — the row-major F32 CORPUS itself (see `F32` below: the f32 rows through the same ring,
// rounded to bf16 in registers; SURVEY §8(d)'s bytes, and what a shard without a mirror runs).
//
// Corpus side: the sweep streams the shard's bf16 MIRROR (`half`, nmn_scan.hip: half_rows_kernel) — 2 bytes per
// element, the same matrix the 1-4 query VALU sweep reads — so a sweep moves rows*dim*2 bytes.  Its rounding is
// not compensated in the sweep: the mirror's MEASURED error norms (max |e_r| / |v_r|) go into the candidate margin
// (qprep_kernel) and the exact rescore (nmn_exact.hip) restores bit parity.  (An earlier version streamed a
// split hi+lo mirror at 4 bytes per element with three MFMAs per product; with the margin machinery in place the
// lo half bought nothing but traffic.)
// Query side: the stationary queries are rounded to bf16 as well (one MFMA per product); qprep_kernel measures each
// query's rounding error |q - bf16(q)| / |q| and adds it to that query's margin.  (Keeping a lo half of the queries —
// a second MFMA per product — cost 10 % of the sweep and bought a margin nobody needed.)
//
// Structure (one workgroup = 4 waves = 64 or 128 queries x 64-row tiles, persistent over a tile range):
//   * queries are STATIONARY in registers as MFMA B-fragments (v_mfma_f32_16x16x32_bf16; 4 VGPRs per 32-wide k-step
//     and query group).  Wave w owns query groups w (and w + 4 when the pass holds more than 64 queries) for the WHOLE
//     row: 96 VGPRs per group at dim 768, 192 at 1536;
//   * the corpus STREAMS through LDS: [64 rows][128*KS bf16] stages (16 / 32 KiB) filled by global_load_lds_dwordx4
//     (LDS-DMA: full row segments, no VGPRs) in a ring of 8 / 4 (all but one in flight, 112 / 96 KiB per CU);
//   * the LDS image is XOR-swizzled through the DMA SOURCE address (chunk ^= row & 15) so that the 16 rows of a
//     ds_read_b128 service group fall on 16 different bank slots;
//   * every wave reads the whole stage (one ds_read_b128 per 16-row block and k-step: 8 bf16 of one row per lane,
//     exactly the A fragment) — 4x the LDS traffic of splitting K over the waves, about a third of the LDS bandwidth at
//     the HBM rate — and in exchange its accumulators ARE the final dot products: no partial sums meeting through
//     LDS, no barrier beyond the stage hand-over (the K-split layout this replaced was 5-45 % slower at 64 queries,
//     45 % at dim 128, and could not go beyond 64);
//   * epilogue per tile and query group: scores (float4 per lane), per-(query,tile) maxima, per-(query,workgroup)
//     maxima — the hierarchy select_kernel consumes.
// Euclidean batches ride the same sweep: |q - v|^2 = |q|^2 + |v|^2 - 2 q.v from the dot product and the stored row
// magnitudes.  The expansion cancels for near neighbours, so its error is bounded in SQUARED-distance space
// (qprep_kernel: QInfo.pad < 0, applied by margin_key) and every candidate is re-scored exactly as always.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "nmn_internal.h"

namespace nmn {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#ifndef NMN_MFMA_BATCH
#define NMN_MFMA_BATCH 8  // A-fragment reads issued together, one batch ahead of the MFMAs that consume them
#endif
constexpr int kStageK = 128;     // granularity of the row length this kernel accepts (elements)
// A stage is [64 rows][128*KS bf16] (KS = 1 or 2 k-steps per wave and stage): 16 KiB or 32 KiB.  Rows whose length is a
// multiple of 256 use KS = 2: half as many stage hand-overs (a counted wait and a workgroup barrier each) per byte.
#ifndef NMN_MFMA_RING_KB   // measurement builds (tools/build_variant.sh): 64 + NMN_MFMA_OCC=2 puts two workgroups on a CU, 144 = 9 x 16 KiB
#define NMN_MFMA_RING_KB 128
#endif
#ifndef NMN_MFMA_OCC
#define NMN_MFMA_OCC 1
#endif
#ifdef NMN_MFMA_NO_FENCES  // (measurement build: the stage body left to the compiler's scheduler)
#define NMN_MFMA_FENCE() do { } while (0)
#else
#define NMN_MFMA_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int kRingBytes = NMN_MFMA_RING_KB * 1024;  // LDS given to the DMA ring: 8 stages of 16 KiB or 4 of 32 KiB

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

// two f32 -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 (compiler-visible, so the
// scheduler can interleave it with MFMAs; an inline-asm version is opaque to it)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}

// x[0..7] (f32) -> 8 bf16, round to nearest even
__device__ __forceinline__ s8 to_bf16x8(const f4& a, const f4& b) {
    u4 h;
    h[0] = cvt_pk_bf16(a.x, a.y);
    h[1] = cvt_pk_bf16(a.z, a.w);
    h[2] = cvt_pk_bf16(b.x, b.y);
    h[3] = cvt_pk_bf16(b.z, b.w);
    return __builtin_bit_cast(s8, h);
}

constexpr int kNormSlots = 16;   // tiles whose row magnitudes live in LDS at once: the ring's tiles in flight (<= kMaxRing) plus the
                                 // tile whose epilogue is deferred into the next tile's first stage

// issue the LDS-DMA of one stage into LDS buffer `buf`.  `stage_base` = mirror + (tile*64*ld + kc*128) elements
// (wave-uniform); `loff[pp]` = this lane's byte offset for piece pp, computed once per kernel.  A piece is one 1-KiB
// DMA instruction = 64/LR rows of LR = 16*KS chunks; wave w moves pieces PIECES*w .. PIECES*w + PIECES-1; lane i -> row
// (64/LR)*p + i/LR, LDS chunk i%LR, global chunk (i%LR) ^ (row&15)  (the swizzle lives on the source side: the LDS side
// of an LDS-DMA is always wave-base + lane*16).
template <int AUX, int PIECES>
__device__ __forceinline__ void stage_dma(const char* stage_base, const uint32_t (&loff)[PIECES], float* buf,
                                          uint32_t wave) {
#pragma unroll
    for (int pp = 0; pp < PIECES; pp++) {
        const char* src = stage_base + loff[pp];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + (wave * PIECES + (uint32_t)pp) * 256u),
                                         16, 0, AUX);  // AUX = 2: non-temporal
    }
}

template <int AUX, int PIECES>
__device__ __forceinline__ void stage_dma_piece(const char* stage_base, const uint32_t (&loff)[PIECES], float* buf, uint32_t wave,
                                                int pp) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stage_base + loff[pp]),
                                     (__attribute__((address_space(3))) void*)(buf + (wave * PIECES + (uint32_t)pp) * 256u), 16, 0, AUX);
}

// |v| of the 64 rows of a tile, also by LDS-DMA (one dword per lane): the streaming loop then contains
// no ordinary VGPR-destination load, so nothing makes the compiler drain the DMA queue with vmcnt(0).
__device__ __forceinline__ void norms_dma(const float* __restrict__ norms, uint64_t tile, float* nbuf, uint32_t lane) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(norms + tile * kTileRows + lane),
                                     (__attribute__((address_space(3))) void*)nbuf, 4, 0, 0);
}

// wait until at most `stages_after` younger stages (PIECES DMA ops each) are still in flight.  vmcnt retires in issue
// order on gfx9-class parts (loads, LDS-DMA and stores alike), so this guarantees the oldest stage has landed; the few
// extra ops some waves carry (norm DMA, epilogue stores) only make the wait slightly conservative.  (Counting those
// extras exactly — a per-wave tally and a branch tree that picks the immediate — was measured: no gain, the
// bookkeeping cost what the shorter waits saved.)
template <int N>
__device__ __forceinline__ void wait_vm_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int PIECES>
__device__ __forceinline__ void wait_stage(uint32_t stages_after) {
    static_assert(PIECES == 4 || PIECES == 8, "vmcnt immediates below");
    switch (stages_after * PIECES) {
        case 0: wait_vm_imm<0>(); break;
        case 4: wait_vm_imm<4>(); break;
        case 8: wait_vm_imm<8>(); break;
        case 12: wait_vm_imm<12>(); break;
        case 16: wait_vm_imm<16>(); break;
        case 20: wait_vm_imm<20>(); break;
        default: wait_vm_imm<24>(); break;
    }
}

// Workgroup = 4 waves, one per SIMD (the kernel needs 300-440 of the 512 registers a lone wave may use).
// QG = query groups of 16 kept stationary by the workgroup (4: 64 queries per sweep, 8: 128); wave w multiplies and
// finishes groups w, w + 4.  KC = stages per row (ld / (128*KS)), KS = 128-element k-blocks per stage.
// Euclidean score from the matrix-core dot product: |q - v|^2 = |q|^2 + |v|^2 - 2 q.v, score = 1 / (1 + sqrt(.)).
// The cancellation makes the ABSOLUTE error of the squared distance the quantity the margin bounds (qprep_kernel:
// QInfo.pad < 0); a slightly negative result of the subtraction is a distance of zero.  v_sqrt / v_rcp: 1 ulp each.
// NEG: the IVF list-scan metric, score = -distance.
template <bool NEG>
__device__ __forceinline__ float l2_score(float qq, float vn, float dot) {
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, dot, __builtin_fmaf(vn, vn, qq)), 0.0f);
    const float d = __builtin_amdgcn_sqrtf(d2);
    return NEG ? -d : __builtin_amdgcn_rcpf(1.0f + d);
}

// I8: the sweep streams the shard's 8-BIT mirror (nmn_scan_i8.hip: int8 codes, one scale per row) instead of the bf16 one — the
// same bytes-per-stage geometry (a stage is [64 rows][256 * KS bytes], a k-step 64 bytes of a row = one 16-byte fragment per
// lane), half the bytes per element.  The stationary queries are the int8 planes h, l of q = s_q (h + l / 256) + e_q (qprep):
// two v_mfma_i32_16x16x64_i8 per fragment into two int32 accumulator sets, exact integer arithmetic; the epilogue forms
// (h.c + l.c / 256) * s_q * s_r and the score from it.  Margins: qprep_kernel, approx_pass 1 | 2 | 4.
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool NEG>
__device__ __forceinline__ float l2_score_i8(float qq8, float vv, float dot) {  // |q~ - v~|^2 = |q~|^2 + |v~|^2 - 2 q~.v~
    const float d2 = __builtin_fmaxf(__builtin_fmaf(-2.0f, dot, qq8 + vv), 0.0f);
    const float d = __builtin_amdgcn_sqrtf(d2);
    return NEG ? -d : __builtin_amdgcn_rcpf(1.0f + d);
}

// F32: the sweep streams the ROW-MAJOR F32 CORPUS itself (no mirror: 4 bytes per element, SURVEY §8(d)'s bytes) — the same stage
// geometry in BYTES ([64 rows][256 * KS bytes] = 64 * KS f32 of a row), a k-step = 32 elements = 128 bytes of a row = TWO 16-byte
// LDS reads per lane, rounded to bf16 in registers (v_cvt_pk_bf16_f32, round to nearest even) on their way into the same
// v_mfma_f32_16x16x32_bf16.  HBM bytes are those of the f32 rows, read once per 64-128 queries; the rounding of the rows is
// bounded a priori (|e_r| <= 2^-8 |v_r|: qprep_kernel without measured error norms) instead of measured at a mirror's build.
// bytes of dynamic LDS the kernel uses without the running bound (= where the running bound's block starts); one formula for the
// kernel and for launch_one_mfma
template <int KS, int QG, int WAVES, bool I8>
__host__ __device__ constexpr uint32_t kRunLdsBase() {
    return (uint32_t)(kRingBytes + kNormSlots * 64 * 4 + (QG * 2 == WAVES ? QG * 64 * 4 * 16 : 0) +
                      WAVES * (QG * 2 == WAVES ? 1 : QG / WAVES) * 16 * 16 +
#ifdef NMN_MFMA_PACK
                      (I8 ? kNormSlots * 64 * 4 + WAVES * (16 * 64 + 16) * 4 : 0));
#else
                      (I8 ? kNormSlots * 64 * 4 : 0));
#endif
}
constexpr int kRunVals = 1024;  // running maxima per query: one per workgroup of the sweep (ScanParams::run_slots; sweeps of <= 1024 workgroups)
// the running bound's LDS block: the bound ring (every launch has it: the stage body reads a slot unconditionally), and — launches with
// run_S only — the queries' margins and a refresh buffer per wave
template <int QG>
__host__ __device__ constexpr uint32_t kRunLdsRing() { return (uint32_t)(kNormSlots * QG * 16 * 4); }
template <int QG, int WAVES>
__host__ __device__ constexpr uint32_t kRunLdsBytes() {
    return kRunLdsRing<QG>() + (uint32_t)(QG * 16 * (int)sizeof(QInfo) + WAVES * kRunVals * 4);
}

// ONE (8-bit form, round 6): ONE query plane — the stationary query is s_q h alone, one v_mfma_i32_16x16x64_i8 per fragment instead
// of two (and half the B-fragment registers).  The query's rounding |q - s_q h| is measured by qprep_kernel (approx_pass bit 16) and
// enters the margin like any other: ~2^-8 |q| instead of ~2^-16 |q|, i.e. about the rows' own 8-bit rounding again.  Cosine and dot
// product batches (nmn_scan_mfma_i8x.hip); 10M x 768, 64 queries: 1.53 -> 1.35 ms as a timing build (profiles/r06o_*).
template <int KC, int KS, int QG, int METRIC, bool MASKED, int AUX, int WAVES, bool I8 = false, bool F32 = false, bool ONE = false>
__global__ void __launch_bounds__(WAVES * 64, NMN_MFMA_OCC) scan_mfma_kernel(ScanParams p) {
    static_assert(!(I8 && F32), "one streamed matrix");
    static_assert(!ONE || I8, "one query plane: the 8-bit form only");
    constexpr int kStageElems = 128 * KS;                        // bf16 elements of a row per stage (I8: 256 * KS, F32: 64 * KS — the same BYTES)
    constexpr int kStageBytes = kTileRows * kStageElems * 2;     // 16 / 32 KiB of bf16
    constexpr int kRowPitch = kStageElems / 2;                   // LDS row pitch of a stage, in floats
    constexpr int kRing = kRingBytes / kStageBytes;              // 8 / 4 stages (all but one in flight)
    constexpr int kPieces = 16 * KS / WAVES;                     // 1-KiB DMA instructions per wave and stage (4 waves: 4 / 8, 8 waves: 2 / 4)
    constexpr uint32_t LR = 16 * KS;                             // lanes (16-B chunks) per row of a stage
    constexpr bool kL2 = METRIC == NMN_METRIC_EUCLIDEAN || METRIC == kMetricNegL2;  // 1/(1+d), or -d (IVF list scans)
    constexpr bool kNeedNorms = I8 || METRIC == NMN_METRIC_COSINE || kL2;  // |v| of the tile's rows (I8: their scales, always)
    // the score is the accumulator times a per-row factor (from LDS) times a per-query factor: cosine, and the 8-bit dot product
    constexpr bool kScaled = METRIC == NMN_METRIC_COSINE || (I8 && METRIC == NMN_METRIC_DOT_PRODUCT);
    extern __shared__ __attribute__((aligned(16))) float lds[];  // ring | norms
    float* nrm = lds + kRingBytes / 4;                           // [kNormSlots tiles][64] row magnitudes — cosine: their INVERSES
                                                                 // (ScanParams::inv_norms: one rcp per row at ingest, not 16 per lane here)
    // (I8, Euclidean) |v~|^2 of the tiles' rows, behind the pending tile maxima (see tk_pend)
    float* const nrm2 = nrm + kNormSlots * 64 + (QG * 2 == WAVES ? QG * 64 * 16 : 0) + WAVES * (QG * 2 == WAVES ? 1 : QG / WAVES) * 16 * 4;
    // (kPack) per wave: the scores of up to 16 writing queries of a tile, [16][64] f32 bits + their query numbers, behind nrm2
    // (measurement build -DNMN_MFMA_PACK: VERDICT r02 #2's packed stores.  Built, parity green, and measured against the direct
    //  stores on the same box: 10M x 768 cosine, 64 queries 1.51 -> 1.60 ms, 128 queries 2.87 -> 3.07 ms — the LDS round trip and
    //  the extra wave-uniform branch cost more than the three store instructions they save; only 5M x 1536 Euclidean gained,
    //  1.51 -> 1.42 ms.  Off.)
#ifdef NMN_MFMA_PACK
    constexpr bool kPack = I8;
#else
    constexpr bool kPack = false;
#endif
    float* const pack_lds = nrm2 + kNormSlots * 64;
    // (ScanParams::run_*) behind everything above — launch_one_mfma sizes the block for it:
    //   bnd   [kNormSlots tiles][QG * 16] the published bounds as picked up with each tile's row magnitudes (LDS-DMA by wave 0)
    //   qmrg  [QG * 16] QInfo            the margins of the workgroup's queries (read once, before the loop)          } launches with
    //   slotb [WAVES][kRunVals]          the workgroups' running maxima of the query a wave is refreshing (LDS-DMA)   } run_S only
    constexpr uint32_t kQ = (uint32_t)QG * 16u;
    uint32_t* const bnd = reinterpret_cast<uint32_t*>(lds) + (kRunLdsBase<KS, QG, WAVES, I8>() >> 2);
    QInfo* const qmrg = reinterpret_cast<QInfo*>(bnd + kNormSlots * kQ);
    uint32_t* const slotb = reinterpret_cast<uint32_t*>(qmrg + kQ);
    const uint32_t run_S = p.run_S;  // (scalar; 0 = the bound comes from p.skip_key, if any)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g = lane >> 4, n = lane & 15u;
    const uint32_t ld = p.ld;
    const uint32_t row_bytes = I8 ? ld : F32 ? ld * 4u : ld * 2u;  // bytes of one row of the streamed matrix
    // Workgroup -> (tile range bx, query block by).  With several query blocks the grid is 1-D and folded so that the
    // workgroups that stream the SAME tiles for different query blocks get ids 8 apart: same XCD (ids go round the 8 XCDs),
    // dispatched together — the second reader of a tile then finds it in that XCD's L2 / the infinity cache instead of
    // going to HBM again (gridDim.y == 1 marks the folded form; p.tile_step is unaffected).
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (p.fold_ny > 1) {
        const uint32_t span = 8u * p.fold_ny, grp_ = blockIdx.x / span, r_ = blockIdx.x % span;
        by = r_ / 8u;
        bx = grp_ * 8u + (r_ % 8u);
    }
    // a launch may cover only the workgroups [bx_base, bx_base + bx_count) of the sweep (nmn_api.hip: the bound that gates the
    // score stores is tightened between two such launches); workgroup ids keep their meaning for `wmax` and the selection
    if (p.bx_count && bx >= p.bx_count) return;  // (padding of the folded grid)
    bx += p.bx_base;
    const uint32_t q0 = by * (uint32_t)(QG * 16);
    // QG = 2 (rows of 2048 / 3072 / 4096 elements: half a group's B-fragments already take 128 / 192 / 256 VGPRs): the workgroup keeps 32
    // queries, and a group is shared by TWO waves that split the k-steps of every stage between them (kh = 0 / 1); their
    // partial sums meet once per tile through LDS and the kh = 0 wave finishes the group.
    // WAVES = 8 (two waves per SIMD, <= 256 registers each): the stalls of one wave — the barrier, the LDS round trips, the
    // epilogue — are covered by its SIMD partner; one wave per SIMD left the sweep at the edge of being issue-bound, and its
    // time moved 7 % from one box (clock) to the next.  64 queries: 4 groups x K-halves; 128 queries: 8 groups, whole K each.
    constexpr bool kHalfK = QG * 2 == WAVES;
    static_assert(kHalfK || QG % WAVES == 0, "query groups: one (or more) per wave, or one per wave PAIR");
    const uint32_t grp = kHalfK ? (wave % (uint32_t)QG) : wave;  // the (first) query group this wave multiplies
    const uint32_t kh = kHalfK ? (wave / (uint32_t)QG) : 0u;     // its half of the k-steps of a stage

    // ---- stationary operand: the wave's query groups x its k-steps of every stage -----------------
    constexpr int kStageSteps = F32 ? 2 * KS : 4 * KS;          // k-steps (one MFMA deep: 32 elements, I8: 64) a stage holds
    constexpr int kSteps = kHalfK ? kStageSteps / 2 : kStageSteps;  // ... of which this wave multiplies
    static_assert(kSteps >= 1, "K-halves need two k-steps per stage");
    constexpr int kBK = KC * kSteps;                  // ... of a row
    constexpr int kBG = kHalfK ? 1 : QG / WAVES;      // query groups of this wave: groups wave, wave + WAVES, ...
    s8 bhi[kBK][kBG];
    s8 blo[(I8 && !ONE) ? kBK : 1][kBG];  // (I8, two planes) the l plane of the query split
#pragma unroll
    for (int qg = 0; qg < kBG; qg++) {
        const uint32_t qq = q0 + ((uint32_t)qg * (uint32_t)WAVES + grp) * 16u + n;
        const bool ok = qq < p.nq;
        const float* qv = p.qpad + (size_t)(ok ? qq : q0) * ld;
        if constexpr (I8) {
            // qi8[q][2][ld] int8: the 16 bytes of k-step kc this lane's group multiplies, from the h plane and from the l plane
            const char* qb = reinterpret_cast<const char*>(p.qi8) + (size_t)(ok ? qq : q0) * 2u * ld;
#pragma unroll
            for (int kc = 0; kc < kBK; kc++) {
                const uint32_t k0b = ((uint32_t)(kc / kSteps) * (4u * KS) + kh * (uint32_t)kSteps + (uint32_t)(kc % kSteps)) * 64u + g * 16u;
                u4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
                if (ok) {
                    h = *reinterpret_cast<const u4*>(qb + k0b);
                    if constexpr (!ONE) l = *reinterpret_cast<const u4*>(qb + ld + k0b);
                }
                bhi[kc][qg] = __builtin_bit_cast(s8, h);
                if constexpr (!ONE) blo[kc][qg] = __builtin_bit_cast(s8, l);
            }
            continue;
        } else {
            blo[0][qg] = (s8){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int kc = 0; kc < kBK; kc++) {
            // k-step kc of this wave = k-step kh*kSteps + kc % kSteps of stage kc / kSteps
            const uint32_t k0 = ((uint32_t)(kc / kSteps) * (uint32_t)kStageSteps + kh * (uint32_t)kSteps + (uint32_t)(kc % kSteps)) * 32u + g * 8u;  // k0..k0+7
            f4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                a = *reinterpret_cast<const f4*>(qv + k0);
                b = *reinterpret_cast<const f4*>(qv + k0 + 4u);
            }
            bhi[kc][qg] = to_bf16x8(a, b);
        }
    }
    // The queries of this lane: C column n of query group h*4 + wave for each of the wave's kBG groups.
    constexpr int kHalves = kBG;
    constexpr int kAccGroups = kBG;
    uint32_t qn_h[kHalves], skip_h[kHalves], wmax_h[kHalves];
    bool q_ok_h[kHalves];
    float qmag_h[kHalves];
    float qsc_h[kHalves], qq8_h[kHalves];  // (I8) s_q and |q~|^2 of the query split
#pragma unroll
    for (int h = 0; h < kHalves; h++) {
        qn_h[h] = q0 + ((uint32_t)h * (uint32_t)WAVES + grp) * 16u + n;
        q_ok_h[h] = kh == 0 && (uint32_t)h * (uint32_t)WAVES + grp < (uint32_t)QG && qn_h[h] < p.nq;
        qmag_h[h] = q_ok_h[h] ? p.qinfo[qn_h[h]].qmag : 0.f;
        qsc_h[h] = (I8 && q_ok_h[h]) ? p.qinfo[qn_h[h]].qscale : 0.f;
        qq8_h[h] = (I8 && q_ok_h[h]) ? p.qinfo[qn_h[h]].qq8 : 0.f;
        skip_h[h] = (q_ok_h[h] && p.skip_key) ? p.skip_key[qn_h[h]] : kKeyNaN;  // kKeyNaN: write every tile
        wmax_h[h] = kKeyMasked;
    }

    if (run_S) {  // the margins of the workgroup's queries into LDS: the refresh below must not load from global memory inside the loop
        for (uint32_t i = threadIdx.x; i < kQ; i += (uint32_t)WAVES * 64u)
            qmrg[i] = p.qinfo[min(q0 + i, p.nq - 1u)];
        __syncthreads();
    }
    // the published bounds of the workgroup's queries, as they stand now, into slot `rel` of the bound ring (wave 0; 4 bytes per lane)
    // (picked up every kPick-th tile, for that tile and the kPick - 1 behind it: one DMA instruction per tile measured 0.17 ms of the
    //  8-bit sweep's 1.45 — profiles/r06k_*)
#ifndef NMN_RUN_PICK  // (measurement builds)
#define NMN_RUN_PICK 4
#endif
    constexpr uint32_t kPick = NMN_RUN_PICK;
    auto bound_dma = [&](uint32_t rel) __attribute__((always_inline)) {
        if (rel % kPick) return;
#pragma unroll
        for (uint32_t h = 0; h < kQ; h += 64u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.run_bound + q0 + h + lane),
                                             (__attribute__((address_space(3))) void*)(bnd + ((rel / kPick) % kNormSlots) * kQ + h), 4, 0, 16);  // sc1: agent scope —
            // the bounds are raised by OTHER compute units' atomics (at L2): a load that may hit this CU's vector cache would keep
            // reading the line it saw first (measured: every tile kept writing, the 8-bit sweep 1.5 -> 5.6 ms)
    };
    uint32_t pend_q = 0xFFFFFFFFu;  // (wave-uniform) the query whose slots this wave has in flight / in slotb
    const uint32_t tstep = p.tile_step;                  // 1, or S on the sampling pass (tile index i -> tile i*S)
    const bool sampling = tstep > 1;
    const uint32_t t0 = bx * p.tiles_per_wave;  // tiles per WORKGROUP on this path
    if (t0 >= p.n_tiles) return;
    const uint32_t t1 = min(t0 + p.tiles_per_wave, p.n_tiles);
    // Main sweep behind a sampling pass (p.skip_sampled = S): the tiles that are multiples of S were finished by that pass — tile
    // maxima in tmax, scores written — and are NOT streamed again (3 % of the matrix at S = 32).  The workgroup walks the other
    // tiles of its range: the j-th tile of the shard that is not a multiple of S is j + j / (S - 1) + 1; below tile t there are
    // t - ceil(t / S) of them.  S = 0: every tile (j is the tile).
    const uint32_t S = sampling ? 0u : p.skip_sampled;
    auto walked_below = [&](uint32_t t) -> uint32_t { return S ? t - (t + S - 1u) / S : t; };
    auto tile_of = [&](uint32_t j) -> uint32_t { return S ? j + j / (S - 1u) + 1u : j; };
    const uint32_t j0 = walked_below(t0), j1 = walked_below(t1);
    const uint32_t n_stage = (j1 - j0) * KC;

    uint32_t loff[kPieces];  // per-lane source byte offsets of the DMA pieces this wave moves per stage
#pragma unroll
    for (int pp = 0; pp < kPieces; pp++) {
        const uint32_t r = (64u / LR) * (wave * kPieces + (uint32_t)pp) + lane / LR;
        loff[pp] = r * row_bytes + (((lane % LR) ^ (r & 15u)) * 16u);
    }
    const char* const mirror = I8 ? reinterpret_cast<const char*>(p.corpus_i8)
                                  : F32 ? reinterpret_cast<const char*>(p.corpus) : reinterpret_cast<const char*>(p.corpus_half);
    // per-row factor of the epilogue: 1 / |v| (cosine) or |v| (Euclidean); I8: s_r / |v| (cosine) or s_r (dot, Euclidean), and
    // for Euclidean a second array, |v~|^2 of the row as stored
    const float* const norm_src = I8 ? (METRIC == NMN_METRIC_COSINE ? p.i8_cos : p.i8_scale)
                                     : (METRIC == NMN_METRIC_COSINE ? p.inv_norms : p.norms);
    auto stage_src = [&](uint32_t tile_, uint32_t kc_) -> const char* {
        return mirror + (uint64_t)tile_ * tstep * kTileRows * row_bytes + (uint64_t)kc_ * (kStageElems * 2u);
    };

    // prologue: stages 0..kRing-2 in flight (stage s lives in ring slot s % kRing)
#pragma unroll
    for (uint32_t s0 = 0; s0 < kRing; s0++) {
        if (s0 < n_stage) {
            // (the magnitudes of stage kRing - 1's tile too: its pieces go out during the first iteration)
            if (run_S && !(p.run_dbg & 8u) && wave == 0 && s0 % KC == 0) bound_dma(s0 / KC);
            if (kNeedNorms && wave == 0 && s0 % KC == 0)
            {
                norms_dma(norm_src, (uint64_t)tile_of(j0 + s0 / KC) * tstep, nrm + ((s0 / KC) % kNormSlots) * 64u, lane);
                if constexpr (I8 && kL2) norms_dma(p.i8_vv, (uint64_t)tile_of(j0 + s0 / KC) * tstep, nrm2 + ((s0 / KC) % kNormSlots) * 64u, lane);
            }
            if (s0 < kRing - 1)
                stage_dma<AUX, kPieces>(stage_src(tile_of(j0 + s0 / KC), s0 % KC), loff, lds + (s0 % kRing) * (kStageBytes / 4), wave);
        } else if (s0 < kRing - 1) {
            // A range shorter than the ring (a small shard: one tile per workgroup is KC stages, the ring of 16-KiB stages holds
            // eight): the loop's counted wait — "at most kRing - 2 younger stages in flight" — only says that stage sidx has landed
            // if kRing - 1 stages WERE issued here.  The missing ones go out as the same dummy pieces the loop issues past the end
            // of the range (every lane re-reads the first 16 bytes of the mirror into a slot nobody reads).  Without them the wait
            // returned at once and the first stage was read on the strength of whatever else had drained the queue by then —
            // found when the 8-bit sweep got its bitmap variant (20 000 x 768, 64 queries: rows missing from 47 answers).
            const uint32_t zero[kPieces] = {};
            stage_dma<AUX, kPieces>(mirror, zero, lds + (s0 % kRing) * (kStageBytes / 4), wave);
        }
    }

    // LDS offset (floats) of this lane's 16-B read per row block and k-step ks: row n, chunk ks*4+g of the row's 16*KS
    // (swizzled ^ n)
    // (F32: a k-step is chunks 8 ks' + 2 g and + 2 g + 1 of the row — 8 consecutive f32, the same k order as the bf16 form; the two
    //  chunks differ in their lowest bit only, so the swizzled pair is off[ks] and off[ks] ^ 4 floats)
    uint32_t off[kSteps];
#pragma unroll
    for (int ks = 0; ks < kSteps; ks++)
        off[ks] = F32 ? n * kRowPitch + ((((kh * (uint32_t)kSteps + (uint32_t)ks) * 8u + g * 2u) ^ n) * 4u)
                      : n * kRowPitch + ((((kh * (uint32_t)kSteps + (uint32_t)ks) * 4u + g) ^ n) * 4u);

    // ---- epilogue of one tile: scores, per-(query,tile) maximum, score writes — for the accumulators `facc` of tile `ftile`.
    // (One wave per SIMD: nothing overlaps it, so it is kept short — see kLazy below.  Deferring it into the next tile's first
    // stage was tried: its branches (partial tiles, score writes) cut that stage's basic block in two and cost the read / MFMA
    // interleave more than the overlap returned.  Round 5 tried it again branch-free: the arithmetic in seven chunks behind the MFMAs of the
    // next tile's first seven k-steps, predicated by selects, the stores behind its second stage — exact (117 tests), f32 rows 4.91 vs
    // 4.93 ms, bf16 mirror 2.60 vs 2.66: nothing / worse; tools/micro/mfma_deferred_epilogue.patch, profiles/r05zj_*.)
    // tile maxima of the current group of four tiles (see publish): [wave][query group of the wave][16 queries][4 tiles] in LDS
    uint32_t* const tk_pend = reinterpret_cast<uint32_t*>(nrm + kNormSlots * 64 + (kHalfK ? QG * 64 * 16 : 0));
    auto finish_half = [&](auto half_c, const f4 (&facc)[4][kAccGroups], uint32_t ftile, uint32_t frel, bool flast, const f4 (&npre)[4],
                           const uint32_t (&skp)[kHalves]) __attribute__((always_inline)) {
        constexpr int H = decltype(half_c)::value;
        const uint32_t qn = qn_h[H];
        const bool q_ok = q_ok_h[H];
        const float qmag = qmag_h[H];
        // (running bound: the value picked up with this tile's row magnitudes — kRing - 1 stages old, and every older value of a
        //  bound that only rises is a valid one)
        const uint32_t skip = (run_S && !(p.run_dbg & 1u)) ? skp[H] : skip_h[H];  // (skp: read from the bound ring at the top of the tile's last stage, like npre)
        f4 fin[4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) fin[rb] = facc[rb][H];
        // C layout: col = lane&15 (query), row = rb*16 + (lane>>4)*4 + reg
        const uint64_t rtile = (uint64_t)ftile * tstep;  // real tile index (sampling pass: every tstep-th)
        const uint64_t r0 = rtile * kTileRows;
        // (the tile's row magnitudes — inverse magnitudes, 8-bit scales — were read into `npre` at the top of the tile's LAST stage, under
        //  its MFMAs: the epilogue starts without an LDS round trip)
        const float* nslot2 = nrm2 + (frel % kNormSlots) * 64u;  // (I8, Euclidean)
        (void)nslot2;
        uint64_t mword = ~0ull;
        if constexpr (MASKED) {
            // one bitmap for the batch, or one per query (lanes with the same n = the same query: same word)
            const uint64_t* mq = p.qmasks ? (q_ok ? p.qmasks[qn] : nullptr) : p.mask;
            if (mq) mword = mq[rtile];
        }
        const uint64_t left = p.n_rows - r0;
        if (left < 64) mword &= (1ull << left) - 1ull;
        // per-query factor: 1 / |q| (cosine); I8: s_q / |q| (cosine), s_q (dot product; also what scales the Euclidean dot)
        const float qsc = qsc_h[H];
        const float inv_q = I8 ? (METRIC == NMN_METRIC_COSINE ? (qmag == 0.f ? 0.f : qsc * __builtin_amdgcn_rcpf(qmag)) : qsc)
                               : (qmag == 0.f ? 0.f : __builtin_amdgcn_rcpf(qmag));
        const float qq = I8 ? qq8_h[H] : qmag * qmag;
        (void)inv_q;
        (void)qq;
        // what follows a tile's key: the maximum over the four lane groups of a query (v_permlane32_swap / v_permlane16_swap: no
        // LDS round trip), the tile and workgroup maxima, and whether this lane's query writes the tile's scores.  Scores are
        // only worth their HBM write when the tile can still hold a candidate: with a per-query bound from the sampling pass
        // ~2 % of the tiles qualify (64 queries x 10M rows would otherwise write 2.56 GB per sweep, +1.45 ms on a 5.3 ms sweep).
        auto publish = [&](uint32_t tkey) __attribute__((always_inline)) -> bool {
            const auto r32 = __builtin_amdgcn_permlane32_swap(tkey, tkey, false, false);
            tkey = max((uint32_t)r32[0], (uint32_t)r32[1]);
            const auto r16 = __builtin_amdgcn_permlane16_swap(tkey, tkey, false, false);
            tkey = max((uint32_t)r16[0], (uint32_t)r16[1]);
            // Tile maxima leave in groups of four tiles (one 16-byte store per query instead of four 4-byte ones): a store costs the
            // wave its issue slot for 100+ cycles behind the DMA pieces whatever it carries, and this one is paid on EVERY tile.
            // Tiles at the ragged ends of the workgroup's range (and everything when the rows of tmax are not 16-byte aligned)
            // go out one by one.
#ifdef NMN_MFMA_TMAX_SINGLE  // A/B build: one store per tile
            if (q_ok && g == 0) p.tmax[(uint64_t)qn * p.tmax_stride + ftile] = tkey;
#else
            {
                // (the four keys of a group wait in LDS, one 16-byte slot per query: registers are what the 128-query kernel has none of)
                const uint32_t slot = ftile & 3u;  // (wave-uniform)
                uint32_t* mine = tk_pend + (((uint32_t)wave * (uint32_t)kHalves + (uint32_t)H) * 16u + n) * 4u;
                if (g == 0) mine[slot] = tkey;
                if (slot == 3u || flast) {
                    // (a group whose first tile was the sampling pass's — multiples of S are multiples of 4 — keeps that entry)
                    const uint32_t g0 = ftile & ~3u, first = max((S && g0 % S == 0u) ? g0 + 1u : g0, t0);
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
#endif
            // (running bound: the workgroup's maximum so far, published when it rises — a plain store into the workgroup's OWN word, rare
            //  after the first tiles.  The first form kept 128 slot maxima per query by atomicMax from every tile that reached the bound:
            //  ~1M device-scope atomics per batch, queued in order with the DMA pieces in front of every counted wait — 4.95 -> 7.0 ms.)
            if (run_S && !(p.run_dbg & 2u) && q_ok && g == 0 && tkey > wmax_h[H] && !sampling) p.run_slots[(size_t)qn * (uint32_t)kRunVals + bx] = tkey;
            wmax_h[H] = max(wmax_h[H], tkey);
            // the sampling pass finishing its tiles for the main sweep (p.tmax_main set): the key into the sweep's tmax as well, and the
            // tile's scores written whatever they are (no bound exists yet; 1/S of the tiles)
            const bool finish_sampled = sampling && p.tmax_main != nullptr;
            if (finish_sampled && q_ok && g == 0) p.tmax_main[(uint64_t)qn * p.tmax_main_stride + rtile] = tkey;
#ifdef NMN_MFMA_NO_SCORE_WRITES
            return false;
#else
            const bool wr_ = q_ok && (!sampling || finish_sampled) && tkey != kKeyMasked && tkey >= skip;
            return wr_;
#endif
        };
        // The two cases are two complete code paths (key, publish, stores): merged behind one `publish` the compiler carried
        // the score words of the rare path through the common one — 16 registers zeroed per tile, and earlier the scaled
        // products parked in AGPRs and fetched back (32 moves) for the one tile in thirty that writes.
        // Cosine / dot product with every row taking part (the common case): only the tile MAXIMUM is needed, so the per-row
        // work is one multiply by the row's inverse magnitude and a max; the query's 1/|q| (>= 0: monotone, rounding included)
        // is applied once to the maximum, and the 16 score words are formed — multiplying again — only where they are written.
        constexpr bool kLazy = METRIC == NMN_METRIC_COSINE || METRIC == NMN_METRIC_DOT_PRODUCT;
        // Euclidean score of one row from its accumulator
        auto l2_of = [&](float acc_v, float vn_v, float vv_v) __attribute__((always_inline)) -> float {
            if constexpr (I8) return l2_score_i8<METRIC == kMetricNegL2>(qq, vv_v, acc_v * (inv_q * vn_v));  // vn_v = s_r here
            else return l2_score<METRIC == kMetricNegL2>(qq, vn_v, acc_v);
        };
        if (mword == ~0ull) {
            float m = -__builtin_inff();
            u4 bits[4];
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
                f4 sc = fin[rb];
                if constexpr (kScaled) {
                    // (a zero row has inverse magnitude 0: its score is 0 like cosine_similarity's; v_rcp at ingest: 1 ulp,
                    // the margin has 1000x that slack)
                    sc = sc * npre[rb];
                }
                if constexpr (kL2) {
                    const f4 vn = npre[rb];
                    f4 vv = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (I8) vv = *reinterpret_cast<const f4*>(nslot2 + (uint32_t)rb * 16u + g * 4u);
#pragma unroll
                    for (int e = 0; e < 4; e++) sc[e] = l2_of(sc[e], vn[e], vv[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (!kLazy) bits[rb][e] = f2u(sc[e]);
                    m = __builtin_fmaxf(m, sc[e]);  // v_max_f32 skips NaNs; an all-NaN lane reports -inf, an upper bound of its key
                }
            }
            if constexpr (kScaled) m = m * inv_q;
            const bool wr = publish(score_to_key(m));
            auto words = [&](int rb) __attribute__((always_inline)) -> u4 {
                u4 w;
                if constexpr (kLazy) {
                    f4 sc = fin[rb];
                    if constexpr (kScaled) sc = sc * npre[rb];
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = f2u(kScaled ? sc[e] * inv_q : sc[e]);
                } else {
                    w = bits[rb];
                }
                return w;
            };
            if constexpr (kPack) {
                // Packed stores.  A query that writes a tile owns 256 contiguous bytes of scores[] — but in the accumulator layout they
                // sit in FOUR lanes x four registers, so the direct form is four store instructions per (tile, query group) with as
                // few as 4 of 64 lanes live, and a vector-memory instruction costs the wave its issue slot for 100+ cycles behind the
                // DMA pieces whatever it carries.  Under the 8-bit margin 11 % of the (tile, query) pairs write (85 % of the groups
                // have a writer): 0.18 of the sweep's 1.55 ms.  Here the writers park their 64 scores in a per-wave LDS block, and
                // 16 lanes per written query store 16 bytes each: ONE instruction per four written queries.
                const unsigned long long wm = __ballot(wr);
                if (wm) {  // (wave-uniform)
                    const uint32_t qm = (uint32_t)((wm | (wm >> 16) | (wm >> 32) | (wm >> 48)) & 0xFFFFull);  // queries (n) that write
                    const uint32_t rank = (uint32_t)__builtin_popcount(qm & ((1u << n) - 1u));
                    float* const stage = pack_lds + wave * (16u * 64u + 16u);
                    uint32_t* const qsel = reinterpret_cast<uint32_t*>(stage + 16u * 64u);
                    if (wr) {
#pragma unroll
                        for (int rb = 0; rb < 4; rb++) *reinterpret_cast<u4*>(stage + rank * 64u + (uint32_t)rb * 16u + g * 4u) = words(rb);
                        if (g == 0) qsel[rank] = qn;
                    }
                    asm volatile("" ::: "memory");  // (the LDS queue of a wave is in order: the reads below see the writes above)
                    const uint32_t nw = (uint32_t)__builtin_popcount(qm);
                    for (uint32_t b = 0; b < nw; b += 4u) {
                        const uint32_t slot = b + (lane >> 4), piece = lane & 15u;
                        if (slot < nw) {
                            const u4 v = *reinterpret_cast<const u4*>(stage + slot * 64u + piece * 4u);
                            *reinterpret_cast<u4*>(p.scores + score_at(r0 + piece * 4u, qsel[slot], p.nql)) = v;
                        }
                    }
                    asm volatile("" ::: "memory");
                }
            } else if (wr) {
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
                    *reinterpret_cast<u4*>(p.scores + score_at(r0 + (uint32_t)rb * 16u + g * 4u, qn, p.nql)) = words(rb);
            }
        } else {
            uint32_t tkey = kKeyMasked;
            u4 bits[4];
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
                const uint32_t rr = (uint32_t)rb * 16u + g * 4u;  // first of this lane's 4 rows
                f4 vn = {1.f, 1.f, 1.f, 1.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (kNeedNorms) vn = npre[rb];
                if constexpr (I8 && kL2) vv = *reinterpret_cast<const f4*>(nslot2 + rr);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const bool valid = ((mword >> (rr + (uint32_t)e)) & 1ull) != 0;
                    float sc = fin[rb][e];
                    if constexpr (kScaled) sc = (sc * vn[e]) * inv_q;  // vn = 1 / |v| here (I8: s_r / |v|, or s_r)
                    if constexpr (kL2) sc = l2_of(sc, vn[e], vv[e]);
                    bits[rb][e] = valid ? f2u(sc) : kScoreSentinelBits;
                    if (valid) tkey = max(tkey, score_to_key(sc));
                }
            }
            if (publish(tkey)) {
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
                    *reinterpret_cast<u4*>(p.scores + score_at(r0 + (uint32_t)rb * 16u + g * 4u, qn, p.nql)) = bits[rb];
            }
        }
    };
    auto finish_tile = [&](const f4 (&facc)[4][kAccGroups], uint32_t ftile, uint32_t frel, bool flast, const f4 (&npre)[4],
                           const uint32_t (&skp)[kHalves]) __attribute__((always_inline)) {
#ifdef NMN_MFMA_NO_EPILOGUE
        {  // measurement only (-DNMN_MFMA_NO_EPILOGUE build): the sweep without its epilogue (answers are wrong)
            float sink_v = 0.f;
#pragma unroll
            for (int rb = 0; rb < 4; rb++)
#pragma unroll
                for (int qg = 0; qg < kAccGroups; qg++) sink_v += facc[rb][qg][0] + facc[rb][qg][1] + facc[rb][qg][2] + facc[rb][qg][3];
            if (sink_v == 1.2345e-30f) p.tmax[0] = 1u;
            return;
        }
#endif
        if (!kHalfK || kh == 0) finish_half(std::integral_constant<int, 0>{}, facc, ftile, frel, flast, npre, skp);
        if constexpr (kHalves > 1) finish_half(std::integral_constant<int, 1>{}, facc, ftile, frel, flast, npre, skp);
    };

    uint32_t sidx = 0;  // running stage index of this workgroup
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t tile = tile_of(j);
        f4 acc[4][kAccGroups];  // [row block][query group of this wave]
        f4 npre[4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};  // the tile's per-row factors (see finish_half)
        uint32_t skp[kHalves];  // the running bounds of this lane's queries as picked up for this tile (meaningful when run_S)
#pragma unroll
        for (int h = 0; h < kHalves; h++) skp[h] = kKeyNaN;
        v4i ach[I8 ? 4 : 1][kAccGroups], acl[I8 ? 4 : 1][kAccGroups];  // (I8) int32 sums of the h plane / the l plane
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int qg = 0; qg < kAccGroups; qg++) {
                acc[rb][qg] = (f4){0.f, 0.f, 0.f, 0.f};
                if constexpr (I8) {
                    ach[rb][qg] = (v4i){0, 0, 0, 0};
                    acl[rb][qg] = (v4i){0, 0, 0, 0};
                }
            }
#pragma unroll
        for (int kc = 0; kc < KC; kc++, sidx++) {
            const float* buf = lds + (sidx % kRing) * (kStageBytes / 4);
            // RAW: stage sidx has landed once every wave saw its own pieces retire (counted vmcnt: the
            // younger stages stay in flight) and all waves met at the barrier.  WAR: a wave reaches this
            // barrier only after consuming (lgkmcnt) its reads of stage sidx-1, whose ring slot is the
            // one the DMA issued right below (stage sidx+kRing-1) overwrites.
            wait_vm_imm<(kRing - 2) * kPieces>();  // (pieces are issued for every stage, real or dummy: always kRing - 2 younger stages)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // The pieces of stage sidx + kRing - 1 go into the ring slot consumed one iteration ago.  They are NOT issued here in
            // one burst: an LDS-DMA instruction holds the wave's issue port for 100-185 cycles when four waves fire eight each
            // right behind the barrier (measured with s_memtime: 1150 of the ~2400 cycles of a stage, more than its MFMAs) but
            // only ~25-60 in the shadow of running MFMAs — so they are spread over the stage's MFMA stream below, one per
            // k-step.  And they are issued UNCONDITIONALLY: past the end of the workgroup's range every lane re-reads the first
            // 16 bytes of the mirror into a slot nobody will read (one cache line per instruction), so that the stage body is
            // one basic block — a branch around each piece would cut it into nine scheduling regions and with them the
            // read / MFMA / DMA interleave laid out below.
            const uint32_t ns = sidx + (kRing - 1);
            const bool issue = ns < n_stage;
            const uint32_t nt = tile_of(j0 + ns / KC), nkc = ns % KC;
            const char* const nsrc = issue ? stage_src(nt, nkc) : mirror;
            const uint32_t lmask = issue ? 0xFFFFFFFFu : 0u;  // (scalar: the per-lane offsets are ANDed away in the tail)
            float* const nbuf = lds + (ns % kRing) * (kStageBytes / 4);
            // The stage body, in the order it is meant to issue (every __builtin_amdgcn_sched_barrier(0) is a fence the
            // scheduler moves nothing across): the A-fragment reads of batch b + 1 go out BEFORE the MFMAs of batch b, so the
            // LDS round trip of a batch hides behind a whole batch of MFMAs; each k-step's 4 * kBG MFMAs are followed by that
            // k-step's DMA piece(s) of the stage ahead (see above).  A fragment = one ds_read_b128: chunk g of the k-step = 8
            // consecutive bf16 of row n.  (sched_group_barrier pipelines could not place the LDS-DMA instructions — they are
            // both VMEM and DS to the scheduler — and left them in one clump.)
            constexpr int kKsPerBatch = F32 ? 1 : (kBK * kBG * (I8 ? 8 : 4) >= (I8 ? 192 : 256)) ? 1 : 2;  // k-steps whose fragments are read together (8 reads);
                                                                         // 192+ VGPRs of stationary fragments: 4 reads at a time
            static_assert(kSteps % kKsPerBatch == 0, "batches tile the stage");
            constexpr int kNB = kSteps / kKsPerBatch;
            s8 a[F32 ? 1 : kSteps][4];
            u4 raw[F32 ? kSteps : 1][4][2];  // (F32) the two 16-byte halves of a fragment as read: converted right before its MFMAs
            auto read_batch = [&](int b) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = b * kKsPerBatch; ks < (b + 1) * kKsPerBatch; ks++)
#pragma unroll
                    for (int rb = 0; rb < 4; rb++) {
                        if constexpr (F32) {
                            // LDS chunk (c ^ n) of row n holds the row's chunk c (the swizzle lives on the DMA source side): off[ks] is the
                            // k-step's chunk 2 g (elements 0..3 of this lane's eight), off[ks] ^ 4 floats its chunk 2 g + 1 (elements 4..7)
                            const uint32_t o0 = off[ks], o1 = off[ks] ^ 4u;
                            raw[ks][rb][0] = *reinterpret_cast<const u4*>(buf + rb * 16 * kRowPitch + o0);
                            raw[ks][rb][1] = *reinterpret_cast<const u4*>(buf + rb * 16 * kRowPitch + o1);
                        } else {
                            a[ks][rb] = __builtin_bit_cast(s8, *reinterpret_cast<const u4*>(buf + rb * 16 * kRowPitch + off[ks]));
                        }
                    }
            };
#ifdef NMN_MFMA_BURST_DMA  // A/B: all pieces in one burst behind the barrier (round 1's order)
#pragma unroll
            for (int pp = 0; pp < kPieces; pp++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc + (loff[pp] & lmask)),
                                                 (__attribute__((address_space(3))) void*)(nbuf + (wave * kPieces + (uint32_t)pp) * 256u), 16, 0, AUX);
            NMN_MFMA_FENCE();
#endif
            if (kc == KC - 1) {  // (compile time) — unconditionally: a branch on run_S would cut the stage's basic block in two
#pragma unroll
                for (int h = 0; h < kHalves; h++)
                    skp[h] = bnd[(((j - j0) / (uint32_t)NMN_RUN_PICK) % kNormSlots) * kQ + ((uint32_t)h * (uint32_t)WAVES + grp) * 16u + n];  // (kPick = 4)
            }
            if constexpr (kNeedNorms) {
                if (kc == KC - 1) {  // (compile time: the stage loop is unrolled)
#pragma unroll
                    for (int rb = 0; rb < 4; rb++)
                        npre[rb] = *reinterpret_cast<const f4*>(nrm + ((j - j0) % kNormSlots) * 64u + (uint32_t)rb * 16u + g * 4u);
                }
            }
            read_batch(0);
            NMN_MFMA_FENCE();
#pragma unroll
            for (int b = 0; b < kNB; b++) {
                if (b + 1 < kNB) read_batch(b + 1);
#pragma unroll
                for (int ks = b * kKsPerBatch; ks < (b + 1) * kKsPerBatch; ks++) {
                    s8 afrag[4];  // (F32) the k-step's fragments, rounded to bf16
                    (void)afrag;
#pragma unroll
                    for (int rb = 0; rb < 4; rb++)
#pragma unroll
                        for (int qg = 0; qg < kBG; qg++) {
                            if constexpr (I8) {
                                const v4i av = __builtin_bit_cast(v4i, a[ks][rb]);
#ifdef NMN_MFMA_NO_MFMA  // (measurement build: the fragments are read and folded, no matrix-core work — wrong answers, timing only)
                                ach[rb][qg] ^= av;
                                continue;
#endif
                                ach[rb][qg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, __builtin_bit_cast(v4i, bhi[kc * kSteps + ks][qg]), ach[rb][qg], 0, 0, 0);
#ifndef NMN_MFMA_I8_NO_LO  // (measurement build: the sweep without the l plane's products — wrong answers, timing only)
                                if constexpr (!ONE) acl[rb][qg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, __builtin_bit_cast(v4i, blo[kc * kSteps + ks][qg]), acl[rb][qg], 0, 0, 0);
#endif
                            } else if constexpr (F32) {
#ifdef NMN_MFMA_F32_NOCVT  // (measurement build: the fragments go to the matrix cores unconverted — wrong answers, timing only)
                                if (qg == 0) afrag[rb] = __builtin_bit_cast(s8, raw[ks][rb][0] ^ raw[ks][rb][1]);
#else
                                if (qg == 0) afrag[rb] = to_bf16x8(__builtin_bit_cast(f4, raw[ks][rb][0]), __builtin_bit_cast(f4, raw[ks][rb][1]));
#endif
                                acc[rb][qg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[rb], bhi[kc * kSteps + ks][qg], acc[rb][qg], 0, 0, 0);
                            } else {
                                acc[rb][qg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][rb], bhi[kc * kSteps + ks][qg], acc[rb][qg], 0, 0, 0);
                            }
                        }
                    // one piece of the stage ahead per k-step (K-halves: two, their waves multiply half the k-steps of a stage each)
#ifndef NMN_MFMA_BURST_DMA
                    // (pieces ks * kPieces / kSteps .. (ks + 1) * kPieces / kSteps: two, one, or one every other k-step)
#pragma unroll
                    for (int pp = ks * kPieces / kSteps; pp < (ks + 1) * kPieces / kSteps; pp++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc + (loff[pp] & lmask)),
                                                         (__attribute__((address_space(3))) void*)(nbuf + (wave * kPieces + (uint32_t)pp) * 256u), 16, 0, AUX);
#endif
                    NMN_MFMA_FENCE();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // row magnitudes of the tile whose first stage was just issued (its epilogue is >= kRing - 1 stages away); outside
            // the stage's basic block.  One extra entry in wave 0's vmcnt queue per tile: its next waits are one piece conservative.
            // (issued for the tile of stage ns + 1, i.e. BEFORE that stage's pieces go out in the next iteration: in-order vmcnt then
            // lands it with them, and every wave passes a barrier behind wave 0's wait before the tile's epilogue reads it)
            if (run_S && !(p.run_dbg & 8u) && wave == 0 && ns + 1u < n_stage && (ns + 1u) % KC == 0) bound_dma((ns + 1u) / KC);
            if (kNeedNorms && wave == 0 && ns + 1u < n_stage && (ns + 1u) % KC == 0) {
                const uint32_t nrel = (ns + 1u) / KC, nt1 = tile_of(j0 + nrel);
                norms_dma(norm_src, (uint64_t)nt1 * tstep, nrm + (nrel % kNormSlots) * 64u, lane);
                if constexpr (I8 && kL2) norms_dma(p.i8_vv, (uint64_t)nt1 * tstep, nrm2 + (nrel % kNormSlots) * 64u, lane);
            }
        }
        if constexpr (I8) {
            // h.c + (l.c) / 256: both sums are exact integers well below 2^24 * 256 (rows of <= 1536 elements: |h.c| <= 1536 * 127^2
            // = 2.5e7 — converted with one rounding of 2^-24 relative, far inside the margin's f32 slack)
#pragma unroll
            for (int rb = 0; rb < 4; rb++)
#pragma unroll
                for (int qg = 0; qg < kAccGroups; qg++)
                    if constexpr (ONE) {
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[rb][qg][e] = (float)ach[rb][qg][e];  // (exact: |h.c| <= 3072 * 127^2 < 2^26, one rounding of 2^-24)
                    } else
#pragma unroll
#ifdef NMN_MFMA_I8_INT_COMBINE  // (measurement build, VERDICT r05 #4's second lever: the planes combined in integers — one shift-add and
                    // one conversion per element instead of two conversions and an FMA; exact only while |h.c| < 2^23, i.e. rows of <= 512
                    // elements: at 768 the answers are WRONG, timing only — the 1/256 would fold into the per-query factor)
                    for (int e = 0; e < 4; e++) acc[rb][qg][e] = (float)((ach[rb][qg][e] << 8) + acl[rb][qg][e]);
#else
                    for (int e = 0; e < 4; e++) acc[rb][qg][e] = (float)ach[rb][qg][e] + (float)acl[rb][qg][e] * 0.00390625f;
#endif
        }
        if constexpr (kHalfK) {
            // the two K-halves of a group meet: wave kh = 1 publishes, wave kh = 0 adds and finishes.  (The next
            // publication is a whole tile of stage barriers away: no second barrier needed.)
            float* xch = nrm + kNormSlots * 64;  // [2 groups][64 lanes][4 row blocks] f4
            if (kh == 1) {
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
                    *reinterpret_cast<f4*>(xch + ((grp * 64u + lane) * 4u + (uint32_t)rb) * 4u) = acc[rb][0];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kh == 0) {
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
                    acc[rb][0] += *reinterpret_cast<const f4*>(xch + ((grp * 64u + lane) * 4u + (uint32_t)rb) * 4u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            finish_tile(acc, tile, j - j0, j + 1u == j1, npre, skp);  // (the wave pairs already pay a barrier per tile here: their epilogue stays in place)
        } else {
            finish_tile(acc, tile, j - j0, j + 1u == j1, npre, skp);
        }
        if (run_S && !(p.run_dbg & 4u)) {  // (scalar branch) the running bound's refresh: one query per wave every kRefresh tiles
            // Workgroup b refreshes at its tiles rel = b (mod kRefresh) — so at every moment a part of the resident workgroups is
            // refreshing, each of them a different query of its waves' groups ((rel / kRefresh + b) mod 16): every query of
            // the batch is refreshed by somebody every few tiles, from the sweep's first tiles on.  (All workgroups refreshing at rel = 0 mod 16,
            // as first built, found nothing published at rel = 0 and came back at rel = 16: twenty tiles of every workgroup wrote all
            // their scores before the first bound existed.)  The maxima are consumed kDelay tiles after they were asked for: by then at
            // least kRing - 2 stages of pieces have been issued behind them and waited for (in-order vmcnt).
#ifndef NMN_RUN_REFRESH  // (measurement builds)
#define NMN_RUN_REFRESH 64
#endif
            constexpr uint32_t kRefresh = NMN_RUN_REFRESH;  // (16 measured 0.28 ms of the 8-bit sweep's 1.45: the waves of a workgroup meet at every stage's barrier,
                                               //  so a refreshing wave holds the other three up)
            constexpr uint32_t kDelay = (uint32_t)((kRing - 2 + KC - 1) / KC);
            static_assert(kDelay >= 1 && kDelay < kRefresh, "refresh timing");
            const uint32_t rel = j - j0, phase = bx % kRefresh;
            if ((rel % kRefresh) == (phase + kDelay) % kRefresh && pend_q != 0xFFFFFFFFu) {
                // The run_S-th largest of the maxima (run_S = k) is reached by k different workgroups, i.e. k different
                // tiles: a valid lower bound on the k-th best approximate score.  Bitwise search over the upper 22 bits of the key
                // (count of values >= candidate by ballots: no lane exchange), the low 10 bits left zero — a slightly lower bound.
                wait_vm_imm<(kRing - 2) * kPieces>();  // (what every stage's wait asks for: kDelay tiles of pieces are behind the maxima)
                uint32_t v[kRunVals / 64];
#pragma unroll
                for (int i = 0; i < kRunVals / 256; i++) {
                    const u4 x = *reinterpret_cast<const u4*>(slotb + wave * (uint32_t)kRunVals + (uint32_t)i * 256u + lane * 4u);
                    v[4 * i + 0] = x[0]; v[4 * i + 1] = x[1]; v[4 * i + 2] = x[2]; v[4 * i + 3] = x[3];
                }
                uint32_t pre = 0;
                for (int bit = 31; bit >= 10; bit--) {
                    const uint32_t c = pre | (1u << bit);
                    uint32_t cnt = 0;
#pragma unroll
                    for (int i = 0; i < kRunVals / 64; i++) cnt += (uint32_t)__builtin_popcountll(__ballot(v[i] >= c));
                    if (cnt >= run_S) pre = c;  // (wave-uniform)
                }
                if (pre > kKeyNaN && lane == 0) atomicMax(p.run_bound + pend_q, margin_key(pre, qmrg[pend_q - q0]));
            }
            if ((rel % kRefresh) == phase) {
                const uint32_t turn = ((rel / kRefresh) + bx) % (16u * (uint32_t)kBG);  // (phase = bx mod 64 fixes bx mod 16: query t is asked for at phases t, t + 16, t + 32, t + 48)
                const uint32_t qq = q0 + ((turn / 16u) * (uint32_t)WAVES + grp) * 16u + (turn % 16u);
                pend_q = (kh == 0 && qq < p.nq) ? qq : 0xFFFFFFFFu;
                if (pend_q != 0xFFFFFFFFu) {
#pragma unroll
                    for (int i = 0; i < kRunVals / 256; i++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.run_slots + (size_t)pend_q * (uint32_t)kRunVals + (uint32_t)i * 256u + lane * 4u),
                                                         (__attribute__((address_space(3))) void*)(slotb + wave * (uint32_t)kRunVals + (uint32_t)i * 256u), 16, 0, 16);  // sc1 (as for the bounds)
                }
            }
        }
    }
    wait_vm_imm<0>();  // the dummy pieces of the tail must have landed before this workgroup's LDS is handed to the next one
    if (sampling) return;  // the sampling pass leaves tile maxima (and, finishing its tiles for the main sweep, their scores)
#pragma unroll
    for (int h = 0; h < kHalves; h++)
        if (q_ok_h[h] && g == 0) {
            // the workgroup's maximum covers every tile of its range: those the sampling pass finished come from tmax
            if (S)
                for (uint32_t ts = ((t0 + S - 1u) / S) * S; ts < t1; ts += S)
                    wmax_h[h] = max(wmax_h[h], p.tmax[(uint64_t)qn_h[h] * p.tmax_stride + ts]);
            p.wmax[(size_t)qn_h[h] * p.wmax_stride + bx] = wmax_h[h];
        }
}

template <int KC, int KS, int QG, int METRIC, bool MASKED, int WAVES, bool I8 = false, bool F32 = false, bool ONE = false>
static hipError_t launch_one_mfma(const ScanParams& p, hipStream_t s) {
    const uint32_t blocks_all = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    if (p.bx_base >= blocks_all) return hipSuccess;
    const uint32_t blocks = p.bx_count ? std::min(p.bx_count, blocks_all - p.bx_base) : blocks_all - p.bx_base;
    const uint32_t ny = (p.nq + QG * 16 - 1) / (QG * 16);
    ScanParams pf = p;
    pf.bx_count = blocks;
    dim3 grid(blocks, ny);
    static const bool no_fold = getenv("NMN_MFMA_NO_FOLD") != nullptr;
    if (ny > 1 && !no_fold) {  // folded 1-D grid (see the kernel): tile ranges padded to a multiple of 8
        pf.fold_ny = ny;
        grid = dim3(((blocks + 7u) / 8u) * 8u * ny, 1);
    }
    // ring | row magnitudes | the K-halves' exchange | pending tile maxima | (8-bit) |v~|^2 of the tiles' rows (with the K-halves'
    // exchange of the long rows the packed-store blocks of the measurement build would not fit in 160 KiB) | (run_S) the running bound's block
    const size_t lds = (size_t)kRunLdsBase<KS, QG, WAVES, I8>() + (p.run_S ? (size_t)kRunLdsBytes<QG, WAVES>() : (size_t)kRunLdsRing<QG>());
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (p.skip_sampled && (p.skip_sampled < 4u || (p.skip_sampled & 3u))) return hipErrorInvalidValue;  // (tile_of divides by S - 1; tmax groups of four)
    if (p.run_S && (p.run_S > 256u || blocks_all > (uint32_t)kRunVals || p.tile_step > 1u || (p.skip_sampled && !(p.run_dbg & 1u)))) return hipErrorInvalidValue;
    // AUX = 2: non-temporal LDS-DMA (the mirror is read once)
#ifndef NMN_MFMA_AUX  // cache policy of the LDS-DMA (cpol bits: 1 sc0, 2 nt, 16 sc1); measurement builds override
#define NMN_MFMA_AUX 2
#endif
    auto kern = scan_mfma_kernel<KC, KS, QG, METRIC, MASKED, NMN_MFMA_AUX, WAVES, I8, F32, ONE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, pf);
    return hipGetLastError();
}

}  // namespace nmn
