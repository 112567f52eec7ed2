// nmn_ingest.hip — ONE pass over freshly written f32 rows: their magnitudes in the reference's order, their bf16
// mirror, and the mirror's rounding-error norms.  (Round 1 took three: norms_kernel read the rows 8 lanes per row, 4 bytes
// per lane — 2.07 TB/s —, half_rows_kernel read them again to write the mirror, half_err_kernel folded the errors.)
//
// THIS TRANSLATION UNIT IS BUILT WITH -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt (build.py): the magnitude
// is simd::magnitude (tensor_store/src/hnsw.rs:198-229) bit for bit — eight accumulators acc[l] = acc[l] + (v[8c+l]*v[8c+l])
// over the chunks c in order, multiply and add rounded separately, the lanes summed left to right from -0.0, sqrt
// correctly rounded.  (Only rows with dim % 8 == 0 come here: no scalar tail; a zero-padded chunk adds +0.0 exactly.)
//
// Shape of the work: the chain of accumulator lane l is sequential over the chunks of a row, so a row offers 8-fold
// parallelism and no more.  Sixteen lanes per row (the sweep's load shape) would pass every partial sum from lane to lane
// through DPP with one lane in eight doing useful work per step.  Instead a LANE OWNS A ROW — 8 independent chains, no
// cross-lane traffic at all — and coalescing is restored by staging through LDS:
//   * wave = 64 rows (a tile).  The tile streams through the wave's own LDS ring in stages of [64 rows][32 floats] = 8 KiB,
//     filled by global_load_lds_dwordx4 (eight 1-KiB instructions, each 8 rows x 128 contiguous bytes = whole lines), 3 of 4
//     stages in flight, counted s_waitcnt vmcnt — no workgroup barrier anywhere: the four waves of a workgroup never meet.
//   * the LDS image is XOR-swizzled through the DMA source address (16-byte chunk ^= (row >> 1) & 7) so that the
//     ds_read_b128 of 64 lanes, each at its own row, is conflict-free.
//   * per stage and lane: 8 ds_read_b128, 32 mul + 32 add (the chains), 16 v_cvt_pk_bf16_f32, the error terms; the bf16
//     halves of two stages (128 B per row = one line) go through a second swizzled LDS buffer and leave as 8 coalesced
//     16-byte-per-lane stores of 8 rows x 128 B.
//   * maxima (|v|, |e_r|, |e_r| / |v_r|) are kept per wave in registers across all its tiles: 3 atomics per wave per launch.
// Traffic: rows * ld * 4 read + rows * ld * 2 written (+ 4 B/row) — 46.1 GB for 10M x 768; bound: HBM.
#include <algorithm>

#include "nmn_internal.h"

#pragma clang fp contract(off)

namespace nmn {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

constexpr int kStageFloats = 32;                 // floats of a row per stage (4 chunks of the reference's 8)
constexpr int kStageBytes = 64 * kStageFloats * 4;  // 8 KiB
constexpr int kRing = 4;                         // stages per wave (3 in flight)
constexpr int kOutBytes = 64 * 128;              // bf16 of two stages: 128 B per row
constexpr int kWaveLds = kRing * kStageBytes + kOutBytes;  // 40 KiB
constexpr int kWaves = 4;                        // per workgroup: 160 KiB of LDS, one workgroup per CU

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off));
    return v;
}

// HALF: also write the bf16 mirror and fold its rounding errors
template <bool HALF>
__global__ void __launch_bounds__(kWaves * 64, 1) ingest_kernel(const float* __restrict__ corpus, uint32_t ld, uint64_t row0,
                                                                uint64_t n, float* __restrict__ norms, float* __restrict__ inv_norms,
                                                                uint32_t* __restrict__ max_norm_bits, float* __restrict__ half,
                                                                uint32_t* __restrict__ err_bits) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ring = lds_all + wave * (kWaveLds / 4);
    float* obuf = ring + kRing * (kStageBytes / 4);
    const uint64_t n_tiles = (n + 63) / 64;
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;
    if (gw >= n_tiles) return;
    const uint32_t KC = ld / kStageFloats;                        // stages per tile
    const uint64_t my_tiles = (n_tiles - gw + n_waves - 1) / n_waves;
    const uint64_t n_stage = my_tiles * KC;
    const uint32_t swz = (lane >> 1) & 7u;                        // this lane's row swizzle (lane == row of the tile)

    // DMA source offsets of the 8 pieces of a stage: piece p, lane i -> row 8p + i/8 of the tile, LDS chunk i%8, source
    // chunk (i%8) ^ ((row >> 1) & 7).  Rows past the end are clamped to the last row (loaded, never stored).
    const uint32_t pr = lane >> 3, pc = lane & 7u;
    auto src_of = [&](uint64_t tile, uint32_t kc, uint32_t p) -> const char* {
        const uint32_t r = 8u * p + pr;
        uint64_t gi = tile * 64 + r;
        if (gi >= n) gi = n - 1;
        return reinterpret_cast<const char*>(corpus + (row0 + gi) * (uint64_t)ld + (uint64_t)kc * kStageFloats) +
               ((pc ^ ((r >> 1) & 7u)) * 16u);
    };
    // the stage the DMA issues next: (tile, kc, ring slot), advanced incrementally (a 64-bit division per stage costs more
    // than the stage's arithmetic)
    uint64_t i_tile = gw;
    uint32_t i_kc = 0, i_slot = 0;
    auto issue = [&]() {
        float* dst = ring + i_slot * (kStageBytes / 4);
#pragma unroll
        for (uint32_t p = 0; p < 8; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_of(i_tile, i_kc, p),
                                             (__attribute__((address_space(3))) void*)(dst + p * 256u), 16, 0, 2);  // nt: read once
        i_slot = (i_slot + 1u) & (kRing - 1u);
        if (++i_kc == KC) {
            i_kc = 0;
            i_tile += n_waves;
        }
    };
    static_assert((kRing & (kRing - 1)) == 0, "ring slots wrap with a mask");
#pragma unroll
    for (uint32_t s = 0; s < kRing - 1; s++)
        if (s < n_stage) issue();

    float acc[8];
    float err2 = 0.f;
    float mx_norm = 0.f, mx_err = 0.f, mx_rel = 0.f;
    uint64_t tile = gw;
    uint32_t kc = 0, slot = 0;
    for (uint64_t s = 0; s < n_stage; s++) {
        if (kc == 0) {
#pragma unroll
            for (int l = 0; l < 8; l++) acc[l] = 0.0f;
            err2 = 0.f;
        }
        // the oldest stage has landed once at most the younger ones (8 DMA each) are outstanding; bf16 stores in
        // flight only make the wait conservative (stores and loads share vmcnt)
        const uint64_t after = n_stage - 1 - s;
        if (after >= kRing - 2) wait_vm<(kRing - 2) * 8>();
        else if (after == 1) wait_vm<8>();
        else wait_vm<0>();
        asm volatile("" ::: "memory");
        if (s + kRing - 1 < n_stage) issue();  // into the slot consumed (lgkmcnt(0) below) one iteration ago
        const float* buf = ring + slot * (kStageBytes / 4) + lane * kStageFloats;
        v4f x[8];
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) x[c] = *reinterpret_cast<const v4f*>(buf + ((c ^ swz) * 4u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the reference's chains: chunk c8 = 4 kc + j holds x[2j] (lanes 0-3) and x[2j+1] (lanes 4-7)
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float a = x[2 * j][e], b = x[2 * j + 1][e];
                const float pa = a * a, pb = b * b;
                acc[e] = acc[e] + pa;
                acc[4 + e] = acc[4 + e] + pb;
            }
        }
        if constexpr (HALF) {
            // 32 floats -> 16 packed pairs = four 16-byte chunks of the mirror row; rounding error folded on the way
            u4v pk[4];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const f2v lo = {x[c][0], x[c][1]}, hi = {x[c][2], x[c][3]};
                const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf2v));
                const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf2v));
                const float e0 = x[c][0] - __uint_as_float(p0 << 16), e1 = x[c][1] - __uint_as_float(p0 & 0xFFFF0000u);
                const float e2 = x[c][2] - __uint_as_float(p1 << 16), e3 = x[c][3] - __uint_as_float(p1 & 0xFFFF0000u);
                err2 = err2 + e0 * e0;
                err2 = err2 + e1 * e1;
                err2 = err2 + e2 * e2;
                err2 = err2 + e3 * e3;
                pk[c >> 1][(c & 1) * 2] = p0;
                pk[c >> 1][(c & 1) * 2 + 1] = p1;
            }
            // into the out buffer: row = lane, 16-byte chunk (kc & 1) * 4 + c of the 128-byte pair line, swizzled like the ring
#pragma unroll
            for (uint32_t c = 0; c < 4; c++)
                *reinterpret_cast<u4v*>(obuf + lane * 32u + (((((kc & 1u) * 4u) + c) ^ swz) * 4u)) = pk[c];
            const bool pair_done = (kc & 1u) == 1u || kc + 1 == KC;
            if (pair_done) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t valid_chunks = (kc & 1u) ? 8u : 4u;  // a row length of an odd number of stages ends on half a line
                const uint32_t pair = kc >> 1;
#pragma unroll
                for (uint32_t p = 0; p < 8; p++) {
                    const uint32_t r = 8u * p + pr;
                    const u4v v = *reinterpret_cast<const u4v*>(obuf + r * 32u + ((pc ^ ((r >> 1) & 7u)) * 4u));
                    const uint64_t gi = tile * 64 + r;
                    if (gi < n && pc < valid_chunks)
#ifdef NMN_INGEST_PLAIN_STORES  // A/B build
                        *reinterpret_cast<u4v*>(reinterpret_cast<char*>(half) + (row0 + gi) * (uint64_t)ld * 2ull + (uint64_t)pair * 128ull +
                                                pc * 16u) = v;
#else
                        // (non-temporal: the mirror is written once and not read by this kernel)
                        __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(reinterpret_cast<char*>(half) + (row0 + gi) * (uint64_t)ld * 2ull +
                                                                              (uint64_t)pair * 128ull + pc * 16u));
#endif
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next pair overwrites the buffer
            }
        }
        if (kc + 1 == KC) {
            float r = -0.0f;
#pragma unroll
            for (int l = 0; l < 8; l++) r = r + acc[l];
            const float mag = __builtin_sqrtf(r);
            const uint64_t gi = tile * 64 + lane;
            const bool live = gi < n;
            if (live) {
                norms[row0 + gi] = mag;
                inv_norms[row0 + gi] = mag == 0.0f ? 0.0f : 1.0f / mag;  // for the batched cosine sweep's epilogue
            }
            if (live && mag == mag) mx_norm = __builtin_fmaxf(mx_norm, mag);
            if constexpr (HALF) {
                const float e = __builtin_sqrtf(err2) * 1.0005f;  // (slack: the reference of this bound is a different summation order)
                if (live && e == e && e > 0.f) {
                    mx_err = __builtin_fmaxf(mx_err, e);
                    if (mag > 0.f) {
                        const float rel = e / mag * 1.0005f;
                        if (rel == rel) mx_rel = __builtin_fmaxf(mx_rel, rel);
                    }
                }
            }
        }
        slot = (slot + 1u) & (kRing - 1u);
        if (++kc == KC) {
            kc = 0;
            tile += n_waves;
        }
    }
    // non-negative floats: bit order == value order
    mx_norm = wave_max(mx_norm);
    if (lane == 0 && mx_norm > 0.f) atomicMax(max_norm_bits, __float_as_uint(mx_norm));
    if constexpr (HALF) {
        mx_err = wave_max(mx_err);
        mx_rel = wave_max(mx_rel);
        if (lane == 0 && mx_err > 0.f) atomicMax(err_bits, __float_as_uint(mx_err));
        if (lane == 0 && mx_rel > 0.f) atomicMax(err_bits + 1, __float_as_uint(mx_rel));
    }
}

// ---- exact score of EVERY row for flagged queries, a lane per row -------------------------------------------------------------
// Euclidean (lib.rs:2249-2253) is ONE strictly sequential sum over the elements of a row, s = ((((-0.0 + d0*d0) + d1*d1) + ...),
// d = q - v, multiply and add rounded separately: a row offers no parallelism at all, and the eight-lanes-per-row form of
// nmn_exact.hip (euclid_sumsq_seq: products by eight lanes, the sum pulled through shuffles in element order) runs at one add
// per lane group and shuffle — 11 ms per 10M x 768 sweep, 85 ms at 1536: the cliff behind every Euclidean exact fallback and
// k > 4096 search.  Dot product / cosine (hnsw.rs:168-193) are eight chains per row, which the same form serves at 5.9 TB/s.
// Here a LANE OWNS A ROW and walks it in element order from the LDS ring the ingest kernel streams through
// (global_load_lds_dwordx4, swizzled, three stages in flight per wave), the query broadcast from LDS: 64 independent rows per
// wave, 6.5 TB/s.  Zero padding of row and query adds +0.0 terms, which leave every sum as it is.  Scores go out exactly as
// rescore_kernel's fallback duty / exact_scan_kernel write them: f32 bits at score_at(row, q, nql), the sentinel for rows the
// bitmap excludes or past the end.
struct ExactRowsParams {
    const float* corpus;
    const float* norms;             // cosine
    const float* qpad;              // [nq][ld]
    const QInfo* qinfo;             // cosine: |q|
    const QState* qstate;           // nullable: only flagged queries are computed
    int flagged;                    // 1: overflow == 1 (the pipeline's fallback duty); 2: overflow != 0 (exact_scan_kernel's rule)
    const uint64_t* mask;           // nullable
    const uint64_t* const* qmasks;  // nullable [nq]
    uint32_t* scores;
    uint64_t n_rows;
    uint32_t ld, nql, nq;
    int metric;
};

// FAM 0: Euclidean family (one sequential sum); FAM 1: dot product / cosine (eight strided accumulators)
template <int FAM>
__global__ void __launch_bounds__(kWaves * 64, 1) exact_rows_kernel(ExactRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const uint32_t q = blockIdx.y;
    if (p.qstate) {  // (block-uniform)
        const uint32_t o = p.qstate[q].overflow;
        if (p.flagged == 1 ? o != 1u : o == 0u) return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kRingLds = kRing * kStageBytes;  // 32 KiB per wave
    float* ring = lds_all + wave * (kRingLds / 4);
    float* qlds = lds_all + kWaves * (kRingLds / 4);  // [ld] the query, shared by the four waves
    const uint32_t ld = p.ld;
    for (uint32_t i = threadIdx.x; i < ld; i += kWaves * 64) qlds[i] = p.qpad[(size_t)q * ld + i];
    __syncthreads();
    const uint64_t n = p.n_rows, n_tiles = (n + 63) / 64;
    const uint64_t gw = (uint64_t)blockIdx.x * kWaves + wave, n_waves = (uint64_t)gridDim.x * kWaves;
    if (gw >= n_tiles) return;
    const uint32_t KC = ld / kStageFloats;
    const uint64_t my_tiles = (n_tiles - gw + n_waves - 1) / n_waves;
    const uint64_t n_stage = my_tiles * KC;
    const uint32_t swz = (lane >> 1) & 7u;
    const uint32_t pr = lane >> 3, pc = lane & 7u;
    auto src_of = [&](uint64_t tile, uint32_t kc, uint32_t piece) -> const char* {
        const uint32_t r = 8u * piece + pr;
        uint64_t gi = tile * 64 + r;
        if (gi >= n) gi = n - 1;  // (loaded, never scored)
        return reinterpret_cast<const char*>(p.corpus + gi * (uint64_t)ld + (uint64_t)kc * kStageFloats) + ((pc ^ ((r >> 1) & 7u)) * 16u);
    };
    uint64_t i_tile = gw;
    uint32_t i_kc = 0, i_slot = 0;
    auto issue = [&]() {
        float* dst = ring + i_slot * (kStageBytes / 4);
#pragma unroll
        for (uint32_t piece = 0; piece < 8; piece++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_of(i_tile, i_kc, piece),
                                             (__attribute__((address_space(3))) void*)(dst + piece * 256u), 16, 0, 2);
        i_slot = (i_slot + 1u) & (kRing - 1u);
        if (++i_kc == KC) {
            i_kc = 0;
            i_tile += n_waves;
        }
    };
#pragma unroll
    for (uint32_t st = 0; st < kRing - 1; st++)
        if (st < n_stage) issue();

    const uint64_t* mask = p.qmasks ? p.qmasks[q] : p.mask;
    const float qmag = (FAM == 1 && p.metric == NMN_METRIC_COSINE) ? p.qinfo[q].qmag : 0.0f;
    float sum = -0.0f;
    float acc[8];
    uint64_t tile = gw;
    uint32_t kc = 0, slot = 0;
    for (uint64_t st = 0; st < n_stage; st++) {
        if (kc == 0) {
            sum = -0.0f;
#pragma unroll
            for (int l = 0; l < 8; l++) acc[l] = 0.0f;
        }
        const uint64_t after = n_stage - 1 - st;
        if (after >= kRing - 2) wait_vm<(kRing - 2) * 8>();
        else if (after == 1) wait_vm<8>();
        else wait_vm<0>();
        asm volatile("" ::: "memory");
        if (st + kRing - 1 < n_stage) issue();
        const float* buf = ring + slot * (kStageBytes / 4) + lane * kStageFloats;
        const float* qc = qlds + kc * kStageFloats;
        v4f x[8], y[8];
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            x[c] = *reinterpret_cast<const v4f*>(buf + ((c ^ swz) * 4u));
            y[c] = *reinterpret_cast<const v4f*>(qc + c * 4u);  // (every lane the same address: a broadcast)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (FAM == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float d = y[c][e] - x[c][e];
                    const float pr2 = d * d;
                    sum = sum + pr2;
                }
        } else {
            // the reference's chains: chunk 4 kc + j holds x[2j] (accumulators 0-3) and x[2j+1] (accumulators 4-7)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float pa = y[2 * j][e] * x[2 * j][e], pb = y[2 * j + 1][e] * x[2 * j + 1][e];
                    acc[e] = acc[e] + pa;
                    acc[4 + e] = acc[4 + e] + pb;
                }
        }
        if (kc + 1 == KC) {
            const uint64_t row = tile * 64 + lane;
            bool valid = row < n;
            if (valid && mask) valid = ((mask[row >> 6] >> (row & 63)) & 1ull) != 0;
            float sc;
            if constexpr (FAM == 0) {
                if (p.metric == kMetricNegL2Sq) sc = -sum;
                else {
                    const float dist = __builtin_sqrtf(sum);
                    sc = p.metric == kMetricNegL2 ? -dist : 1.0f / (1.0f + dist);
                }
            } else {
                float r = -0.0f;
#pragma unroll
                for (int l = 0; l < 8; l++) r = r + acc[l];
                sc = r;
                if (p.metric == NMN_METRIC_COSINE) {
                    const float vmag = valid ? p.norms[row] : 1.0f;
                    sc = (qmag == 0.0f || vmag == 0.0f) ? 0.0f : r / (qmag * vmag);  // lib.rs:2257-2266: one mul, one div, in that order
                }
            }
            // (the score matrix is padded to whole tiles: rows past the end get the sentinel like excluded ones)
            p.scores[score_at(row, q, p.nql)] = valid ? __float_as_uint(sc) : kScoreSentinelBits;
        }
        slot = (slot + 1u) & (kRing - 1u);
        if (++kc == KC) {
            kc = 0;
            tile += n_waves;
        }
    }
}

}  // namespace

// what the lane-per-row kernel takes: Euclidean family with rows of whole 32-float stages; dot product / cosine in addition with
// whole reference chunks (dim % 8 == 0: no scalar tail); the query fits LDS next to the rings
bool exact_rows_supported(uint32_t ld, uint32_t dim, int metric) {
    const bool l2 = metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2 || metric == kMetricNegL2Sq;
    const bool dot = metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT;
    return (l2 || (dot && dim % 8u == 0)) && dim >= 1 && ld % kStageFloats == 0 && ld >= (uint32_t)kStageFloats && ld <= 6144u;  // 128 KiB of rings + the query <= 152 KiB of LDS
}

// exact scores of every row for the flagged queries (all `nq` when qstate is null) -> scores[]
hipError_t launch_exact_rows(const float* corpus, const float* norms, uint32_t ld, uint64_t n_rows, const float* qpad, const QInfo* qinfo,
                             const QState* qstate, int flagged, const uint64_t* mask, const uint64_t* const* qmasks, uint32_t* scores,
                             uint32_t nql, uint32_t nq, int metric, hipStream_t s) {
    if (n_rows == 0 || nq == 0) return hipSuccess;
    ExactRowsParams p{};
    p.corpus = corpus;
    p.norms = norms;
    p.qpad = qpad;
    p.qinfo = qinfo;
    p.qstate = qstate;
    p.flagged = flagged;
    p.mask = mask;
    p.qmasks = qmasks;
    p.scores = scores;
    p.n_rows = n_rows;
    p.ld = ld;
    p.nql = nql;
    p.nq = nq;
    p.metric = metric;
    const uint64_t n_tiles = (n_rows + 63) / 64;
    // (one workgroup per CU and some; with many queries in the pass fewer per query: the launch returns at once unless a query is
    //  flagged, and 512 x 64 workgroups that only look at a flag and leave cost 21 us per 64-query batch)
    const uint32_t per_q = std::min<uint32_t>(256u, std::max<uint32_t>(32u, 2048u / std::max<uint32_t>(nq, 1u)));  // (one workgroup per CU: it holds 128 KiB of LDS, and an idle launch waits for as many CUs to fall free under the other stream's sweep)
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_tiles + kWaves - 1) / kWaves, per_q);
    const size_t lds = (size_t)kWaves * kRing * kStageBytes + (size_t)ld * 4;
    const bool l2 = metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2 || metric == kMetricNegL2Sq;
    auto kern = l2 ? exact_rows_kernel<0> : exact_rows_kernel<1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks, nq), dim3(kWaves * 64), lds, s, p);
    return hipGetLastError();
}

namespace {

// ---- ONE read of freshly written rows: magnitudes in the reference's order AND their 8-bit mirror --------------------------------
// What a shard whose mirror is the 8-bit one derives from new rows (until round 4 three kernels read them three
// times — ingest_kernel for the magnitudes, q8_rows_kernel twice over for max |v| and the codes, q8_err_kernel for the error
// maxima).  The int8 code of an element needs the row's scale s_r = max |v| / 127, i.e. the whole row, before the first code
// can be formed: the row has to wait ON CHIP.  64 rows x 3 KB do not fit a wave's share of the LDS, so here the shape is the
// sweep's instead of the ingest kernel's: SIXTEEN LANES PER ROW, four rows per wave and step, the row held in REGISTERS —
// lane l of a row group loads the 16-byte chunks l, l + 16, ... (KJ = ld / 64 non-temporal loads, whole 256-byte segments per
// row and instruction), so the row is 4 * KJ VGPRs (48 at 768 elements, 128 at 2048 — the longest this kernel takes).
//   * magnitude, hnsw.rs:198-229: accumulator e of the reference's eight walks the chunks of eight c = 0, 1, 2, ... in order.
//     Element 64 j + 4 l + t sits in load j, lane l, component t: chunk c = 8 j + l / 2, accumulator e = 4 (l & 1) + t.  So a
//     chain runs through the lanes of one parity, 0 -> 2 -> ... -> 14 (odd: 1 -> ... -> 15), then on to load j + 1 — a
//     rotate-right-by-two of the 16-lane DPP row per step: acc = row_ror:2(acc) + v * v, four independent chains (t) per lane
//     interleaved.  Every lane executes every step; the lane pair the chain has reached holds its true value, the others hold
//     values nobody reads (pair p at step p takes what pair p - 1 computed at step p - 1, before that pair overwrites it).
//     Seven lanes in eight idle through the chain: 8 * KJ dependent steps per four rows — a fraction of what the row's bytes
//     cost to fetch (see the traffic note below), and the price of not reading them twice.  Multiply and add are rounded
//     separately (this translation unit is built with -ffp-contract=off), the eight sums meet left to right from -0.0, sqrt is
//     correctly rounded: simd::magnitude bit for bit, as ingest_kernel computes it.
//   * scale, codes, |s c|^2, s / |v|, the row's quantization error |v - s c|: exactly q8_rows_kernel's definitions
//     (nmn_scan_i8.hip: a non-finite element -> scale 0, codes 0, infinite error), reductions over the DPP row.
//   * maxima (|v|, |e_r|, |e_r| / |v_r|) per wave in registers: three atomics per wave and launch.
// Traffic: rows * ld * 4 read + rows * ld written (+ 24 B per row) — 38.6 GB for 10M x 768; bound: HBM.
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float row16_max(float v) {  // all-reduce over a 16-lane DPP row (row_ror 8, 4, 2, 1)
    v = __builtin_fmaxf(v, dppf<0x128>(v));
    v = __builtin_fmaxf(v, dppf<0x124>(v));
    v = __builtin_fmaxf(v, dppf<0x122>(v));
    v = __builtin_fmaxf(v, dppf<0x121>(v));
    return v;
}
__device__ __forceinline__ float row16_add(float v) {
    v = v + dppf<0x128>(v);
    v = v + dppf<0x124>(v);
    v = v + dppf<0x122>(v);
    v = v + dppf<0x121>(v);
    return v;
}
__device__ __forceinline__ int row16_addi(int v) {
    v += dppi<0x128>(v);
    v += dppi<0x124>(v);
    v += dppi<0x122>(v);
    v += dppi<0x121>(v);
    return v;
}

constexpr int kQ8Waves = 4;  // waves per workgroup (they never meet)

template <int KJ>
__global__ void __launch_bounds__(kQ8Waves * 64) ingest_q8_kernel(const float* __restrict__ corpus, uint32_t ld, uint64_t row0, uint64_t n,
                                                                  float* __restrict__ norms, float* __restrict__ inv_norms,
                                                                  uint32_t* __restrict__ max_norm_bits, int8_t* __restrict__ q8,
                                                                  float* __restrict__ scale, float* __restrict__ vv, float* __restrict__ cosf,
                                                                  uint32_t* __restrict__ err_bits) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l = lane & 15u, rg = lane >> 4;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    float mx_norm = 0.f, mx_err = 0.f, mx_rel = 0.f;
    for (uint64_t rb = wave * 4; rb < n; rb += n_waves * 4) {
        const uint64_t ri = rb + rg;
        const bool live = ri < n;
        const uint64_t r = row0 + (live ? ri : n - 1);  // (a ragged last group re-reads the last row; nothing of it is stored)
        const float* src = corpus + r * (uint64_t)ld + l * 4u;
        v4f x[KJ];
#pragma unroll
        for (int j = 0; j < KJ; j++) x[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + j * 64));
        // ---- the reference's eight chains, and max |v| on the side
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float mx = 0.f;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < KJ; j++) {
            float p[4];
            {
                const f2v a = {x[j][0], x[j][1]}, b = {x[j][2], x[j][3]};
                const f2v pa = a * a, pb = b * b;  // (multiplies only: rounded as the reference's, never fused with the chain's adds)
                p[0] = pa[0];
                p[1] = pa[1];
                p[2] = pb[0];
                p[3] = pb[1];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) mx = __builtin_fmaxf(mx, __builtin_fabsf(x[j][t]));  // (v_max skips NaNs: see `bad` below)
#pragma unroll
            for (int step = 0; step < 8; step++) {
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = dppf<0x122>(acc[t]) + p[t];  // row_ror:2 — the chain moves on to the next lane pair
                // (keep the four chains interleaved: left alone the scheduler runs one chain's eight steps back to back, each behind
                //  an s_nop for the DPP read-after-write hazard)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // lanes 14 (accumulators 0-3) and 15 (4-7) of the row group hold the eight sums: lane 15 adds them left to right from -0.0
        float lo[4];
#pragma unroll
        for (int t = 0; t < 4; t++) lo[t] = dppf<0x111>(acc[t]);  // row_shr:1 — lane 15 sees lane 14's
        float rsum = -0.0f;
#pragma unroll
        for (int t = 0; t < 4; t++) rsum = rsum + lo[t];
#pragma unroll
        for (int t = 0; t < 4; t++) rsum = rsum + acc[t];
        const float mag = dppf<0x15F>(__builtin_sqrtf(rsum));  // row_newbcast:15 — lane 15's (the only meaningful one) to its row group
        // ---- the row's scale and codes
        mx = row16_max(mx);
        // a non-finite element: an infinity is the row's maximum; a NaN went through the sum of squares (non-negative terms: the
        // sum is NaN only if an element is) — one test per row instead of one per element
        bad = !(mx <= 3.0e38f) || !(mag == mag);
        const float sc = (bad || mx == 0.f) ? 0.f : mx / 127.0f;
        const float inv = sc > 0.f ? 127.0f / mx : 0.f;
        float err2 = 0.f;
        float ccf = 0.f;  // c.c of this lane's codes: integers, exact in f32 (<= 128 elements x 127^2 < 2^24 per lane)
        uint32_t* dst = reinterpret_cast<uint32_t*>(q8 + r * (uint64_t)ld) + l;
        // (the kernel is VALU-issue bound beside the chains above: every instruction per element counts — the clamp is one
        //  v_med3, c.c and the error accumulate by explicit FMAs, the four codes of a load are packed by v_cvt_pk_u8_f32 on
        //  c + 128 and flipped to two's complement by one XOR per word)
        // two elements per instruction where the ISA has a packed form (v_pk_mul / v_pk_fma / v_pk_add_f32)
        const f2v inv2 = {inv, inv}, nsc2 = {-sc, -sc}, ok2 = {bad ? 0.f : 1.f, bad ? 0.f : 1.f}, b128 = {128.0f, 128.0f};
        f2v err2v = {0.f, 0.f}, ccv = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; j++) {
            uint32_t pk = 0u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const f2v xv = {x[j][2 * h], x[j][2 * h + 1]};
                f2v cv = xv * inv2;
                cv[0] = __builtin_amdgcn_fmed3f(__builtin_rintf(cv[0]), -127.0f, 127.0f);  // (a NaN comes out as -127: finite)
                cv[1] = __builtin_amdgcn_fmed3f(__builtin_rintf(cv[1]), -127.0f, 127.0f);
                cv = cv * ok2;  // (a non-finite row: scale 0, codes 0)
                const f2v ev = __builtin_elementwise_fma(nsc2, cv, xv);
                err2v = __builtin_elementwise_fma(ev, ev, err2v);
                ccv = __builtin_elementwise_fma(cv, cv, ccv);
                const f2v bv = cv + b128;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(bv[0], (uint32_t)(2 * h), pk);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(bv[1], (uint32_t)(2 * h + 1), pk);
            }
            if (live) __builtin_nontemporal_store(pk ^ 0x80808080u, dst + j * 16);  // (4 bytes per lane, 64 contiguous bytes per row group)
        }
        err2 = err2v[0] + err2v[1];
        ccf = ccv[0] + ccv[1];
        int cc = (int)ccf;
        err2 = row16_add(err2);
        cc = row16_addi(cc);
        if (live && l == 15u) {
            norms[r] = mag;
            inv_norms[r] = mag == 0.0f ? 0.0f : 1.0f / mag;
            scale[r] = sc;
            vv[r] = (sc * sc) * (float)cc;
            cosf[r] = (mag > 0.f && mag <= 3.0e38f) ? sc / mag : 0.f;
            if (mag == mag) mx_norm = __builtin_fmaxf(mx_norm, mag);
            // (q8_err_kernel's rules: slack for the order of the sums; a non-finite row vouches for nothing)
            const float e = __builtin_sqrtf(bad ? __builtin_inff() : err2) * 1.0005f;
            if (e > 0.f) {
                mx_err = __builtin_fmaxf(mx_err, e);
                if (mag > 0.f || !(mag == mag)) {
                    float rel = e / mag * 1.0005f;
                    if (!(rel == rel)) rel = __builtin_inff();
                    mx_rel = __builtin_fmaxf(mx_rel, rel);
                }
            }
        }
    }
    // non-negative floats: bit order == value order
    mx_norm = wave_max(mx_norm);
    mx_err = wave_max(mx_err);
    mx_rel = wave_max(mx_rel);
    if (lane == 0) {
        if (mx_norm > 0.f) atomicMax(max_norm_bits, __float_as_uint(mx_norm));
        if (mx_err > 0.f) atomicMax(err_bits, __float_as_uint(mx_err));
        if (mx_rel > 0.f) atomicMax(err_bits + 1, __float_as_uint(mx_rel));
    }
}

template <int KJ>
hipError_t launch_ingest_q8_kj(const float* corpus, uint32_t ld, uint64_t row0, uint64_t n, float* norms, float* inv_norms, uint32_t* max_norm_bits,
                               int8_t* q8, float* scale, float* vv, float* cosf, uint32_t* err_bits, hipStream_t s) {
    const uint64_t groups = (n + 3) / 4;
    // ~16 waves per CU on a full device; every wave owns the row groups wave, wave + n_waves, ...
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((groups + kQ8Waves - 1) / kQ8Waves, 256ull * 4ull);
    hipLaunchKernelGGL(ingest_q8_kernel<KJ>, dim3(blocks), dim3(kQ8Waves * 64), 0, s, corpus, ld, row0, n, norms, inv_norms, max_norm_bits, q8, scale,
                       vv, cosf, err_bits);
    return hipGetLastError();
}

}  // namespace

// rows the fused kernel takes: no scalar tail in the reference's magnitude, whole 128-element halves, at most 2048 elements
// (the row waits in registers: 4 VGPRs per 64 elements)
bool ingest_q8_supported(uint32_t ld, uint32_t dim) { return dim % 8u == 0 && ld % 128u == 0 && ld >= 128u && ld <= 2048u; }

hipError_t launch_ingest_q8(const float* corpus, uint32_t ld, uint64_t row0, uint64_t n, float* norms, float* inv_norms, uint32_t* max_norm_bits,
                            int8_t* q8, float* scale, float* vv, float* cosf, uint32_t* err_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
#define NMN_Q8_CASE(KJ_) \
    case KJ_: return launch_ingest_q8_kj<KJ_>(corpus, ld, row0, n, norms, inv_norms, max_norm_bits, q8, scale, vv, cosf, err_bits, s);
    switch (ld / 64u) {
        NMN_Q8_CASE(2) NMN_Q8_CASE(4) NMN_Q8_CASE(6) NMN_Q8_CASE(8) NMN_Q8_CASE(10) NMN_Q8_CASE(12) NMN_Q8_CASE(14) NMN_Q8_CASE(16)
        NMN_Q8_CASE(18) NMN_Q8_CASE(20) NMN_Q8_CASE(22) NMN_Q8_CASE(24) NMN_Q8_CASE(26) NMN_Q8_CASE(28) NMN_Q8_CASE(30) NMN_Q8_CASE(32)
        default: return hipErrorInvalidValue;
    }
#undef NMN_Q8_CASE
}

// rows whose layout the one-pass kernel takes: whole reference chunks (no scalar tail) and whole 32-float stages
bool ingest_supported(uint32_t ld, uint32_t dim) { return dim % 8u == 0 && ld % kStageFloats == 0 && ld >= (uint32_t)kStageFloats; }

// magnitudes of rows [row0, row0 + n) (reference order) and, with `half`, their bf16 mirror rows + the mirror's error norms
hipError_t launch_ingest(const float* corpus, uint32_t ld, uint64_t row0, uint64_t n, float* norms, float* inv_norms,
                         uint32_t* max_norm_bits, float* half, uint32_t* err_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t n_tiles = (n + 63) / 64;
    // one workgroup per CU and a few more for the tail; every wave owns tiles gw, gw + n_waves, ...
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_tiles + kWaves - 1) / kWaves, 512);
    const size_t lds = (size_t)kWaves * kWaveLds;
    auto kern = half ? ingest_kernel<true> : ingest_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWaves * 64), lds, s, corpus, ld, row0, n, norms, inv_norms, max_norm_bits, half, err_bits);
    return hipGetLastError();
}

}  // namespace nmn
