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
// PROBE: the same ring, waits and barriers with the arithmetic, the LDS reads and every store removed — what the data movement of this
// very kernel reaches on this device (nmn_index_read_probe: bench.py's `ring_only_read_ceiling`; 7.0-7.2 TB/s at 10M x 768).
template <int METRIC, bool PROBE = false>
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
    for (uint32_t c = wave; !PROBE && c * 256u < ld; c += 4u) {
        if (c * 256u + lane * 4u < ld)  // (row strides are multiples of 128 floats: the last KiB may be half)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.qpad + c * 256u + lane * 4u),
                                             (__attribute__((address_space(3))) void*)(qlds + c * 256u), 16, 0, 0);
    }
    const float qmag = PROBE ? 1.f : p.qinfo[0].qmag;

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
            if (!PROBE && kCos && wave == 0 && s0 % KC == 0) norms_dma(t0 + s0 / KC, s0 / KC);
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
            if (!PROBE && kc == 0 && wave == 0 && tile > t0) {
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
            f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
            if constexpr (!PROBE) {
#pragma unroll
                for (int sub = 0; sub < kSub; sub++) {
                    a[sub][0] = *reinterpret_cast<const f4*>(buf + off[sub]);
                    a[sub][1] = *reinterpret_cast<const f4*>(buf + (off[sub] ^ 4u));
                }
                q0 = *reinterpret_cast<const f4*>(qlds + kc * 128u + j * 8u);
                q1 = *reinterpret_cast<const f4*>(qlds + kc * 128u + j * 8u + 4u);
            }
#pragma unroll
            for (int sub = 0; sub < kSub; sub++) {
                // two pieces of the stage ahead per sub-step, in the shadow of the arithmetic
#pragma unroll
                for (int pp = sub * 2; pp < sub * 2 + 2; pp++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nsrc + (loff[pp] & lmask)),
                                                     (__attribute__((address_space(3))) void*)(nbuf + (wave * kRingPieces + (uint32_t)pp) * 256u), 16, 0, 2);
                if constexpr (PROBE) continue;
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
            if (!PROBE && kCos && wave == 0 && ns + 1u < n_stage && nkc == 0) norms_dma(nt, nt - t0);
        }
        if constexpr (PROBE) continue;
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
    if (!PROBE && wave == 0) {
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

// The ring with nothing behind it: the read ceiling of the headline sweep's own data movement (same workgroups, stages, pieces).
hipError_t launch_ring_probe(const float* corpus, uint64_t n_rows, uint32_t ld, uint32_t tiles_per_wg, hipStream_t s) {
    ScanParams p{};
    p.corpus = corpus;
    p.n_rows = n_rows;
    p.ld = ld;
    p.n_tiles = (uint32_t)(n_rows / kTileRows);  // whole tiles only (no row guard in the probe)
    p.tiles_per_wave = std::max<uint32_t>(1, tiles_per_wg);
    if (p.n_tiles == 0) return hipSuccess;
    const uint32_t blocks = (p.n_tiles + p.tiles_per_wave - 1) / p.tiles_per_wave;
    const size_t lds = (size_t)kRingStages * kRingStageBytes + 8 * 64 * 4 + 8 * 4 + (size_t)p.ld * 4;
    auto kern = scan_ring_kernel<NMN_METRIC_DOT_PRODUCT, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, p);
    return hipGetLastError();
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

}  // namespace nmn
