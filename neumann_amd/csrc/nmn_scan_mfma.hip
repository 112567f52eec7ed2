// nmn_scan_mfma.hip — the batched-query sweep on the matrix cores over the shard's MIRRORS (bf16, 8-bit): shape dispatch of
// scan_mfma_kernel (nmn_scan_mfma_kernel.h, which also describes the kernel).  The f32-rows form: nmn_scan_mfma_f32.hip.
#include "nmn_scan_mfma_kernel.h"

namespace nmn {

template <int KC, int KS, int QG, int METRIC, int WAVES = 4>
static hipError_t launch_kc(const ScanParams& p, hipStream_t s) {
    return (p.mask || p.qmasks) ? launch_one_mfma<KC, KS, QG, METRIC, true, WAVES>(p, s)
                                : launch_one_mfma<KC, KS, QG, METRIC, false, WAVES>(p, s);
}

// row length / 128: rows that are a multiple of 256 elements stream in 32-KiB stages (KS = 2), the others in 16-KiB ones
template <int METRIC>
static hipError_t launch_metric(const ScanParams& p, hipStream_t s) {
    // more than 64 queries in the pass and rows of <= 768, 1024 or 1280 elements: 128 stationary queries per workgroup
    // (the matrix cores are ~16 % busy at 64).  B-fragments: 192 VGPRs at 768, 256 / 320 at 1024 / 1280 — there the compiler
    // spills in the one-time fragment build, not in the loop (5M x 1024: two 64-query sweeps 3.43 ms, one of 128 2.18 ms).
    // 1536 would need 384 for the fragments alone.
    // NMN_MFMA_WAVES=8 (measurement knob): 8-wave workgroups, two waves per SIMD at <= 256 registers each — 64 queries as 4
    // groups x K-halves on wave pairs, 128 as 8 groups.  Built for 768-element rows only: same answers, and on 10M x 768 the
    // same time as one wave per SIMD (2.59-2.63 vs 2.53-2.65 ms in one run): the sweep is not waiting on per-wave stalls.
    static const bool waves8 = [] {
        const char* e = getenv("NMN_MFMA_WAVES");
        return e ? atoi(e) == 8 : false;
    }();
    if (waves8) {
        if (p.nq > 64) {
            switch (p.ld / kStageK) {
                case 6: return launch_kc<3, 2, 8, METRIC, 8>(p, s);
                default: break;
            }
        } else {
            switch (p.ld / kStageK) {
                case 6: return launch_kc<3, 2, 4, METRIC, 8>(p, s);
                default: break;
            }
        }
    }
    if (p.nq > 64) {
        switch (p.ld / kStageK) {
            case 1: return launch_kc<1, 1, 8, METRIC>(p, s);
            case 2: return launch_kc<1, 2, 8, METRIC>(p, s);
            case 3: return launch_kc<3, 1, 8, METRIC>(p, s);
            case 4: return launch_kc<2, 2, 8, METRIC>(p, s);
            case 5: return launch_kc<5, 1, 8, METRIC>(p, s);
#ifdef NMN_MFMA_768_KS1
            case 6: return launch_kc<6, 1, 8, METRIC>(p, s);  // 16-KiB stages (measurement build)
#else
            case 6: return launch_kc<3, 2, 8, METRIC>(p, s);
#endif
            case 8: return launch_kc<4, 2, 8, METRIC>(p, s);
            case 10: return launch_kc<5, 2, 8, METRIC>(p, s);
            default: break;
        }
    }
    switch (p.ld / kStageK) {
        case 1: return launch_kc<1, 1, 4, METRIC>(p, s);   // 128
        case 2: return launch_kc<1, 2, 4, METRIC>(p, s);   // 256
        case 3: return launch_kc<3, 1, 4, METRIC>(p, s);   // 384
        case 4: return launch_kc<2, 2, 4, METRIC>(p, s);   // 512
        case 5: return launch_kc<5, 1, 4, METRIC>(p, s);   // 640
#ifdef NMN_MFMA_768_KS1
        case 6: return launch_kc<6, 1, 4, METRIC>(p, s);   // 768 in 16-KiB stages (measurement build)
#else
#ifdef NMN_MFMA_768_KS1
        case 6: return launch_kc<6, 1, 4, METRIC>(p, s);   // 768 in 16-KiB stages (measurement build)
#else
        case 6: return launch_kc<3, 2, 4, METRIC>(p, s);   // 768
#endif
#endif
        case 8: return launch_kc<4, 2, 4, METRIC>(p, s);   // 1024
        case 10: return launch_kc<5, 2, 4, METRIC>(p, s);  // 1280
        case 12: return launch_kc<6, 2, 4, METRIC>(p, s);  // 1536
        case 16: return launch_kc<8, 2, 2, METRIC>(p, s);  // 2048: 32 stationary queries, K-halves on wave pairs
        case 24: return launch_kc<12, 2, 2, METRIC>(p, s); // 3072
        case 32: return launch_kc<32, 1, 2, METRIC>(p, s); // 4096 (256 VGPRs of B-fragments: the compiler spills ~100 registers,
                                                           // in the one-time fragment build only — 5.7 TB/s measured)
        default: return hipErrorInvalidValue;
    }
}

// The 8-bit mirror on the matrix cores: 64 stationary queries per workgroup (more queries: several query blocks of one
// launch, folded onto one XCD like the long rows of the bf16 sweep), rows of whole 256-element groups up to 1536, and 2048 / 3072 with
// 32 stationary queries per workgroup (K-halves on wave pairs); with one bitmap
// for the batch or one per query (the epilogue's business: the sweep reads every row either way).
template <int KC, int KS, int METRIC, int QG = 4>
static hipError_t launch_kc_i8(const ScanParams& p, hipStream_t s) {
    return (p.mask || p.qmasks) ? launch_one_mfma<KC, KS, QG, METRIC, true, 4, true>(p, s)
                                : launch_one_mfma<KC, KS, QG, METRIC, false, 4, true>(p, s);
}
template <int METRIC>
static hipError_t launch_metric_i8(const ScanParams& p, hipStream_t s) {
    static const bool waves8 = [] {  // measurement knob (NMN_MFMA_WAVES=8): two waves per SIMD, query groups on wave pairs that split K
        const char* e = getenv("NMN_MFMA_WAVES");
        return e ? atoi(e) == 8 : false;
    }();
    if (waves8 && p.ld == 768u && !p.mask && !p.qmasks) return launch_one_mfma<3, 1, 4, METRIC, false, 8, true>(p, s);
    // more than 64 queries in the pass and rows of <= 768 elements: 128 stationary queries per workgroup, two query groups per
    // wave (192 VGPRs of h / l fragments at 768, as the bf16 form) — one pass over the mirror instead of two folded query blocks
    static const bool no_qg8 = getenv("NMN_MFMA_I8_NO_128") != nullptr;  // (A/B switch)
    if (p.nq > 64 && !no_qg8) {
        switch (p.ld / 256u) {
            case 1: return launch_kc_i8<1, 1, METRIC, 8>(p, s);
            case 2: return launch_kc_i8<1, 2, METRIC, 8>(p, s);
            case 3: return launch_kc_i8<3, 1, METRIC, 8>(p, s);
            default: break;
        }
    }
    switch (p.ld / 256u) {  // = row bytes / 256: the unit the bf16 launcher calls ld / kStageK
        case 1: return launch_kc_i8<1, 1, METRIC>(p, s);   // 256
        case 2: return launch_kc_i8<1, 2, METRIC>(p, s);   // 512
        case 3: return launch_kc_i8<3, 1, METRIC>(p, s);   // 768
        case 4: return launch_kc_i8<2, 2, METRIC>(p, s);   // 1024
        case 5: return launch_kc_i8<5, 1, METRIC>(p, s);   // 1280
        case 6: return launch_kc_i8<3, 2, METRIC>(p, s);   // 1536
        case 8: return launch_kc_i8<4, 2, METRIC, 2>(p, s);   // 2048: 32 stationary queries, K-halves on wave pairs (as the bf16 form)
        case 12: return launch_kc_i8<6, 2, METRIC, 2>(p, s);  // 3072
        default: return hipErrorInvalidValue;
    }
}
bool scan_mfma_i8_supported(uint32_t ld, uint32_t dim, int metric) {
    if (!(metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2))
        return false;
    // (3072 under a Euclidean metric stays on the bf16 mirror: on isotropic rows the 8-bit margin is ~1 sigma of the score
    //  spread there — 5 000+ candidates per query, the crowd path, no gain: 2M x 3072, 64 queries 3.35 vs 3.37 ms)
    if (ld == 3072u && (metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2)) return false;
    return dim <= ld && ld % 256u == 0 && ld >= 256u && (ld <= 1536u || ld == 2048u || ld == 3072u);
}

// Can the MFMA sweep serve this shape?  Cosine / dot / Euclidean, row length a multiple of 128 floats: up to 768, or 1024 / 1280 /
// 1536 with 64 stationary queries per workgroup (their bf16 B-fragments take up to 192 VGPRs at 1536), or 2048 / 3072 / 4096 with 32.
bool scan_mfma_supported(uint32_t ld, uint32_t dim, int metric) {
    if (!(metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT || metric == NMN_METRIC_EUCLIDEAN ||
          metric == kMetricNegL2) ||
        dim > ld || ld % kStageK != 0)  // (ld > dim: zero padding up to the next multiple of 128, nmn_index_create)
        return false;
    const uint32_t kc = ld / kStageK;
    return (kc >= 1 && kc <= 6) || kc == 8 || kc == 10 || kc == 12 || kc == 16 || kc == 24 || kc == 32;
}

// p.tiles_per_wave = tiles per WORKGROUP; wmax is indexed by workgroup.
hipError_t launch_scan_mfma(const ScanParams& p, hipStream_t s) {
    if (!p.corpus_i8 && !p.corpus_half) return launch_scan_mfma_f32(p, s);  // no mirror: the f32 rows themselves (nmn_scan_mfma_f32.hip)
    if (p.corpus_i8 && p.i8_one_plane) return launch_scan_mfma_i8_one(p, s);  // one query plane (nmn_scan_mfma_i8x.hip)
    if (p.corpus_i8) {  // the 8-bit mirror
        switch (p.metric) {
            case NMN_METRIC_COSINE: return launch_metric_i8<NMN_METRIC_COSINE>(p, s);
            case NMN_METRIC_EUCLIDEAN: return launch_metric_i8<NMN_METRIC_EUCLIDEAN>(p, s);
            case kMetricNegL2: return launch_metric_i8<kMetricNegL2>(p, s);
            default: return launch_metric_i8<NMN_METRIC_DOT_PRODUCT>(p, s);
        }
    }
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_metric<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN: return launch_metric<NMN_METRIC_EUCLIDEAN>(p, s);
        case kMetricNegL2: return launch_metric<kMetricNegL2>(p, s);
        default: return launch_metric<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

}  // namespace nmn
