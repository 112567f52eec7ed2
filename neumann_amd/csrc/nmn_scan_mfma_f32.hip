// nmn_scan_mfma_f32.hip — the batched-query sweep on the matrix cores over the ROW-MAJOR F32 CORPUS (BASELINE config 3 as
// SURVEY §8(d) prices it: rows * dim * 4 bytes per batch of <= 64-128 queries; SURVEY §7 hard part b: operands rounded to bf16
// in registers, HBM bytes unchanged).  Shape dispatch of scan_mfma_kernel<..., F32 = true> (nmn_scan_mfma_kernel.h).  What it
// replaces: 16 VALU sweeps of four queries each for a 64-query batch on a shard without a mirror (a shard short of HBM, or
// nmn_index_set_mirror(idx, 0)).  Reference loop: vector_engine/src/lib.rs:2049-2101, 2115-2228.
#include "nmn_scan_mfma_kernel.h"

namespace nmn {

// stages of 32 KiB ([64 rows][128 f32]: KS = 2), KC = ld / 128 stages per row
template <int KC, int METRIC, int QG>
static hipError_t launch_kc_f32(const ScanParams& p, hipStream_t s) {
#ifdef NMN_MFMA_F32_KS1  // measurement build: 16-KiB stages ([64 rows][64 f32]), a ring of eight (rows of <= 1536 elements with 64 queries)
    if constexpr (QG == 4 && KC <= 12)
        return (p.mask || p.qmasks) ? launch_one_mfma<2 * KC, 1, QG, METRIC, true, 4, false, true>(p, s)
                                    : launch_one_mfma<2 * KC, 1, QG, METRIC, false, 4, false, true>(p, s);
#endif
    return (p.mask || p.qmasks) ? launch_one_mfma<KC, 2, QG, METRIC, true, 4, false, true>(p, s)
                                : launch_one_mfma<KC, 2, QG, METRIC, false, 4, false, true>(p, s);
}

template <int METRIC>
static hipError_t launch_metric_f32(const ScanParams& p, hipStream_t s) {
    const uint32_t kc = p.ld / kStageK;
    if (p.nq > 64) {  // 128 stationary queries per workgroup where their fragments fit (two query groups per wave)
        switch (kc) {
            case 1: return launch_kc_f32<1, METRIC, 8>(p, s);
            case 2: return launch_kc_f32<2, METRIC, 8>(p, s);
            case 3: return launch_kc_f32<3, METRIC, 8>(p, s);
            case 4: return launch_kc_f32<4, METRIC, 8>(p, s);
            case 5: return launch_kc_f32<5, METRIC, 8>(p, s);
            case 6: return launch_kc_f32<6, METRIC, 8>(p, s);
            default: break;
        }
    }
    switch (kc) {
        case 1: return launch_kc_f32<1, METRIC, 4>(p, s);    // 128
        case 2: return launch_kc_f32<2, METRIC, 4>(p, s);    // 256
        case 3: return launch_kc_f32<3, METRIC, 4>(p, s);    // 384
        case 4: return launch_kc_f32<4, METRIC, 4>(p, s);    // 512
        case 5: return launch_kc_f32<5, METRIC, 4>(p, s);    // 640
        case 6: return launch_kc_f32<6, METRIC, 4>(p, s);    // 768
        case 8: return launch_kc_f32<8, METRIC, 4>(p, s);    // 1024
        case 10: return launch_kc_f32<10, METRIC, 4>(p, s);  // 1280
        case 12: return launch_kc_f32<12, METRIC, 4>(p, s);  // 1536
        case 16: return launch_kc_f32<16, METRIC, 2>(p, s);  // 2048: 32 stationary queries, K-halves on wave pairs
        case 24: return launch_kc_f32<24, METRIC, 2>(p, s);  // 3072
        case 32: return launch_kc_f32<32, METRIC, 2>(p, s);  // 4096
        default: return hipErrorInvalidValue;
    }
}

// p.corpus_half == p.corpus_i8 == nullptr; the shapes are scan_mfma_supported()'s
hipError_t launch_scan_mfma_f32(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_metric_f32<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_EUCLIDEAN: return launch_metric_f32<NMN_METRIC_EUCLIDEAN>(p, s);
        case kMetricNegL2: return launch_metric_f32<kMetricNegL2>(p, s);
        default: return launch_metric_f32<NMN_METRIC_DOT_PRODUCT>(p, s);
    }
}

}  // namespace nmn
