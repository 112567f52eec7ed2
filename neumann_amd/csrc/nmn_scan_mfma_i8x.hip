// nmn_scan_mfma_i8x.hip — the 8-bit matrix-core sweep with ONE query plane (scan_mfma_kernel<..., I8, ONE = true>, described in
// nmn_scan_mfma_kernel.h): cosine and dot-product batches, the shapes of nmn_scan_mfma.hip's 8-bit dispatch.  A translation unit of
// its own so that its instantiations compile beside the others.
#include "nmn_scan_mfma_kernel.h"

namespace nmn {

template <int KC, int KS, int METRIC, int QG = 4>
static hipError_t launch_kc_i8_one(const ScanParams& p, hipStream_t s) {
    return (p.mask || p.qmasks) ? launch_one_mfma<KC, KS, QG, METRIC, true, 4, true, false, true>(p, s)
                                : launch_one_mfma<KC, KS, QG, METRIC, false, 4, true, false, true>(p, s);
}

template <int METRIC>
static hipError_t launch_metric_i8_one(const ScanParams& p, hipStream_t s) {
    static const bool no_qg8 = getenv("NMN_MFMA_I8_NO_128") != nullptr;  // (A/B switch, as in nmn_scan_mfma.hip)
    if (p.nq > 64 && !no_qg8) {
        switch (p.ld / 256u) {
            case 1: return launch_kc_i8_one<1, 1, METRIC, 8>(p, s);
            case 2: return launch_kc_i8_one<1, 2, METRIC, 8>(p, s);
            case 3: return launch_kc_i8_one<3, 1, METRIC, 8>(p, s);
            default: break;
        }
    }
    switch (p.ld / 256u) {
        case 1: return launch_kc_i8_one<1, 1, METRIC>(p, s);   // 256
        case 2: return launch_kc_i8_one<1, 2, METRIC>(p, s);   // 512
        case 3: return launch_kc_i8_one<3, 1, METRIC>(p, s);   // 768
        case 4: return launch_kc_i8_one<2, 2, METRIC>(p, s);   // 1024
        case 5: return launch_kc_i8_one<5, 1, METRIC>(p, s);   // 1280
        case 6: return launch_kc_i8_one<3, 2, METRIC>(p, s);   // 1536
        case 8: return launch_kc_i8_one<4, 2, METRIC, 2>(p, s);   // 2048
        case 12: return launch_kc_i8_one<6, 2, METRIC, 2>(p, s);  // 3072
        default: return hipErrorInvalidValue;
    }
}

bool scan_mfma_i8_one_plane_supported(int metric) { return metric == NMN_METRIC_COSINE || metric == NMN_METRIC_DOT_PRODUCT; }

hipError_t launch_scan_mfma_i8_one(const ScanParams& p, hipStream_t s) {
    switch (p.metric) {
        case NMN_METRIC_COSINE: return launch_metric_i8_one<NMN_METRIC_COSINE>(p, s);
        case NMN_METRIC_DOT_PRODUCT: return launch_metric_i8_one<NMN_METRIC_DOT_PRODUCT>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nmn
