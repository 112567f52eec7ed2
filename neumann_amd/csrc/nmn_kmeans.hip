// nmn_kmeans.hip — device kernels of the k-means that trains an IVF index (tensor_store/src/delta_vector.rs:737-901),
// bit for bit: built with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt like nmn_exact.hip.
//
// KMeans::fit spends its time in two loops: nearest_centroid for every vector (n * k * dim, per iteration) and
// update_centroids.  The first is the exact centroid sweep nmn_ivf.hip already has (exact_scan + first-minimum).
// The second is a strictly sequential f32 sum PER CLUSTER AND DIMENSION in vector order (`*sum += val`,
// delta_vector.rs:876-881) followed by `sum / count as f32`: clusters and dimensions are independent, so one thread
// owns one (cluster, dimension) pair and walks the cluster's members in vector order — same additions, same order.
#include "nmn_internal.h"

#pragma clang fp contract(off)

namespace nmn {

// members[offsets[c] .. offsets[c+1]) = rows of cluster c in ascending row order; new_centroids [k][ld]
__global__ void __launch_bounds__(256) kmeans_update_kernel(const float* __restrict__ corpus, uint32_t ld, uint32_t dim,
                                                            const uint32_t* __restrict__ members,
                                                            const uint64_t* __restrict__ offsets, uint32_t k,
                                                            float* __restrict__ new_centroids) {
    const uint32_t c = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k || j >= ld) return;
    const uint64_t m0 = offsets[c], m1 = offsets[c + 1];
    float sum = 0.0f;  // `vec![0.0f32; dim]`
    if (j < dim)
        for (uint64_t m = m0; m < m1; m++) sum = sum + corpus[(uint64_t)members[m] * ld + j];
    float out = 0.0f;  // empty cluster: `vec![0.0; dim]`
    if (j < dim && m1 > m0) out = sum / (float)(m1 - m0);  // `s / count as f32`
    new_centroids[(uint64_t)c * ld + j] = out;
}

hipError_t launch_kmeans_update(const float* corpus, uint32_t ld, uint32_t dim, const uint32_t* members,
                                const uint64_t* offsets, uint32_t k, float* new_centroids, hipStream_t s) {
    if (k == 0) return hipSuccess;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((ld + 255) / 256, k), dim3(256), 0, s, corpus, ld, dim, members, offsets, k,
                       new_centroids);
    return hipGetLastError();
}

// k-means++ bookkeeping: dist[i] = min(dist[i], d_new[i]) with f32::min semantics (a NaN operand yields the other
// one); d_new arrives as the exact sweep's NEGATED squared distance bits in plain row order.
__global__ void __launch_bounds__(256) kmeans_min_update_kernel(float* __restrict__ dist, const uint32_t* __restrict__ neg_bits,
                                                                uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float d = -u2f(neg_bits[i]);
        const float cur = dist[i];
        dist[i] = __builtin_fminf(cur, d);
    }
}

hipError_t launch_kmeans_min_update(float* dist, const uint32_t* neg_bits, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(kmeans_min_update_kernel, dim3(blocks), dim3(256), 0, s, dist, neg_bits, n);
    return hipGetLastError();
}

}  // namespace nmn
