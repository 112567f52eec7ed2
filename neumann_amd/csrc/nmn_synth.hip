// nmn_synth.hip — counter-based synthetic corpus generator (bench / tests only; not part of the
// reference path).  value(seed,row,col) is integer-hash based with ONE exact int->f32 conversion
// and ONE f32 multiply, so host (nmn_synth_value) and device agree bit for bit and any row can be
// regenerated on the CPU without materialising the 30 GB corpus twice (SURVEY.md §7 hard part g).
#include "nmn_internal.h"

namespace nmn {

__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ inline float synth_value(uint64_t seed, uint64_t row, uint32_t col) {
    const uint64_t h = mix64(mix64(seed ^ (row * 0xD6E8FEB86659FD93ull)) + (uint64_t)col);
    const int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) +
                      (int32_t)(h >> 48) - 131070;
    return (float)s * 0x1.bb685ep-16f;  // f32(1/37837): unit variance for a sum of four u16
}

__global__ void __launch_bounds__(256) synth_fill_kernel(float* __restrict__ corpus, uint32_t ld, uint32_t dim,
                                                         uint64_t seed, uint64_t global_row0,
                                                         uint64_t local_row0, uint64_t n) {
    const uint32_t ld4 = ld >> 2;
    const uint64_t total = n * (uint64_t)ld4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld4;
        const uint32_t c4 = (uint32_t)(i - r * ld4);
        const uint64_t rowkey = mix64(seed ^ ((global_row0 + r) * 0xD6E8FEB86659FD93ull));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t col = c4 * 4u + (uint32_t)e;
            if (col < dim) {
                const uint64_t h = mix64(rowkey + (uint64_t)col);
                const int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                                  (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48) - 131070;
                v[e] = (float)s * 0x1.bb685ep-16f;
            } else {
                v[e] = 0.f;
            }
        }
        float4* dst = reinterpret_cast<float4*>(corpus + (local_row0 + r) * (uint64_t)ld) + c4;
        *dst = make_float4(v[0], v[1], v[2], v[3]);
    }
}

hipError_t launch_synth_fill(float* corpus, uint32_t ld, uint32_t dim, uint64_t seed, uint64_t global_row0,
                             uint64_t local_row0, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_fill_kernel, dim3(256 * 16), dim3(256), 0, s, corpus, ld, dim, seed, global_row0,
                       local_row0, n);
    return hipGetLastError();
}

float synth_value_host(uint64_t seed, uint64_t row, uint32_t col) { return synth_value(seed, row, col); }

}  // namespace nmn
