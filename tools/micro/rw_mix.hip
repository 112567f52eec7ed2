// What can a streaming kernel reach that READS 4 bytes and WRITES 1 byte per element (ingest_q8_kernel's mix: f32 rows in, int8 codes
// out) with no arithmetic at all?  hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip && ./rw_mix
// Each lane: U x 16-byte non-temporal loads (64 B x U per lane), U/4 x 16-byte non-temporal stores; grid-stride over 30.72 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int U, bool WRITE>
__global__ __launch_bounds__(256) void rw_kernel(const v4f* __restrict__ src, u4* __restrict__ dst, size_t n4) {
    // a wave takes blocks of U x 64 float4 (U KiB): load u covers 1 KiB contiguous, store w covers 1 KiB contiguous
    const size_t lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t b = wave; b * U * 64 < n4; b += n_waves) {
        const size_t i = b * 64 + lane;  // (index arithmetic below: block b, slot u -> float4 (b * U + u) * 64 + lane)
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = __builtin_nontemporal_load(src + (b * U + u) * 64 + lane);
#pragma unroll
        for (int w = 0; w < U / 4; w++) {
            u4 o;
            o[0] = __float_as_uint(x[4 * w][0]) ^ __float_as_uint(x[4 * w][1]);
            o[1] = __float_as_uint(x[4 * w + 1][0]) ^ __float_as_uint(x[4 * w + 1][2]);
            o[2] = __float_as_uint(x[4 * w + 2][1]) ^ __float_as_uint(x[4 * w + 2][3]);
            o[3] = __float_as_uint(x[4 * w + 3][0]) ^ __float_as_uint(x[4 * w + 3][3]);
            (void)i;
            if (WRITE) __builtin_nontemporal_store(o, dst + (b * (U / 4) + w) * 64 + lane);
            else if ((o[0] ^ o[1] ^ o[2] ^ o[3]) == 0x12345678u) dst[0] = o;  // (never: keeps every load alive)
        }
    }
}

int main() {
    const size_t bytes = 10000000ull * 768 * 4, n4 = bytes / 16;
    v4f* d;
    u4* o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, bytes / 4);
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpc : {4, 8, 16}) {
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9;
            for (int it = 0; it < 5; it++) {
                hipEventRecord(e0);
                const int blocks = 256 * wpc / 4;
                if (mode == 0) hipLaunchKernelGGL((rw_kernel<12, false>), dim3(blocks), dim3(256), 0, 0, d, o, n4);
                else hipLaunchKernelGGL((rw_kernel<12, true>), dim3(blocks), dim3(256), 0, 0, d, o, n4);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double gb = mode ? bytes * 1.25 / 1e9 : bytes / 1e9;
            printf("%2d waves/CU, %s: %.3f ms  -> %.2f TB/s over %.1f GB = %.3f of 8 TB/s\n", wpc, mode ? "read 30.72 GB + write 7.68 GB" : "read 30.72 GB only          ",
                   best, gb / best, gb, gb / best / 8.0);
        }
    }
    return 0;
}
