// Does a tile read TWICE by the same wave (pass A over the whole tile, then pass B over it again) come from a cache the second time?
// What a two-pass-per-tile ingest (magnitudes + row maxima first, int8 codes second) would rely on.
//   hipcc --offload-arch=gfx950 -O3 -o reread reread.hip && ./reread [rows_per_tile] [waves_per_cu] [nt_first]
// Every wave owns tiles of R rows x 3 KiB (768 f32); reads each tile once (ONCE=1) or twice; 12 x 16 B in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool TWICE, bool NT1>
__global__ __launch_bounds__(256) void reread_kernel(const v4f* __restrict__ src, size_t n_tiles, uint32_t R, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, grp = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t ld4 = 192;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t t = wave; t < n_tiles; t += n_waves) {
        const v4f* tile = src + t * (size_t)R * ld4;
        for (int pass = 0; pass < (TWICE ? 2 : 1); pass++) {
            for (uint32_t r = grp; r < R; r += 4) {
                const v4f* rowp = tile + (size_t)r * ld4;
                v4f x[12];
#pragma unroll
                for (int c = 0; c < 12; c++) x[c] = (NT1 && pass == 0) ? __builtin_nontemporal_load(rowp + c * 16 + j) : rowp[c * 16 + j];
#pragma unroll
                for (int c = 0; c < 12; c++) acc += x[c];
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
    const uint32_t R = argc > 1 ? atoi(argv[1]) : 64;
    const int wpc = argc > 2 ? atoi(argv[2]) : 4;
    const size_t rows = 10000000 / R * R;
    const size_t bytes = rows * 3072;
    v4f* d;
    float* sink;
    hipMalloc(&d, bytes);
    hipMalloc(&sink, 4);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * wpc / 4;
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((reread_kernel<false, true>), dim3(blocks), dim3(256), 0, 0, d, rows / R, R, sink);
            if (mode == 1) hipLaunchKernelGGL((reread_kernel<true, false>), dim3(blocks), dim3(256), 0, 0, d, rows / R, R, sink);
            if (mode == 2) hipLaunchKernelGGL((reread_kernel<true, true>), dim3(blocks), dim3(256), 0, 0, d, rows / R, R, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("R=%u rows/tile (%u KiB), %d waves/CU, %s: %.3f ms  (%.2f TB/s of the corpus bytes once)\n", R, R * 3, wpc,
               mode == 0 ? "read once (nt)" : mode == 1 ? "read twice (default policy)" : "read twice (first pass nt)", best, bytes / best / 1e9);
    }
    return 0;
}
