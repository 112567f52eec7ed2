// Does a pure streaming read depend on WHERE the driver put the buffer?  Allocates the same 15.36 GB buffer again and again
// (with filler allocations of varying size in between, kept or freed), times a plain non-temporal read of all of it, and
// prints the virtual address next to the time.  Optional: the buffer carved at an offset out of a larger allocation.
//
//   hipcc --offload-arch=gfx950 -O3 -o placement tools/micro/placement.hip && ./placement [GB=15.36] [builds=10] [offset_MiB=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// every workgroup streams a contiguous range (as the sweeps do), 16 B per lane and load, 8 loads in flight per lane
__global__ void __launch_bounds__(256) read_kernel(const v4f* __restrict__ buf, uint64_t n_vec, uint64_t per_wg, float* sink) {
    const uint64_t b0 = (uint64_t)blockIdx.x * per_wg, b1 = std::min(b0 + per_wg, n_vec);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    uint64_t i = b0 + threadIdx.x;
    for (; i + 7 * 256 < b1; i += 8 * 256) {
        v4f x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = __builtin_nontemporal_load(buf + i + (uint64_t)u * 256);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += x[u];
    }
    for (; i < b1; i += 256) acc += __builtin_nontemporal_load(buf + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-30f) *sink = 1.f;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 15.36;
    const int builds = argc > 2 ? atoi(argv[2]) : 10;
    const size_t offset = (argc > 3 ? (size_t)atol(argv[3]) : 0) << 20;
    const size_t bytes = ((size_t)(gb * 1e9) / 4096) * 4096;
    const uint64_t n_vec = bytes / 16;
    float* sink;
    CK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<void*> fillers;
    for (int b = 0; b < builds; b++) {
        // a filler of varying size before the buffer moves where the buffer lands; odd builds keep it, even builds free it first
        void* filler = nullptr;
        const size_t fsz = ((size_t)(b * 37 % 11) * 333u + 17u) << 20;
        CK(hipMalloc(&filler, fsz));
        if (b % 2 == 0) { CK(hipFree(filler)); filler = nullptr; }
        char* raw;
        CK(hipMalloc((void**)&raw, bytes + offset));
        CK(hipMemsetAsync(raw, 0, bytes + offset, 0));
        const v4f* buf = reinterpret_cast<const v4f*>(raw + offset);
        const int wgs = 1024 * 2;
        const uint64_t per_wg = ((n_vec + wgs - 1) / wgs + 255) / 256 * 256;
        std::vector<float> t;
        for (int r = 0; r < 12; r++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(read_kernel, dim3(wgs), dim3(256), 0, 0, buf, n_vec, per_wg, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("build %2d  va %p  (mod 1 GiB: %4zu MiB, mod 2 MiB: %4zu KiB)  min %.3f  med %.3f  max %.3f ms   %5.0f GB/s\n", b, (void*)buf,
               ((size_t)buf & ((1ull << 30) - 1)) >> 20, ((size_t)buf & ((1ull << 21) - 1)) >> 10, t[0], t[t.size() / 2], t.back(),
               bytes / t[t.size() / 2] / 1e6);
        CK(hipFree(raw));
        if (filler) fillers.push_back(filler);
    }
    for (void* f : fillers) (void)hipFree(f);
    return 0;
}
