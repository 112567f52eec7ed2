// Does a CU-masked stream keep CUs free for small dependent kernels while a machine-filling sweep runs?  (hipExtStreamCreateWithCUMask)
// hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask.hip -o /tmp/cu_mask && /tmp/cu_mask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void big(float* p, int iters) {  // machine-filling, long-lived waves
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 123.f) p[0] = a;
}
__global__ void __launch_bounds__(1024) small(float* p) {  // one big workgroup with a lot of LDS, like select_kernel
    extern __shared__ float l[];
    l[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    if (l[(threadIdx.x + 1) & 1023] == 77.f) p[1] = 1.f;
}
int main() {
    int ncu = 0;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    ncu = pr.multiProcessorCount;
    printf("CUs %d\n", ncu);
    float* d;
    CK(hipMalloc(&d, 1 << 20));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(small), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int reserve : {0, 8, 16, 32}) {
        hipStream_t a, b;
        if (reserve == 0) {
            CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        } else {
            const int words = (ncu + 31) / 32;
            std::vector<uint32_t> ma(words, 0), mb(words, 0);
            for (int c = 0; c < ncu; c++) {
                // reserve every (ncu / reserve)-th CU for stream b
                const bool r = (c % (ncu / reserve)) == 0;
                (r ? mb : ma)[c / 32] |= 1u << (c % 32);
            }
            CK(hipExtStreamCreateWithCUMask(&a, words, ma.data()));
            CK(hipExtStreamCreateWithCUMask(&b, words, mb.data()));
        }
        // calibrate: big alone
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(big, dim3(1024), dim3(256), 0, a, d, 40000);
        CK(hipStreamSynchronize(a));
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 10; it++) hipLaunchKernelGGL(big, dim3(1024), dim3(256), 0, a, d, 40000);
        CK(hipStreamSynchronize(a));
        const double big_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
        // chain of 8 small kernels on b alone
        for (int it = 0; it < 8; it++) hipLaunchKernelGGL(small, dim3(1), dim3(1024), 150 * 1024, b, d);
        CK(hipStreamSynchronize(b));
        t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 80; it++) hipLaunchKernelGGL(small, dim3(1), dim3(1024), 150 * 1024, b, d);
        CK(hipStreamSynchronize(b));
        const double small_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 80;
        // both: 10 bigs on a, meanwhile chains of 8 smalls on b; measure how many small chains complete while a runs
        t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 10; it++) hipLaunchKernelGGL(big, dim3(1024), dim3(256), 0, a, d, 40000);
        int chains = 0;
        while (hipStreamQuery(a) == hipErrorNotReady) {
            for (int it = 0; it < 8; it++) hipLaunchKernelGGL(small, dim3(1), dim3(1024), 150 * 1024, b, d);
            CK(hipStreamSynchronize(b));
            chains++;
        }
        const double both_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("reserve %2d CUs: big alone %.1f us each; small alone %.2f us each; 10 bigs took %.1f us with %d chains of 8 smalls alongside (%.1f us per chain)\n",
               reserve, big_us, small_us, both_us, chains, chains ? both_us / chains : 0.0);
        (void)hipStreamDestroy(a);
        (void)hipStreamDestroy(b);
    }
    return 0;
}
