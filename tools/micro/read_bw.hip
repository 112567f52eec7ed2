// Pure HBM read-bandwidth probe for MI355X: what a streaming kernel can reach with no compute at all.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw [GiB]
// Variants: loads in flight per lane (U x 16 B), non-temporal or default policy, waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v4f* __restrict__ src, size_t n4, float* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += x[u];
    }
    for (; i < n4; i += stride) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;  // never true: keeps the loads alive
}

// each wave reads CONTIGUOUS 3 KiB rows the way the scan does (16 lanes per row, 4 rows per step)
template <int CH, bool NT>
__global__ __launch_bounds__(256) void read_rows_kernel(const v4f* __restrict__ src, size_t n_rows, uint32_t ld4,
                                                        float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, grp = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t rows_per_wave = (n_rows + n_waves - 1) / n_waves;
    const size_t r0 = wave * rows_per_wave, r1 = r0 + rows_per_wave < n_rows ? r0 + rows_per_wave : n_rows;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t r = r0 + grp; r < r1; r += 4) {
        const v4f* rowp = src + r * ld4;
        for (uint32_t c0 = 0; c0 < ld4; c0 += 16u * CH) {
            v4f x[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) x[c] = NT ? __builtin_nontemporal_load(rowp + c0 + c * 16 + j) : rowp[c0 + c * 16 + j];
#pragma unroll
            for (int c = 0; c < CH; c++) acc += x[c];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}


// the masked sweep's pattern without its bitmap and arithmetic: a sorted list of SELECTED rows (a fraction s of all rows, chosen
// by a hash), each wave walks a contiguous piece of the list, 4 rows per step, 16 lanes per 3 KiB row: what scattered 3 KiB
// rows can stream at.  (selectivity 1.0 = read_rows_kernel through an index list.)
template <int CH, bool NT>
__global__ __launch_bounds__(256) void read_listed_rows_kernel(const v4f* __restrict__ src, const uint32_t* __restrict__ list,
                                                               size_t n_list, uint32_t ld4, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, grp = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (n_list + n_waves - 1) / n_waves;
    const size_t i0 = wave * per, i1 = i0 + per < n_list ? i0 + per : n_list;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = i0 + grp; i < i1; i += 4) {
        const v4f* rowp = src + (size_t)list[i] * ld4;
        for (uint32_t c0 = 0; c0 < ld4; c0 += 16u * CH) {
            v4f x[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) x[c] = NT ? __builtin_nontemporal_load(rowp + c0 + c * 16 + j) : rowp[c0 + c * 16 + j];
#pragma unroll
            for (int c = 0; c < CH; c++) acc += x[c];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

// the access pattern of an MFMA sweep that loads A-fragments straight into registers: workgroup = 4 waves,
// wave w owns the 128-byte k-step w of every 512-byte stage of a row; lane (n = lane & 15, g = lane >> 4) reads
// 16 B at row n of a 16-row block, offset g*16 (hi half) and 64 + g*16 (lo half): 16 rows x 64 B per instruction.
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void frag_read_kernel(const v4f* __restrict__ src, uint32_t n_tiles,
                                                        uint32_t tiles_per_wg, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, n = lane & 15u, g = lane >> 4;
    const uint32_t t0 = blockIdx.x * tiles_per_wg, t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // one "stage" = (tile, kc): 4 row blocks x (hi, lo) = 8 loads per lane
    const uint32_t n_stage = (t1 > t0 ? t1 - t0 : 0) * 6;
    auto addr = [&](uint32_t s, uint32_t rb, uint32_t half) -> const v4f* {
        const uint32_t tile = t0 + s / 6, kc = s % 6;
        return src + ((size_t)(tile * 64u + rb * 16u + n) * 3072u + kc * 512u + w * 128u + half * 64u + g * 16u) / 16u;
    };
    v4f buf[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if ((uint32_t)d < n_stage)
#pragma unroll
            for (int i = 0; i < 8; i++) buf[d][i] = NT ? __builtin_nontemporal_load(addr(d, i >> 1, i & 1)) : *addr(d, i >> 1, i & 1);
    for (uint32_t s0 = 0; s0 < n_stage; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t s = s0 + d;
            if (s < n_stage) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc += buf[d][i];
                const uint32_t ns = s + DEPTH;
                if (ns < n_stage)
#pragma unroll
                    for (int i = 0; i < 8; i++) buf[d][i] = NT ? __builtin_nontemporal_load(addr(ns, i >> 1, i & 1)) : *addr(ns, i >> 1, i & 1);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename L>
static float time_ms(L launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 28.6;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 3072 * 3072;
    void* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    const size_t n4 = bytes / 16;
    const v4f* src = (const v4f*)buf;
    printf("buffer %.2f GB\n", bytes / 1e9);
    for (int waves_per_cu : {8, 16, 32}) {
        const int blocks = 256 * waves_per_cu / 4;
#define RUN(U, NT) { float ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(256), 0, 0, src, n4, sink); }, 5); \
                     printf("grid-stride  waves/CU %2d  U %2d  %s : %7.3f ms  %6.0f GB/s\n", waves_per_cu, U, NT ? "nt " : "def", ms, bytes / ms / 1e6); }
        RUN(4, false) RUN(4, true) RUN(8, true) RUN(12, true) RUN(16, true)
    }
    const size_t n_rows = bytes / 3072;
    for (int waves_per_cu : {8, 16, 32}) {
        const int blocks = 256 * waves_per_cu / 4;
#define RUNR(CH, NT) { float ms = time_ms([&] { hipLaunchKernelGGL((read_rows_kernel<CH, NT>), dim3(blocks), dim3(256), 0, 0, src, n_rows, 192u, sink); }, 5); \
                       printf("row-major    waves/CU %2d  CH %2d  %s : %7.3f ms  %6.0f GB/s\n", waves_per_cu, CH, NT ? "nt " : "def", ms, bytes / ms / 1e6); }
        RUNR(12, true) RUNR(12, false) RUNR(6, true)
    }
    for (double sel : {1.0, 0.5, 0.25, 0.1, 0.05, 0.01}) {
        std::vector<uint32_t> h;
        h.reserve((size_t)(n_rows * sel * 1.05) + 16);
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t x = (r + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
            x ^= x >> 31;
            x *= 0x94D049BB133111EBull;
            x ^= x >> 29;
            if ((double)(x >> 11) * (1.0 / 9007199254740992.0) < sel) h.push_back((uint32_t)r);
        }
        uint32_t* dl;
        CK(hipMalloc((void**)&dl, h.size() * 4 + 16));
        CK(hipMemcpy(dl, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int waves_per_cu : {16, 32}) {
            const int blocks = 256 * waves_per_cu / 4;
#define RUNL(CH, NT) { const size_t nl = h.size(); float ms = time_ms([&] { hipLaunchKernelGGL((read_listed_rows_kernel<CH, NT>), dim3(blocks), dim3(256), 0, 0, src, dl, nl, 192u, sink); }, 5); \
                       printf("listed rows  selectivity %.2f  waves/CU %2d  CH %2d  %s : %7.3f ms  %6.0f GB/s of the rows read\n", sel, waves_per_cu, CH, NT ? "nt " : "def", ms, (double)nl * 3072 / ms / 1e6); }
            RUNL(12, true) RUNL(12, false) RUNL(6, true)
        }
        CK(hipFree(dl));
    }
    const uint32_t n_tiles = (uint32_t)(n_rows / 64);
    for (int wgs : {256, 512, 1024, 2048}) {
        const uint32_t tpw = (n_tiles + wgs - 1) / wgs;
#define RUNF(D, NT) { float ms = time_ms([&] { hipLaunchKernelGGL((frag_read_kernel<D, NT>), dim3(wgs), dim3(256), 0, 0, src, n_tiles, tpw, sink); }, 5); \
                      printf("fragment     wgs %4d  depth %d  %s : %7.3f ms  %6.0f GB/s\n", wgs, D, NT ? "nt " : "def", ms, (double)n_tiles * 64 * 3072 / ms / 1e6); }
        RUNF(3, true) RUNF(6, true) RUNF(6, false) RUNF(12, true)
    }
    return 0;
}
