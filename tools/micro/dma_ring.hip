// LDS-DMA ring probe for MI355X: how fast can ONE workgroup per CU stream a bf16 mirror through LDS with
// global_load_lds_dwordx4, as the matrix-core sweep (nmn_scan_mfma.hip) does, with no MFMA work at all?
//   hipcc --offload-arch=gfx950 -O3 -o dma_ring dma_ring.hip && ./dma_ring
// Variants: SEG  = stage is [64 rows][SEGB bytes] (one 512-B / 256-B segment of each of 64 rows: the round-1 layout)
//           FULL = stage is ROWS consecutive whole rows = one contiguous block of the mirror
// each with its ring depth, temporal policy (aux 0 / 2 = nt) and number of workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if constexpr (N == 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else static_assert(N < 0, "add the immediate");
}

// STAGE bytes per stage, RING stages, WAVES per workgroup.  SEGB > 0: stage = [STAGE/SEGB rows][SEGB bytes] of rows with pitch
// `pitch` (kc-th segment of each row); SEGB == 0: stage = STAGE contiguous bytes.
typedef short s8v __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// WORK: 0 = every wave reads 4 x 16 B per lane of the stage (the ring alone); 1 = every wave reads the WHOLE stage
// (STAGE/1024 ds_read_b128 per wave: the query-split layout of nmn_scan_mfma.hip); 2 = + one v_mfma_f32_16x16x32_bf16 per read;
// 3 = + a per-tile epilogue like the sweep's (16 rcp / max per lane every 3 stages, one 16-byte store per lane and tile)
template <int STAGE, int RING, int AUX, int SEGB, int WAVES, int WORK = 0>
__global__ void __launch_bounds__(WAVES * 64, 1) ring_kernel(const char* __restrict__ src, uint64_t wg_bytes, uint32_t pitch,
                                                             float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PIECES = STAGE / 1024 / WAVES;  // 1-KiB DMA instructions per wave and stage
    static_assert(PIECES * 1024 * WAVES == STAGE, "stage splits into whole pieces");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (uint64_t)blockIdx.x * wg_bytes;
    const uint32_t n_stage = (uint32_t)(wg_bytes / STAGE);
    uint32_t loff[PIECES];
#pragma unroll
    for (int pp = 0; pp < PIECES; pp++) {
        const uint32_t b = (wave * PIECES + pp) * 1024u + lane * 16u;  // byte inside the stage image
        if constexpr (SEGB > 0) {
            const uint32_t r = b / SEGB, c = b % SEGB;
            loff[pp] = r * pitch + (((c / 16u) ^ (r & 15u)) * 16u);
        } else {
            loff[pp] = b;
        }
    }
    auto stage_src = [&](uint32_t s) -> const char* {
        if constexpr (SEGB > 0) {
            const uint32_t kcs = pitch / SEGB;  // stages per tile of STAGE/SEGB rows
            return base + (uint64_t)(s / kcs) * (STAGE / SEGB) * pitch + (s % kcs) * SEGB;
        } else {
            return base + (uint64_t)s * STAGE;
        }
    };
    auto issue = [&](uint32_t s) {
#pragma unroll
        for (int pp = 0; pp < PIECES; pp++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stage_src(s) + loff[pp]),
                                             (__attribute__((address_space(3))) void*)(lds + (s % RING) * (STAGE / 4) + (wave * PIECES + pp) * 256u),
                                             16, 0, AUX);
    };
#pragma unroll
    for (uint32_t s = 0; s < RING - 1; s++)
        if (s < n_stage) issue(s);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    v4f macc[4] = {acc, acc, acc, acc};
    s8v bfrag[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bfrag[i] = (s8v){(short)(lane + i), 1, 2, 3, 4, 5, 6, (short)i};
    for (uint32_t s = 0; s < n_stage; s++) {
        const uint32_t after = n_stage - 1u - s < (uint32_t)(RING - 2) ? n_stage - 1u - s : (uint32_t)(RING - 2);
        if (after == RING - 2) wait_vm<(RING - 2) * PIECES>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + RING - 1 < n_stage) issue(s + RING - 1);
        const float* buf = lds + (s % RING) * (STAGE / 4);
        if constexpr (WORK == 0) {
            // consume: every wave reads a quarter of the stage (4 ds_read_b128)
#pragma unroll
            for (int i = 0; i < 4; i++) acc += *reinterpret_cast<const v4f*>(buf + ((wave * 4 + i) * 64u + lane) * 4u);
        } else {
            constexpr int NR = STAGE / 1024;
            u4v a[NR];
#pragma unroll
            for (int i = 0; i < NR; i++) a[i] = *reinterpret_cast<const u4v*>(buf + ((uint32_t)i * 64u + (lane ^ (uint32_t)(i & 15))) * 4u);
            if constexpr (WORK == 1) {
#pragma unroll
                for (int i = 0; i < NR; i++) acc[0] += __uint_as_float(a[i][0] ^ a[i][3]);
            } else {
#pragma unroll
                for (int i = 0; i < NR; i++)
                    macc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s8v, a[i]), bfrag[i % 8], macc[i & 3], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (WORK == 3) {
            if (s % 3 == 2) {  // "tile" epilogue
                float m = -1e30f;
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float sc = macc[r][e] * __builtin_amdgcn_rcpf(1.0f + __builtin_fabsf(macc[r][e]));
                        m = __builtin_fmaxf(m, sc);
                        macc[r][e] = 0.f;
                    }
                m = __builtin_fmaxf(m, __shfl_xor(m, 16));
                m = __builtin_fmaxf(m, __shfl_xor(m, 32));
                if ((lane >> 4) == 0) sink[64 + ((blockIdx.x * WAVES + wave) * 16u + (lane & 15u))] = m;
            }
        }
    }
    if constexpr (WORK >= 2) acc += macc[0] + macc[1] + macc[2] + macc[3];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int STAGE, int RING, int AUX, int SEGB, int WAVES, int WORK = 0>
static void run(const char* name, const char* src, uint64_t total_bytes, uint32_t pitch, uint32_t wgs, float* sink) {
    // whole tiles of 64 rows per workgroup
    const uint64_t tile_bytes = 64ull * pitch;
    const uint64_t tiles = total_bytes / tile_bytes;
    const uint64_t tiles_per_wg = tiles / wgs;
    const uint64_t wg_bytes = tiles_per_wg * tile_bytes;
    if (wg_bytes % STAGE) {
        printf("%-44s: skipped (workgroup range not a multiple of the stage)\n", name);
        return;
    }
    auto k = ring_kernel<STAGE, RING, AUX, SEGB, WAVES, WORK>;
    const size_t lds = (size_t)STAGE * RING;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        printf("%-44s: LDS %zu too large\n", name, lds);
        return;
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), lds, 0, src, wg_bytes, pitch, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("%-44s: launch failed\n", name);
    else printf("%-44s wgs %5u : %8.3f ms  %7.0f GB/s\n", name, wgs, best, (double)wg_bytes * wgs / (best * 1e-3) / 1e9);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

__global__ void fill_random(uint32_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        // two bf16 values of magnitude ~1 with random mantissas and signs
        p[i] = ((uint32_t)x & 0x807F807Fu) | 0x3F803F80u;
    }
}

int main(int argc, char** argv) {
    const uint32_t pitch = argc > 1 ? (uint32_t)atoi(argv[1]) : 1536;  // bytes per mirror row (768 bf16)
    const uint64_t rows = 10000000ull / 64 * 64;
    const uint64_t bytes = rows * pitch;
    char* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&sink), 64) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    if (argc > 3) hipMemset(buf, 1, bytes);  // constant data (lower power: clocks differ)
    else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(buf), (size_t)(bytes / 4));
    hipDeviceSynchronize();
    printf("mirror %.2f GB, row pitch %u B\n", bytes / 1e9, pitch);
    if (argc > 2) {  // consumption models on the two layouts, 1024 workgroups
        const uint32_t wgs = 1024;
        run<32768, 4, 2, 512, 4, 0>("SEG  4x32K nt   ring alone", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 2, 512, 4, 1>("SEG  4x32K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 2, 512, 4, 2>("SEG  4x32K nt   + reads + MFMA", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 2, 512, 4, 3>("SEG  4x32K nt   + reads + MFMA + epilogue", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 2, 0, 4, 1>("FULL 6x24K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 2, 0, 4, 2>("FULL 6x24K nt   + reads + MFMA", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 2, 0, 4, 3>("FULL 6x24K nt   + reads + MFMA + epilogue", buf, bytes, pitch, wgs, sink);
        run<16384, 8, 2, 256, 4, 2>("SEG  8x16K nt   + reads + MFMA", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 2, 512, 8, 2>("SEG  4x32K nt 8 waves + reads + MFMA (each wave whole stage)", buf, bytes, pitch, wgs, sink);
        return 0;
    }
    if (pitch == 768 && argc > 2) {  // consumption models on the 8-bit sweep's stages
        for (uint32_t wgs : {256u, 1024u}) {
            run<16384, 8, 2, 256, 4, 0>("SEG  8x16K nt   ring alone", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 1>("SEG  8x16K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 2>("SEG  8x16K nt   + reads + 1 MFMA per read", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 3>("SEG  8x16K nt   + reads + MFMA + epilogue", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 768, 4, 1>("ROWS 3x48K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 768, 4, 2>("ROWS 3x48K nt   + reads + 1 MFMA per read", buf, bytes, pitch, wgs, sink);
        }
        return 0;
    }
    if (pitch == 768) {  // the 8-bit mirror of 768-element rows: 256-B segments (the batched sweep's stages) against whole rows
        for (uint32_t wgs : {256u, 512u, 1024u, 2048u}) {
            run<16384, 8, 2, 256, 4>("SEG  64x256B  ring 8x16K nt (the 8-bit sweep)", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 768, 4>("ROWS 64x768B  ring 3x48K nt, swizzled rows", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 0, 4>("FULL 64 rows  ring 3x48K nt", buf, bytes, pitch, wgs, sink);
            run<24576, 5, 2, 0, 4>("FULL 32 rows  ring 5x24K nt", buf, bytes, pitch, wgs, sink);
            run<24576, 6, 2, 0, 4>("FULL 32 rows  ring 6x24K nt", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 0, 4>("FULL 16K      ring 8x16K nt", buf, bytes, pitch, wgs, sink);
            run<12288, 10, 2, 0, 4>("FULL 16 rows  ring 10x12K nt", buf, bytes, pitch, wgs, sink);
        }
        return 0;
    }
    for (uint32_t wgs : {256u, 512u, 1024u, 2048u}) {
        run<32768, 4, 2, 512, 4>("SEG  64x512B  ring 4x32K nt (round 1)", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 0, 512, 4>("SEG  64x512B  ring 4x32K default", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 2, 0, 4>("FULL 16 rows  ring 6x24K nt", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 0, 0, 4>("FULL 16 rows  ring 6x24K default", buf, bytes, pitch, wgs, sink);
        run<24576, 4, 2, 0, 4>("FULL 16 rows  ring 4x24K nt", buf, bytes, pitch, wgs, sink);
        run<49152, 3, 2, 0, 4>("FULL 32 rows  ring 3x48K nt", buf, bytes, pitch, wgs, sink);
        run<12288, 12, 2, 0, 4>("FULL  8 rows  ring 12x12K nt", buf, bytes, pitch, wgs, sink);
        run<24576, 6, 2, 0, 8>("FULL 16 rows  ring 6x24K nt, 8 waves", buf, bytes, pitch, wgs, sink);
        run<32768, 4, 2, 512, 8>("SEG  64x512B  ring 4x32K nt, 8 waves", buf, bytes, pitch, wgs, sink);
    }
    return 0;
}
