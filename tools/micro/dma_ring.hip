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
    // WORK >= 4: the 8-bit batched sweep's arithmetic — two v_mfma_i32_16x16x64_i8 per fragment read (query planes h and l: 24 stationary
    // fragments of 4 VGPRs each, as at 768 elements), eight int32 accumulators; WORK >= 5: + its tile epilogue every 3 stages (planes combined in
    // float, per-row factors from LDS, tile maximum, key, the four lane groups' meeting, the pending maximum in LDS, a 16-byte store per query
    // every fourth tile); WORK == 6: + the stage's DMA pieces spread over its k-steps instead of one burst behind the barrier.
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    v4i_ ih[4] = {}, il[4] = {};
    v4i_ qh[12], ql[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        qh[i] = (v4i_){(int)(lane * 3 + i), 2 * i + 1, (int)lane, 7 - i};
        ql[i] = (v4i_){(int)(lane + 5 * i), i - 3, 11, (int)(lane ^ i)};
    }
    float* const facs = lds + RING * (STAGE / 4);                                   // [64] per-row factors (WORK >= 5; 4 KiB behind the ring)
    uint32_t* const pend = reinterpret_cast<uint32_t*>(lds + RING * (STAGE / 4) + 64) + (wave * 16u + (lane & 15u)) * 4u;
    if constexpr (WORK >= 5) {
        if (threadIdx.x < 64) facs[threadIdx.x] = 1.0f / (1.0f + threadIdx.x);
    }
    uint32_t wmaxk = 0, tile_no = 0;
    for (uint32_t s = 0; s < n_stage; s++) {
        const uint32_t after = n_stage - 1u - s < (uint32_t)(RING - 2) ? n_stage - 1u - s : (uint32_t)(RING - 2);
        if (after == RING - 2) wait_vm<(RING - 2) * PIECES>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (WORK != 6) {
            if (s + RING - 1 < n_stage) issue(s + RING - 1);
        }
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
            } else if constexpr (WORK >= 4) {
                const uint32_t kc3 = s % 3u;
                const bool live = s + RING - 1 < n_stage;
#pragma unroll
                for (int i = 0; i < NR; i++) {
                    const int ks = i / 4, rb = i % 4;
                    const v4i_ av = __builtin_bit_cast(v4i_, a[i]);
                    // (fragment index by stage of the tile: a select chain keeps the three sets in registers)
                    const v4i_ fh = kc3 == 0 ? qh[ks] : kc3 == 1 ? qh[4 + ks] : qh[8 + ks];
                    const v4i_ fl = kc3 == 0 ? ql[ks] : kc3 == 1 ? ql[4 + ks] : ql[8 + ks];
                    ih[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, fh, ih[rb], 0, 0, 0);
                    il[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, fl, il[rb], 0, 0, 0);
                    if constexpr (WORK == 6) {
                        if (rb == 3 && PIECES >= 4) {  // one piece of the stage ahead per k-step
                            const uint32_t sn = s + RING - 1;
                            if (live) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stage_src(sn) + loff[ks * PIECES / 4]),
                                                                      (__attribute__((address_space(3))) void*)(lds + (sn % RING) * (STAGE / 4) + (wave * PIECES + ks * PIECES / 4) * 256u), 16, 0, AUX);
                        }
                    }
                }
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
        if constexpr (WORK >= 5) {
            if (s % 3 == 2) {  // the tile's epilogue
                const uint32_t g = lane >> 4;
                float m = -1e30f;
#pragma unroll
                for (int rb = 0; rb < 4; rb++) {
                    const v4f nv = *reinterpret_cast<const v4f*>(facs + rb * 16 + g * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float v = ((float)ih[rb][e] + (float)il[rb][e] * 0.00390625f) * nv[e];
                        m = __builtin_fmaxf(m, v);
                    }
                    ih[rb] = (v4i_){0, 0, 0, 0};
                    il[rb] = (v4i_){0, 0, 0, 0};
                }
                m *= 0.37f;
                uint32_t key = __float_as_uint(m);
                key = (key & 0x80000000u) ? ~key : (key | 0x80000000u);
                if constexpr (WORK != 8) {
                    const auto r32 = __builtin_amdgcn_permlane32_swap(key, key, false, false);
                    key = max((uint32_t)r32[0], (uint32_t)r32[1]);
                    const auto r16 = __builtin_amdgcn_permlane16_swap(key, key, false, false);
                    key = max((uint32_t)r16[0], (uint32_t)r16[1]);
                }
                const uint32_t slot = tile_no & 3u;
                if (g == 0 && WORK != 8) pend[slot] = key;
                if (slot == 3u && WORK != 7 && WORK != 8) {
                    if (g == 0) {
                        const u4v v = *reinterpret_cast<const u4v*>(pend);
                        *reinterpret_cast<u4v*>(sink + 4096 + ((size_t)((blockIdx.x * WAVES + wave) * 16u + (lane & 15u)) % 4096u) * 1024u + (tile_no & ~3u) % 1024u) = v;
                    }
                }
                wmaxk = max(wmaxk, key);
                tile_no++;
            }
        }
    }
    if constexpr (WORK >= 4) acc[0] += (float)(ih[0][0] + il[1][1] + ih[2][2] + il[3][3]) + (float)wmaxk;
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
    const size_t lds = (size_t)STAGE * RING + (WORK >= 5 ? 4096 + 256 : 0);
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
    if (hipMalloc(reinterpret_cast<void**>(&buf), bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&sink), (size_t)64 << 20) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    if (argc > 3) hipMemset(buf, 1, bytes);  // constant data (lower power: clocks differ)
    else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(buf), (size_t)(bytes / 4));
    hipDeviceSynchronize();
    printf("mirror %.2f GB, row pitch %u B\n", bytes / 1e9, pitch);
    if (pitch == 768 && argc > 2) {  // consumption models on the 8-bit sweep's stages
        for (uint32_t wgs : {256u, 1024u}) {
            run<16384, 8, 2, 256, 4, 0>("SEG  8x16K nt   ring alone", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 1>("SEG  8x16K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 2>("SEG  8x16K nt   + reads + 1 MFMA per read", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 3>("SEG  8x16K nt   + reads + MFMA + epilogue", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 4>("SEG  8x16K nt   + reads + 2 i8 MFMAs per read", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 5>("SEG  8x16K nt   + 2 i8 MFMAs + the sweep's epilogue", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 6>("SEG  8x16K nt   + ... + DMA pieces spread over k-steps", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 7>("SEG  8x16K nt   + epilogue without the group store", buf, bytes, pitch, wgs, sink);
            run<16384, 8, 2, 256, 4, 8>("SEG  8x16K nt   + epilogue: combine + factors + max only", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 768, 4, 1>("ROWS 3x48K nt   + whole-stage reads", buf, bytes, pitch, wgs, sink);
            run<49152, 3, 2, 768, 4, 2>("ROWS 3x48K nt   + reads + 1 MFMA per read", buf, bytes, pitch, wgs, sink);
        }
        return 0;
    }
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
