// nmn_sortk.hip — the large-k path: full ordering of a shard for k > NMN_MAX_TOP_K.
//
// The reference ranks by a full stable sort of every score and truncates to k
// (vector_engine/src/lib.rs:2026-2034), so any k up to the corpus size is legal; callers do use it
// (post-filter oversampling asks for 3k, lib.rs:3567-3568; "return everything ranked").  The
// candidate pipeline (select -> rescore -> final) holds its working set in one workgroup's LDS and
// stops at k = 4096.  Beyond that the shard takes this path instead:
//
//   exact_scan   reference-order score of every participating row          (nmn_exact.hip)
//   keys         composite u64 = order-preserving score key << 32 | ~row; 0 for masked / padding
//   sort         bitonic sort, descending, of the next power of two >= rows:
//                  tile sort in LDS (4096 keys per workgroup), then per stage the strides >= 4096 as
//                  streaming compare-exchange passes and the strides < 4096 as one LDS merge
//   emit         first k keys -> (row_base + row, score); count = min(k, participating rows)
//
// Composite keys are unique, so "descending by composite" IS (score desc, row asc) — the same order
// the candidate pipeline emits.  Everything is HBM-streaming u64 work; 10M rows = 2^24 keys = 78
// stride passes + 13 LDS passes over 128 MiB.
#include <algorithm>

#include "nmn_internal.h"

namespace nmn {

namespace {

constexpr uint32_t kSortTile = 4096;    // keys per workgroup in the LDS kernels
constexpr uint32_t kSortThreads = 1024;

__global__ __launch_bounds__(256) void largek_keys_kernel(const uint32_t* __restrict__ score_bits, uint64_t n_rows,
                                                          uint64_t n_sort, uint64_t* __restrict__ keys) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sort; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t key = 0;
        if (i < n_rows) {
            const uint32_t sk = bits_to_key(score_bits[i]);
            if (sk != kKeyMasked) key = ((uint64_t)sk << 32) | (uint32_t)~(uint32_t)i;
        }
        keys[i] = key;
    }
}

// descending overall: block of size `stage` containing element i is sorted descending when (i & stage) == 0
__device__ __forceinline__ void cmp_swap(uint64_t& a, uint64_t& b, bool desc) {
    const bool out_of_order = desc ? (a < b) : (a > b);
    if (out_of_order) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// sort every tile of kSortTile keys (all stages 2 .. kSortTile) in LDS
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_sort_kernel(uint64_t* __restrict__ keys) {
    __shared__ uint64_t t[kSortTile];
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kSortThreads) t[i] = keys[base + i];
    __syncthreads();
    for (uint32_t stage = 2; stage <= kSortTile; stage <<= 1) {
        for (uint32_t j = stage >> 1; j > 0; j >>= 1) {
            for (uint32_t p = threadIdx.x; p < kSortTile / 2; p += kSortThreads) {
                const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));  // lower index of pair p at stride j
                const bool desc = (((base + i) & stage) == 0);
                cmp_swap(t[i], t[i + j], desc);
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kSortThreads) keys[base + i] = t[i];
}

// one compare-exchange pass at stride j >= kSortTile of stage `stage`
__global__ __launch_bounds__(256) void bitonic_stride_kernel(uint64_t* __restrict__ keys, uint64_t n_pairs, uint64_t j,
                                                             uint64_t stage) {
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        uint64_t a = keys[i], b = keys[i + j];
        const bool desc = (i & stage) == 0;
        const bool out_of_order = desc ? (a < b) : (a > b);
        if (out_of_order) {
            keys[i] = b;
            keys[i + j] = a;
        }
    }
}

// strides kSortTile/2 .. 1 of stage `stage` (> kSortTile) in LDS
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_merge_kernel(uint64_t* __restrict__ keys, uint64_t stage) {
    __shared__ uint64_t t[kSortTile];
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kSortThreads) t[i] = keys[base + i];
    __syncthreads();
    const bool desc = (base & stage) == 0;  // the whole tile lies in one block of the stage
    for (uint32_t j = kSortTile >> 1; j > 0; j >>= 1) {
        for (uint32_t p = threadIdx.x; p < kSortTile / 2; p += kSortThreads) {
            const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
            cmp_swap(t[i], t[i + j], desc);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kSortThreads) keys[base + i] = t[i];
}

__global__ __launch_bounds__(256) void largek_emit_kernel(const uint64_t* __restrict__ keys, uint64_t n_sort, uint32_t k,
                                                          uint64_t row_base, uint64_t* __restrict__ out_rows,
                                                          float* __restrict__ out_scores, uint32_t* __restrict__ out_count) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = i < n_sort ? keys[i] : 0ull;
        uint64_t row = UINT64_MAX;
        float sc = u2f(0xFF800000u);  // -inf
        if (key != 0) {
            row = row_base + (uint64_t)(uint32_t)~(uint32_t)key;
            sc = key_to_score((uint32_t)(key >> 32));
        }
        out_rows[i] = row;
        out_scores[i] = sc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // participating rows = position of the first zero key (zeros sort last); count = min(k, that)
        uint64_t lo = 0, hi = n_sort;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (keys[mid] != 0) lo = mid + 1;
            else hi = mid;
        }
        *out_count = (uint32_t)(lo < (uint64_t)k ? lo : (uint64_t)k);
    }
}

}  // namespace

uint64_t largek_sort_len(uint64_t n_rows) {
    uint64_t n = kSortTile;
    while (n < n_rows) n <<= 1;
    return n;
}

// the bitonic sort of keys[0 .. n) (n a power of two >= kSortTile), descending
static void sort_keys(uint64_t* keys, uint64_t n, hipStream_t s) {
    const uint32_t tiles = (uint32_t)(n / kSortTile);
    hipLaunchKernelGGL(bitonic_tile_sort_kernel, dim3(tiles), dim3(kSortThreads), 0, s, keys);
    const uint32_t grid_pairs = (uint32_t)std::min<uint64_t>((n / 2 + 255) / 256, 256ull * 32ull);
    for (uint64_t stage = (uint64_t)kSortTile << 1; stage <= n; stage <<= 1) {
        for (uint64_t j = stage >> 1; j >= kSortTile; j >>= 1)
            hipLaunchKernelGGL(bitonic_stride_kernel, dim3(grid_pairs), dim3(256), 0, s, keys, n / 2, j, stage);
        hipLaunchKernelGGL(bitonic_tile_merge_kernel, dim3(tiles), dim3(kSortThreads), 0, s, keys, stage);
    }
}

// sel_hist / sel_sync (the scratch of fallback_select_kernel, both or neither): when k is a small part of the shard the k
// best composites are first SELECTED — the device-wide radix select of the exact fallback finds the k-th largest composite
// and appends everything >= it, in any order, to keys[0 .. k) — and only those are sorted: 10M rows, k = 10 000 sorts 2^14
// keys instead of 2^24 (91 passes over 128 MiB).  The selected set is exactly the first k of the full order (composites are
// unique), so the output is the same bit for bit.
hipError_t launch_largek(const uint32_t* score_bits, uint64_t n_rows, uint64_t* keys, uint32_t k, uint64_t row_base,
                         uint64_t* out_rows, float* out_scores, uint32_t* out_count, hipStream_t s, uint32_t* sel_hist,
                         unsigned long long* sel_sync) {
    uint64_t n = largek_sort_len(n_rows);
    uint64_t n_sel = kSortTile;
    while (n_sel < (uint64_t)k) n_sel <<= 1;
    bool selected = false;
    if (sel_hist && sel_sync && n_rows >= (1u << 18) && n_sel * 4 <= n) {
        hipError_t e = hipMemsetAsync(keys, 0, n_sel * sizeof(uint64_t), s);  // unused slots sort last (and mark the count)
        if (e != hipSuccess) return e;
        FallbackParams fb{};
        fb.all = 1;
        fb.scores = score_bits;  // plain row order: score_at(row, 0, 1) == row
        fb.nql = 1;
        fb.nq = 1;
        fb.k = k;
        fb.n_rows = n_rows;
        fb.ghist = sel_hist;
        fb.list = reinterpret_cast<unsigned long long*>(keys);
        fb.list_cap = (uint32_t)n_sel;
        fb.list_count = out_count;  // (overwritten by the emit kernel with the same value)
        fb.sync = sel_sync;
        // (a cooperative launch: the grid is resident together or the launch is refused — then the full sort answers)
        e = launch_fallback_select(fb, s);
        if (e == hipSuccess) {
            n = n_sel;
            selected = true;
        } else {
            (void)hipGetLastError();
        }
    }
    if (!selected) {
        const uint32_t grid_stream = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 32ull);
        hipLaunchKernelGGL(largek_keys_kernel, dim3(grid_stream), dim3(256), 0, s, score_bits, n_rows, n, keys);
    }
    sort_keys(keys, n, s);
    const uint32_t grid_emit = (uint32_t)std::min<uint64_t>(((uint64_t)k + 255) / 256, 4096ull);
    hipLaunchKernelGGL(largek_emit_kernel, dim3(grid_emit), dim3(256), 0, s, keys, n, k, row_base, out_rows, out_scores,
                       out_count);
    return hipGetLastError();
}

}  // namespace nmn
