// nmn_select.hip — top-k selection around the scan: threshold pick + candidate collection, the
// final (exact score, row) sort, the exact-fallback selection and the multi-shard merge.
//
// Replaces `results.sort_by(score desc); results.truncate(k)` over ALL N results
// (vector_engine/src/lib.rs:2027-2034, 2093-2100) and `ResultMerger::merge_top_k`
// (query_router/src/distributed.rs:413-433).  Order everywhere: score descending, ties by ascending
// row id; NaN scores last.
#include "nmn_internal.h"

namespace nmn {

constexpr int kSelThreads = 1024;
constexpr int kBins = 2048;

struct PickResult {
    uint32_t bin;    // bin holding the kk-th largest key
    uint32_t above;  // keys in bins strictly above it
};

// hist[0..nbins) filled; find the bin containing the kk-th largest key (1 <= kk <= total).
// Executed by wave 0; result broadcast through `out` (LDS).
__device__ void pick_bin(const uint32_t* hist, int nbins, uint32_t kk, PickResult* out) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int per = nbins / 64;
        uint32_t s = 0;
        for (int b = 0; b < per; b++) s += hist[lane * per + b];
        // inclusive suffix sum over lanes: S_i = sum_{j>=i} s_j
        uint32_t S = s;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_down(S, off);
            if (lane + off < 64) S += t;
        }
        const unsigned long long m = __ballot(S >= kk);
        const int star = 63 - __builtin_clzll(m);  // m != 0 because S_0 = total >= kk
        if (lane == star) {
            uint32_t above = S - s;
            int b = per - 1;
            for (; b > 0; b--) {
                const uint32_t h = hist[lane * per + b];
                if (above + h >= kk) break;
                above += h;
            }
            out->bin = (uint32_t)(lane * per + b);
            out->above = above;
        }
    }
}

// wave-aggregated append: returns the slot of this lane's element (or UINT32_MAX if !pred)
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, leader);
    const uint32_t ofs = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    return pred ? base + ofs : 0xFFFFFFFFu;
}

// ---- main-path selection: one workgroup per query ---------------------------------------------
// keys = tile maxima (use_tiles) or every row's score key.  Two 11-bit radix passes give T, the
// lower edge of the 2^10-ulp bin holding the k-th largest key: at least k keys are >= T, hence at
// least k rows score >= score(T).  Every row with approx >= score(T) - margin is a candidate; the
// exact top-k is among them.
__global__ void __launch_bounds__(kSelThreads) select_kernel(SelectParams p) {
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_total, s_count;
    const uint32_t q = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t* scores = p.scores + (uint64_t)q * p.score_stride;
    const uint32_t* tmax = p.tmax + (uint64_t)q * p.n_tiles;
    const uint64_t n_pad = (uint64_t)p.n_tiles * kTileRows;
    const uint64_t n_keys = p.use_tiles ? p.n_tiles : n_pad;
    auto getkey = [&](uint64_t i) -> uint32_t { return p.use_tiles ? tmax[i] : bits_to_key(scores[i]); };

    for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
    if (tid == 0) { s_total = 0; s_count = 0; }
    __syncthreads();
    // pass 1: bits 31..21
    uint32_t local_valid = 0;
    for (uint64_t i = tid; i < n_keys; i += kSelThreads) {
        const uint32_t key = getkey(i);
        if (key == kKeyMasked) continue;
        local_valid++;
        atomicAdd(&hist[key >> 21], 1u);
    }
    atomicAdd(&s_total, local_valid);
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) {
        if (tid == 0) {
            p.qstate[q].cand_count = 0;
            p.qstate[q].n_valid = 0;
        }
        return;
    }
    uint32_t Tc = kKeyNaN;  // collect everything that takes part
    if (total > p.k) {
        pick_bin(hist, kBins, p.k, &pick);
        __syncthreads();
        const uint32_t b1 = pick.bin, above1 = pick.above;
        __syncthreads();
        for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
        __syncthreads();
        // pass 2: bits 20..10 inside bin b1
        for (uint64_t i = tid; i < n_keys; i += kSelThreads) {
            const uint32_t key = getkey(i);
            if (key != kKeyMasked && (key >> 21) == b1) atomicAdd(&hist[(key >> 10) & 2047u], 1u);
        }
        __syncthreads();
        pick_bin(hist, kBins, p.k - above1, &pick);
        __syncthreads();
        const uint32_t T = (b1 << 21) | (pick.bin << 10);
        if (T > kKeyNegInf) {
            const float tau = key_to_score(T);
            const QInfo qi = p.qinfo[q];
            const float thr = tau - qi.margin_abs - fabsf(tau) * qi.margin_rel;
            if (thr == thr) {
                Tc = score_to_key(thr);
                if (Tc > T) Tc = T;
                if (Tc < kKeyNaN) Tc = kKeyNaN;
            }
        }
    }
    // pass 3: collect rows with key >= Tc
    uint32_t* out = p.cand_rows + (size_t)q * p.cand_cap;
    if (p.use_tiles) {
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        const uint32_t nw = kSelThreads / 64;
        for (uint64_t tb = (uint64_t)wave * 64u; tb < p.n_tiles; tb += (uint64_t)nw * 64u) {
            const uint64_t t = tb + lane;
            const uint32_t tk = t < p.n_tiles ? tmax[t] : 0u;
            unsigned long long m = __ballot(tk != kKeyMasked && tk >= Tc);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint64_t row = (tb + (uint64_t)b) * kTileRows + lane;
                const uint32_t key = bits_to_key(scores[row]);
                const bool pred = key != kKeyMasked && key >= Tc;
                const uint32_t pos = wave_append(pred, &s_count);
                if (pred && pos < p.cand_cap) out[pos] = (uint32_t)row;
            }
        }
    } else {
        const uint64_t n_round = (n_pad + kSelThreads - 1) / kSelThreads * kSelThreads;
        for (uint64_t i = tid; i < n_round; i += kSelThreads) {
            const uint32_t key = i < n_pad ? bits_to_key(scores[i]) : kKeyMasked;
            const bool pred = key != kKeyMasked && key >= Tc;
            const uint32_t pos = wave_append(pred, &s_count);
            if (pred && pos < p.cand_cap) out[pos] = (uint32_t)i;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t c = s_count;
        QState st;
        st.n_valid = total;
        st.thr_key = Tc;
        st.overflow = c > p.cand_cap ? 1u : 0u;
        st.cand_count = c > p.cand_cap ? 0u : c;
        p.qstate[q] = st;
    }
}

hipError_t launch_select(const SelectParams& p, hipStream_t s) {
    hipLaunchKernelGGL(select_kernel, dim3(p.nq), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

// ---- exact-fallback selection -----------------------------------------------------------------
// scores[] now holds EXACT scores (exact_scan_kernel).  Three radix passes find the exact key of the
// k-th best row; rows strictly above it are taken in any order, rows tied with it in ascending row
// order until k rows are chosen — exactly the (score desc, row asc) prefix, whatever the number of
// ties.  One workgroup per flagged query (slow by design: this path only runs when more than
// cand_cap rows sit within the rounding margin of the k-th score, e.g. masses of duplicates).
__global__ void __launch_bounds__(kSelThreads) exact_select_kernel(ExactSelectParams p) {
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_total, s_count, s_wsum[kSelThreads / 64], s_run;
    const uint32_t q = blockIdx.x;
    if (p.qstate[q].overflow == 0) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t* scores = p.scores + (uint64_t)q * p.score_stride;
    const uint64_t n_pad = (p.n_rows + 63) & ~63ull;

    for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
    if (tid == 0) { s_total = 0; s_count = 0; s_run = 0; }
    __syncthreads();
    uint32_t local_valid = 0;
    for (uint64_t i = tid; i < n_pad; i += kSelThreads) {
        const uint32_t key = bits_to_key(scores[i]);
        if (key == kKeyMasked) continue;
        local_valid++;
        atomicAdd(&hist[key >> 21], 1u);
    }
    atomicAdd(&s_total, local_valid);
    __syncthreads();
    const uint32_t total = s_total;
    const uint32_t kk = min(p.k, total);
    uint32_t* out = p.cand_rows + (size_t)q * p.cand_cap;
    if (kk == 0) {
        if (tid == 0) p.qstate[q].cand_count = 0;
        return;
    }
    pick_bin(hist, kBins, kk, &pick);
    __syncthreads();
    const uint32_t b1 = pick.bin, above1 = pick.above;
    __syncthreads();
    for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
    __syncthreads();
    for (uint64_t i = tid; i < n_pad; i += kSelThreads) {
        const uint32_t key = bits_to_key(scores[i]);
        if (key != kKeyMasked && (key >> 21) == b1) atomicAdd(&hist[(key >> 10) & 2047u], 1u);
    }
    __syncthreads();
    pick_bin(hist, kBins, kk - above1, &pick);
    __syncthreads();
    const uint32_t b2 = pick.bin, above2 = above1 + pick.above;
    __syncthreads();
    for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
    __syncthreads();
    const uint32_t hi = (b1 << 11) | b2;
    for (uint64_t i = tid; i < n_pad; i += kSelThreads) {
        const uint32_t key = bits_to_key(scores[i]);
        if (key != kKeyMasked && (key >> 10) == hi) atomicAdd(&hist[key & 1023u], 1u);
    }
    __syncthreads();
    pick_bin(hist, 1024, kk - above2, &pick);
    __syncthreads();
    const uint32_t Tk = (hi << 10) | pick.bin;
    const uint32_t above = above2 + pick.above;  // rows strictly better than the k-th
    const uint32_t need = kk - above;            // tied rows to take, lowest row ids first
    // (a) strictly better rows, any order
    const uint64_t n_round = (n_pad + kSelThreads - 1) / kSelThreads * kSelThreads;
    for (uint64_t i = tid; i < n_round; i += kSelThreads) {
        const uint32_t key = i < n_pad ? bits_to_key(scores[i]) : kKeyMasked;
        const bool pred = key != kKeyMasked && key > Tk;
        const uint32_t pos = wave_append(pred, &s_count);
        if (pred && pos < p.cand_cap) out[pos] = (uint32_t)i;
    }
    __syncthreads();
    // (b) ties in ascending row order
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    for (uint64_t base = 0; base < n_round; base += kSelThreads) {
        const uint64_t i = base + tid;
        const uint32_t key = i < n_pad ? bits_to_key(scores[i]) : kKeyMasked;
        const bool tie = key == Tk;
        const unsigned long long m = __ballot(tie);
        if (lane == 0) s_wsum[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = s_run;
        for (uint32_t w = 0; w < wave; w++) before += s_wsum[w];
        const uint32_t idx = before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (tie && idx < need) out[above + idx] = (uint32_t)i;
        __syncthreads();
        if (tid == 0) {
            uint32_t t = s_run;
            for (uint32_t w = 0; w < kSelThreads / 64; w++) t += s_wsum[w];
            s_run = t;
        }
        __syncthreads();
        if (s_run >= need) break;
    }
    if (tid == 0) p.qstate[q].cand_count = kk;
}

hipError_t launch_exact_select(const ExactSelectParams& p, hipStream_t s) {
    hipLaunchKernelGGL(exact_select_kernel, dim3(p.nq), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

// ---- final sort: candidates by (exact score desc, row asc) -> top-k ---------------------------
__global__ void __launch_bounds__(kSelThreads) final_kernel(FinalParams p) {
    __shared__ unsigned long long list[NMN_MAX_TOP_K];
    const uint32_t q = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = min(p.qstate[q].cand_count, min(p.cand_cap, (uint32_t)NMN_MAX_TOP_K));
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (uint32_t i = tid; i < np2; i += kSelThreads) {
        unsigned long long v = 0ull;
        if (i < n) {
            const uint32_t row = p.cand_rows[(size_t)q * p.cand_cap + i];
            const uint32_t key = score_to_key(p.cand_scores[(size_t)q * p.cand_cap + i]);
            v = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - row);
        }
        list[i] = v;
    }
    __syncthreads();
    // bitonic sort, descending
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (np2 >> 1); t += kSelThreads) {
                const uint32_t lo = ((t / stride) * stride * 2u) + (t % stride);
                const uint32_t hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = list[lo], b = list[hi];
                if ((a < b) == desc) {
                    list[lo] = b;
                    list[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t cnt = min(n, p.k);
    for (uint32_t i = tid; i < p.k; i += kSelThreads) {
        uint64_t row = UINT64_MAX;
        float sc = u2f(0xFF800000u);  // -inf
        if (i < cnt) {
            const unsigned long long v = list[i];
            row = p.row_base + (uint64_t)(0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
            sc = key_to_score((uint32_t)(v >> 32));
        }
        p.out_rows[(size_t)q * p.k + i] = row;
        p.out_scores[(size_t)q * p.k + i] = sc;
    }
    if (tid == 0) p.out_counts[q] = cnt;
}

hipError_t launch_final(const FinalParams& p, hipStream_t s) {
    hipLaunchKernelGGL(final_kernel, dim3(p.nq), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

// ---- shard merge (merge_top_k, distributed.rs:413-433) ----------------------------------------
// Inputs [list][query][k] are each already in final order; rows are unique across shards, so
// (key desc, row asc) is a strict total order and an element's output slot is simply the number of
// elements of all lists that precede it (binary search per list).
__device__ __forceinline__ bool hit_before(uint32_t ka, uint64_t ra, uint32_t kb, uint64_t rb) {
    return ka > kb || (ka == kb && ra < rb);
}

__global__ void __launch_bounds__(256) merge_kernel(const uint64_t* __restrict__ rows,
                                                    const float* __restrict__ scores,
                                                    const uint32_t* __restrict__ counts, uint32_t n_lists,
                                                    uint32_t nq, uint32_t k, uint64_t* __restrict__ out_rows,
                                                    float* __restrict__ out_scores,
                                                    uint32_t* __restrict__ out_counts) {
    const uint32_t q = blockIdx.x;
    uint32_t total = 0;
    for (uint32_t l = 0; l < n_lists; l++) total += min(counts[(size_t)l * nq + q], k);
    const uint32_t cnt = min(total, k);
    for (uint32_t e = threadIdx.x; e < n_lists * k; e += blockDim.x) {
        const uint32_t l = e / k, i = e - l * k;
        const uint32_t cl = min(counts[(size_t)l * nq + q], k);
        if (i >= cl) continue;
        const size_t base = ((size_t)l * nq + q) * k;
        const float sc = scores[base + i];
        const uint32_t key = score_to_key(sc);
        const uint64_t row = rows[base + i];
        uint32_t rank = i;
        for (uint32_t m = 0; m < n_lists; m++) {
            if (m == l) continue;
            const uint32_t cm = min(counts[(size_t)m * nq + q], k);
            const size_t bm = ((size_t)m * nq + q) * k;
            uint32_t lo = 0, hi = cm;  // first index in list m that does NOT precede e
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (hit_before(score_to_key(scores[bm + mid]), rows[bm + mid], key, row)) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_rows[(size_t)q * k + rank] = row;
            out_scores[(size_t)q * k + rank] = sc;
        }
    }
    for (uint32_t i = cnt + threadIdx.x; i < k; i += blockDim.x) {
        out_rows[(size_t)q * k + i] = UINT64_MAX;
        out_scores[(size_t)q * k + i] = u2f(0xFF800000u);
    }
    if (threadIdx.x == 0) out_counts[q] = cnt;
}

hipError_t launch_merge(const uint64_t* rows, const float* scores, const uint32_t* counts, uint32_t n_lists,
                        uint32_t nq, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                        hipStream_t s) {
    hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(256), 0, s, rows, scores, counts, n_lists, nq, k, out_rows,
                       out_scores, out_counts);
    return hipGetLastError();
}

// ---- count of rows above / equal to a score (certificate) -------------------------------------
__global__ void __launch_bounds__(256) count_cmp_kernel(const uint32_t* __restrict__ scores, uint64_t n_rows,
                                                        uint32_t ref_key, unsigned long long* __restrict__ out2) {
    unsigned long long gt = 0, eq = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t key = bits_to_key(scores[i]);
        if (key == kKeyMasked) continue;
        gt += key > ref_key;
        eq += key == ref_key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        gt += __shfl_down(gt, off);
        eq += __shfl_down(eq, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (gt) atomicAdd(&out2[0], gt);
        if (eq) atomicAdd(&out2[1], eq);
    }
}

hipError_t launch_count_cmp(const uint32_t* scores, uint64_t n_rows, float score, unsigned long long* out2,
                            hipStream_t s) {
    if (n_rows == 0) return hipSuccess;
    uint64_t blocks = (n_rows + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(count_cmp_kernel, dim3((unsigned)blocks), dim3(256), 0, s, scores, n_rows,
                       score_to_key(score), out2);
    return hipGetLastError();
}

}  // namespace nmn
