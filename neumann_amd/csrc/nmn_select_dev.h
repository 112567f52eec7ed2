// nmn_select_dev.h — device-side selection helpers shared by nmn_select.hip (the pipeline's select / final kernels) and
// nmn_exact.hip (the single-launch search of small shards).  No relocatable device code in this build: every translation
// unit gets its own copy (static).
#pragma once
#include "nmn_internal.h"

namespace nmn {

constexpr int kSelThreads = 1024;
constexpr int kBins = 2048;

struct PickResult {
    uint32_t bin;    // bin holding the kk-th largest key
    uint32_t above;  // keys in bins strictly above it
};

// hist[0..nbins) filled (nbins = 1024 or 2048); find the bin containing the kk-th largest key
// (1 <= kk <= total).  Block-wide: every thread owns nbins/1024 adjacent bins, suffix sums by wave
// shuffles + one LDS hop across the 16 waves; exactly one thread sees the crossing and publishes it.
// Callers __syncthreads() before reading *out.
template <int NT = kSelThreads>
static __device__ void pick_bin(const uint32_t* hist, int nbins, uint32_t kk, PickResult* out) {
    __shared__ uint32_t wtot[NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr int kMaxPer = kBins / NT;  // adjacent bins per thread at 2048 bins (1024 threads: 2, 512: 4)
    const int per = nbins / NT;          // 1 .. kMaxPer
    uint32_t h[kMaxPer];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
        h[j] = j < per ? hist[tid * per + j] : 0u;
        s += h[j];
    }
    uint32_t S = s;  // inclusive suffix sum inside the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_down(S, off);
        if (lane + off < 64) S += t;
    }
    if (lane == 0) wtot[wave] = S;
    __syncthreads();
    uint32_t above_waves = 0;
    for (uint32_t w = wave + 1; w < NT / 64; w++) above_waves += wtot[w];
    const uint32_t Sfx = S + above_waves;  // keys in bins >= first bin of this thread
    uint32_t above = Sfx - s;              // keys in bins above this thread's bins
    if (Sfx >= kk && above < kk) {         // the crossing is in one of this thread's bins: the highest j with above(j) + h[j] >= kk
#pragma unroll
        for (int j = kMaxPer - 1; j >= 0; j--) {
            if (j < per) {
                if (above < kk && above + h[j] >= kk) {
                    out->bin = tid * per + j;
                    out->above = above;
                }
                above += h[j];
            }
        }
    }
    __syncthreads();
}

// wave-aggregated append: returns the slot of this lane's element (or UINT32_MAX if !pred).
// Must be called from wave-uniform control flow.
static __device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
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

// One histogram count per lane with `in`, by a whole wave.  The selections that histogram composites serve degenerate inputs —
// at worst every row holds the same score, and then every lane of every wave adds to the SAME bin: 64 serialized LDS atomics
// per wave and element (65 us per digit at 10M rows on 64 workgroups).  So the lanes that share the first active lane's bin
// are counted with one ballot and added once; the others (none in the degenerate case, most on ordinary data) add for
// themselves.  Callable from divergent code: only active lanes take part.
static __device__ __forceinline__ void hist_add_wave(uint32_t* hist, bool in, uint32_t bin) {
    const unsigned long long act = __ballot(in);
    if (act == 0ull) return;
    const int lead = __builtin_ctzll(act);
    const uint32_t b0 = (uint32_t)__shfl((int)bin, lead);
    const unsigned long long same = __ballot(in && bin == b0);
    if ((int)(threadIdx.x & 63u) == lead) atomicAdd(&hist[b0], (uint32_t)__builtin_popcountll(same));
    else if (in && bin != b0) atomicAdd(&hist[bin], 1u);
}

// ---- selection with few barriers (round 5) ------------------------------------------------------------------------------------
// select_kernel is a chain of short workgroup-wide phases, and on a 16-wave workgroup every barrier-separated phase costs ~0.4 us
// whatever it does (profiles/r05h_select_phases.txt: three radix picks over <= 4096 LDS-resident keys took 4 us EACH, ten barriers a
// piece).  These helpers cut the barriers: the histogram is PADDED (one word per 32 bins) so that a lane can read 32 adjacent
// bins without bank conflicts, and then every WAVE finds the crossing bin for itself from the finished histogram — no partial
// sums through LDS, no publication of the result, no barrier.
constexpr uint32_t kHistPad = kBins + kBins / 32;  // words of a padded histogram
static __device__ __forceinline__ uint32_t hpad(uint32_t bin) { return bin + (bin >> 5); }

// hist (padded, kBins bins, complete: the caller has synchronized) -> the bin holding the kk-th largest key and the number of
// keys in the bins above it (1 <= kk <= total), computed redundantly by every wave; all 64 lanes must call.
static __device__ __forceinline__ PickResult wave_pick(const uint32_t* hist, uint32_t kk) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t h[32];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        h[j] = hist[33u * lane + (uint32_t)j];  // bins 32 * lane + j
        s += h[j];
    }
    uint32_t S = s;  // inclusive suffix sum over the lanes (higher lane = higher bins)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_down((int)S, off);
        if (lane + (uint32_t)off < 64u) S += t;
    }
    const uint32_t above = S - s;
    const bool cross = S >= kk && above < kk;
    uint32_t bin = 0, ab = 0, a = above;
    bool found = false;
#pragma unroll
    for (int j = 31; j >= 0; j--) {
        const bool here = !found && a + h[j] >= kk;
        if (here) {
            bin = 32u * lane + (uint32_t)j;
            ab = a;
        }
        found = found || here;
        a += h[j];
    }
    const unsigned long long m = __ballot(cross);
    const int src = m ? __builtin_ctzll(m) : 0;
    PickResult r;
    r.bin = (uint32_t)__shfl((int)bin, src);
    r.above = (uint32_t)__shfl((int)ab, src);
    return r;
}

// cnt items per lane -> the slot of this lane's first item: ONE LDS atomic per wave (a scan of the lanes' counts places them),
// where wave_append pays one per item.  All 64 lanes must call (wave-uniform control flow).
static __device__ __forceinline__ uint32_t wave_append_cnt(uint32_t cnt, uint32_t* counter) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = cnt;
#pragma unroll
    for (uint32_t dd = 1; dd < 64; dd <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, (int)dd);
        if (lane >= dd) incl += t;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (total == 0) return 0;  // (wave-uniform)
    uint32_t base = 0;
    if (lane == 63u) base = atomicAdd(counter, total);
    base = (uint32_t)__shfl((int)base, 63);
    return base + incl - cnt;
}

// Two 11-bit radix digits over n gathered keys (key_at(e) == 0: does not take part): returns T, the
// lower edge of the 2^10-ulp bin that holds the kk-th largest key.  Requires kk <= #valid keys.
// Loads are issued V at a time per thread so a single workgroup still keeps ~8K loads in flight.
template <int NT = kSelThreads, typename KeyAt>
static __device__ uint32_t radix2(KeyAt key_at, uint32_t n, uint32_t kk, uint32_t* hist, PickResult* pick) {
    const uint32_t tid = threadIdx.x;
    constexpr int V = 8;
    uint32_t b1 = 0, above1 = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int b = tid; b < kBins; b += NT) hist[b] = 0;
        __syncthreads();
        for (uint32_t e0 = tid; e0 < n; e0 += NT * V) {
            uint32_t kv[V];
#pragma unroll
            for (int u = 0; u < V; u++) {
                const uint32_t e = e0 + (uint32_t)u * NT;
                kv[u] = e < n ? key_at(e) : kKeyMasked;
            }
#pragma unroll
            for (int u = 0; u < V; u++) {
                const uint32_t key = kv[u];
                const bool in = key != kKeyMasked && (pass == 0 || (key >> 21) == b1);
                hist_add_wave(hist, in, pass == 0 ? key >> 21 : (key >> 10) & 2047u);  // (keys of one query crowd into few bins)
            }
        }
        __syncthreads();
        pick_bin<NT>(hist, kBins, pass == 0 ? kk : kk - above1, pick);
        if (pass == 0) {
            b1 = pick->bin;
            above1 = pick->above;
        }
        __syncthreads();
    }
    return (b1 << 21) | (pick->bin << 10);
}

// `walk(f)`: calls f(row, key) for every element, the same number of times on every lane (key == kKeyMasked: skip).
template <class Walk>
static __device__ uint32_t exact_select_walk(Walk&& walk, uint32_t k, unsigned long long* list, uint32_t* hist, PickResult* pick,
                                      uint32_t* s_misc /* >= 2 words */) {
    const uint32_t tid = threadIdx.x;
    auto comp = [](uint64_t i, uint32_t key) -> unsigned long long {
        return key == kKeyMasked ? 0ull : (((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
    };
    // digit layout over the 64-bit composite, most significant first
    const int shifts[6] = {53, 42, 32, 21, 10, 0};
    const int widths[6] = {11, 11, 10, 11, 11, 10};
    if (tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
    unsigned long long prefix = 0ull;  // digits fixed so far (high bits)
    uint32_t need = 0;                 // rank of the wanted composite among those matching the prefix
    uint32_t kk = 0;
    for (int d = 0; d < 6; d++) {
        const int nb = 1 << widths[d];
        for (int b = tid; b < kBins; b += kSelThreads) hist[b] = 0;
        __syncthreads();
        const int hi_shift = shifts[d] + widths[d];
        uint32_t loc = 0;
        walk([&](uint64_t i, uint32_t key) {
            const unsigned long long c = comp(i, key);
            if (c != 0ull) loc++;
            const bool in = c != 0ull && !(hi_shift < 64 && (c >> hi_shift) != (prefix >> hi_shift));
            hist_add_wave(hist, in, (uint32_t)(c >> shifts[d]) & (uint32_t)(nb - 1));
        });
        if (d == 0) {  // the first walk also counts the participating rows
            atomicAdd(&s_misc[0], loc);
            __syncthreads();
            kk = min(k, s_misc[0]);
            if (kk == 0) return 0;
            need = kk;
        }
        __syncthreads();
        pick_bin(hist, nb, need, pick);
        __syncthreads();
        prefix |= (unsigned long long)pick->bin << shifts[d];
        need -= pick->above;
        // score key fixed and every row holding it is wanted: the row digits cannot change the cut
        const bool done = d == 2 && hist[pick->bin] == need;
        __syncthreads();
        if (done) break;
    }
    // prefix is now the kk-th largest composite (or the score key of the cut with zero row digits, which
    // admits the same set); collect everything >= it (exactly kk entries)
    walk([&](uint64_t i, uint32_t key) {
        const unsigned long long c = comp(i, key);
        const bool pred = c != 0ull && c >= prefix;
        const uint32_t pos = wave_append(pred, &s_misc[1]);
        if (pred && pos < NMN_MAX_TOP_K) list[pos] = c;
    });
    __syncthreads();
    return min(s_misc[1], (uint32_t)NMN_MAX_TOP_K);
}


// ---- wave-level sorting (no barriers): one composite per lane ---------------------------------------------------------------
static __device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
// the wave's 64 values in descending order (lane i = rank i): bitonic network over lane exchanges
static __device__ __forceinline__ unsigned long long wave_sort_desc(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (uint32_t size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, (int)stride);
            const bool keep_max = ((lane & stride) == 0) == ((lane & size) == 0);
            v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}
// a bitonic sequence across the lanes -> descending order
static __device__ __forceinline__ unsigned long long wave_merge_desc(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (uint32_t stride = 32; stride > 0; stride >>= 1) {
        const unsigned long long o = shfl_xor_u64(v, (int)stride);
        v = ((lane & stride) == 0) ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}
// The first min(n_live, k) composites (score key << 32 | ~row) of list[0 .. n) in descending order, as (global row, score) pairs
// padded to k with (UINT64_MAX, -inf).  `n` slots, of which `n_live` are entries (the others 0); entries are distinct (distinct
// rows).  `list` must hold ceil(n / 64) * 64 entries and is overwritten.
//
// Runs and ranks instead of a workgroup-wide bitonic sort: every wave sorts runs of 64 in registers (lane exchanges, no barrier),
// ONE barrier, then an entry's output slot is simply the number of entries before it — its index in its own run plus, by
// binary search, the entries of every other run that are greater (the merge_kernel's argument, within one list); a search
// stops as soon as the rank reaches k.  ~600 candidates (k = 100 under the 8-bit margin): final_kernel 13.4 -> 8.2 us
// (profiles/r04y_*); the bitonic network is 55 barrier steps for 1024 slots.  Lists of up to one entry per thread only; longer
// ones keep the network (-DNMN_SORT_BITONIC: always the network, the A/B).
// Descending bitonic sort of list[0 .. np2) (np2 a power of two, unused slots 0) by the whole workgroup, then the first
// min(n, k) composites (score key << 32 | ~row) as (global row, score) pairs padded to k with (UINT64_MAX, -inf).
// `n` slots are sorted, of which `n_live` are entries (the rest 0: they sort last).
static __device__ void sort_and_emit_bitonic(unsigned long long* list, uint32_t n, uint32_t n_live, uint32_t k, uint64_t row_base,
                                     uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (uint32_t i = n + tid; i < np2; i += nthr) list[i] = 0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (np2 >> 1); t += nthr) {
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
    const uint32_t cnt = min(n_live, k);
    for (uint32_t i = tid; i < k; i += nthr) {
        uint64_t row = UINT64_MAX;
        float sc = u2f(0xFF800000u);  // -inf
        if (i < cnt) {
            const unsigned long long v = list[i];
            row = row_base + (uint64_t)(0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
            sc = key_to_score((uint32_t)(v >> 32));
        }
        out_rows[i] = row;
        out_scores[i] = sc;
    }
    if (tid == 0) *out_count = cnt;
}

static __device__ void sort_and_emit(unsigned long long* list, uint32_t n, uint32_t n_live, uint32_t k, uint64_t row_base,
                                     uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
#ifdef NMN_SORT_BITONIC
    return sort_and_emit_bitonic(list, n, n_live, k, row_base, out_rows, out_scores, out_count);
#endif
    // (more than one entry per thread: a wave runs a full rank search in every pass in which ANY of its lanes holds a top entry —
    //  4096 near-equal candidates of a clustered IVF list took 100 us this way against the network's 36: the network keeps those)
    if (n > max(blockDim.x, 1024u)) return sort_and_emit_bitonic(list, n, n_live, k, row_base, out_rows, out_scores, out_count);  // (a 512-thread workgroup ranks two entries per thread)
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63u, nw = nthr >> 6;
    const uint32_t nruns = (n + 63u) >> 6;
    for (uint32_t r = wave; r < nruns; r += nw) {  // (wave-uniform)
        const uint32_t i = r * 64u + lane;
        const unsigned long long v = i < n ? list[i] : 0ull;
        list[i] = wave_sort_desc(v, lane);
    }
    __syncthreads();
    const uint32_t cnt = min(n_live, k);
    for (uint32_t i = cnt + tid; i < k; i += nthr) {
        out_rows[i] = UINT64_MAX;
        out_scores[i] = u2f(0xFF800000u);  // -inf
    }
    for (uint32_t i = tid; i < nruns * 64u; i += nthr) {
        const unsigned long long v = list[i];
        if (v == 0ull) continue;
        const uint32_t own = i >> 6;
        uint32_t rank = i & 63u;
        // entries of the other runs that come before v: the greater ones — and, in EARLIER runs, equal ones too.  Composites are
        // distinct by construction (distinct rows), but should a list ever carry a row twice the ranks are still a permutation
        // (ties ordered by list position) and every slot below cnt is written.  Two loops, one comparison each: a
        // per-step choice between > and >= cost final_kernel 8.7 -> 12 us at ~600 candidates (profiles/r05j_*).
        for (uint32_t r = 0; r < own && rank < k; r++) {
            const unsigned long long* run = list + r * 64u;  // descending; zeros (never greater) at its end
            uint32_t lo = 0, hi = 64;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (run[mid] >= v) lo = mid + 1u;
                else hi = mid;
            }
            rank += lo;
        }
        for (uint32_t r = own + 1u; r < nruns && rank < k; r++) {
            const unsigned long long* run = list + r * 64u;
            uint32_t lo = 0, hi = 64;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (run[mid] > v) lo = mid + 1u;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            out_rows[rank] = row_base + (uint64_t)(0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
            out_scores[rank] = key_to_score((uint32_t)(v >> 32));
        }
    }
    if (tid == 0) *out_count = cnt;
}

// The 64 largest of list[0 .. n64 * 64) (n64 <= waves of the workgroup; empty slots 0), descending, in WAVE 0 (lane i = rank
// i); every thread of the workgroup calls.  Each wave sorts its 64 in registers, then a tree of merges that keep the top 64
// of two sorted runs (max of A[i] and B[63 - i] is a bitonic sequence of exactly those): one LDS exchange and two barriers
// per level instead of the (log n)^2 / 2 barrier steps of a workgroup-wide bitonic sort.  `list` is overwritten.
static __device__ unsigned long long wg_top64(unsigned long long* list, uint32_t n64) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    unsigned long long v = wave < n64 ? list[tid] : 0ull;
    if (wave < n64) v = wave_sort_desc(v, lane);
    for (uint32_t s = 1; s < n64; s <<= 1) {
        __syncthreads();  // every wave holds its run in registers: the slots may be overwritten
        if (wave < n64) list[tid] = v;
        __syncthreads();
        if (wave < n64 && (wave % (2u * s)) == 0 && wave + s < n64) {
            const unsigned long long o = list[(wave + s) * 64u + (63u - lane)];
            v = wave_merge_desc(v > o ? v : o, lane);
        }
    }
    return v;
}
// wave 0 emits its lanes' composites (descending; 0 = none) as the first k results, padded like sort_and_emit
static __device__ __forceinline__ void emit_top64(unsigned long long v, uint32_t k, uint64_t row_base, uint64_t* out_rows,
                                                  float* out_scores, uint32_t* out_count) {
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x >= 64) return;
    const bool live = v != 0ull && lane < k;
    const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(live));
    if (lane < k) {
        out_rows[lane] = live ? row_base + (uint64_t)(0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull)) : UINT64_MAX;
        out_scores[lane] = live ? key_to_score((uint32_t)(v >> 32)) : u2f(0xFF800000u);
    }
    if (lane == 0) *out_count = cnt;
}

}  // namespace nmn
