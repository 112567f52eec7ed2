// nmn_api.hip — the C ABI of libneumann_gpu.so (declared in include/neumann_gpu.h): shard
// lifecycle, upload, the SIMILAR TOP-K pipeline (search_enqueue: mirror lifecycle, sweep choice, crowd path, retry,
// fallback), the request coalescer of the host-buffer searches (host_submit / host_batch_body: concurrent callers,
// mixed filters, predicates evaluated with the batch), shard merge, synthetic data.  Host code only; every kernel
// lives in nmn_scan / nmn_scan_mfma / nmn_select / nmn_exact / nmn_sortk / nmn_columns / nmn_synth.hip.  No CPU
// compute path exists here: without a HIP device every entry point fails with NMN_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "nmn_index.h"

using namespace nmn;

namespace nmn {
float synth_value_host(uint64_t seed, uint64_t row, uint32_t col);
}

static thread_local std::string g_last_error;

static nmn_status fail_hip(hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    g_last_error = buf;
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) return NMN_ERR_OUT_OF_MEMORY;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
        return NMN_ERR_NO_DEVICE;
    return NMN_ERR_STORAGE;
}
static nmn_status fail_arg(nmn_status code, const char* what) {
    g_last_error = what;
    return code;
}

namespace nmn {
nmn_status set_error(nmn_status code, const char* what) { return fail_arg(code, what); }
nmn_status set_error_hip(hipError_t e, const char* what) { return fail_hip(e, what); }
}  // namespace nmn

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return fail_hip(_e, #expr);    \
    } while (0)

// measurement knobs, read once (README.md lists them)
static bool env_set(const char* name) {
    const char* e = getenv(name);
    return e != nullptr;
}
static bool no_mfma() { static const bool v = env_set("NMN_NO_MFMA"); return v; }
static bool no_half() { static const bool v = env_set("NMN_NO_HALF"); return v; }
static bool no_i8() { static const bool v = env_set("NMN_NO_I8"); return v; }  // A/B: never the 8-bit mirror
static bool no_i8_mfma() { static const bool v = env_set("NMN_NO_I8_MFMA"); return v; }  // A/B: batches stay on the bf16 mirror
static uint64_t i8_min_rows() {  // shards below this stay on the bf16 mirror (the build and its HBM are not worth it)
    static const uint64_t v = [] {
        const char* e = getenv("NMN_I8_MIN_ROWS");
        const long long x = e ? atoll(e) : 4096;
        return (uint64_t)(x > 0 ? x : 0);
    }();
    return v;
}
static bool no_sample() { static const bool v = env_set("NMN_NO_SAMPLE"); return v; }
static bool no_crowd() { static const bool v = env_set("NMN_NO_CROWD"); return v; }
static bool no_tiny() { static const bool v = env_set("NMN_NO_TINY"); return v; }  // A/B: small shards through the 5-launch pipeline
static bool no_grid_select() { static const bool v = env_set("NMN_NO_GRID_SELECT"); return v; }  // A/B: one-workgroup fallback select

// ---- host slots ------------------------------------------------------------------------------------
static bool coalesce_enabled() {
    static const bool on = [] {
        const char* e = getenv("NMN_NO_COALESCE");
        return !(e && e[0] && e[0] != '0');
    }();
    return on;
}

// Searches in flight before a new caller queues instead of starting its own.  A sweep over a large shard is
// HBM-bound: two of them side by side each take twice as long, while a second QUERY in the same sweep is nearly
// free, so callers queue behind ONE running batch and leave together in the next (1M x 768, 64 threads through the
// engine: 4.1k -> 59k queries/s).  Small shards are latency-bound (a dozen launches, a few tens of microseconds):
// two batches in flight keep the device busy while results are handed out (100k x 128: 17.7k -> 114k queries/s).
// NMN_HOST_INFLIGHT overrides (1..kHostSlots).
static int lead_limit(const nmn_index* idx, double selectivity = 1.0) {
    static const int forced = [] {
        const char* e = getenv("NMN_HOST_INFLIGHT");
        const int v = e ? atoi(e) : 0;
        return v < 0 ? 0 : std::min(v, (int)nmn_index::kHostSlots);
    }();
    if (forced) return forced;
    if (!coalesce_enabled()) return nmn_index::kHostSlots;
    // a filtered search reads only what its bitmap selects: several selective ones may run side by side
    const double sweep_bytes = (double)idx->rows * idx->ld * 2.0 * selectivity;
    return sweep_bytes >= (double)(256ull << 20) ? 1 : sweep_bytes >= (double)(32ull << 20) ? 2 : nmn_index::kHostSlots;
}
static double selectivity_of(const nmn_index* idx, const HostReq& r) {
    if (r.pred_cols || !r.mask || r.mask_rows == UINT64_MAX) return 1.0;
    return std::min(1.0, (double)r.mask_rows / (double)std::max<uint64_t>(idx->rows, 1));
}
// In-flight limit when `r` is next to lead.  The extra room a selective filter earns is for light load only: with a
// crowd waiting, ONE leader should take them all (possibly as one mixed-filter sweep) rather than four leaders
// starting four sweeps side by side.
static int lead_limit_for(const nmn_index* idx, const HostReq& r) {
    const int base = lead_limit(idx);
    if (idx->host_queue.size() >= 8) return base;
    return std::max(base, lead_limit(idx, selectivity_of(idx, r)));
}

// How long a batch leader waits for the callers of the previous batch to come back (see last_batch_requests): a
// fraction of the sweep it is about to pay for everybody, only on shards whose sweep is long enough to be worth
// sharing (those that run one batch at a time).  NMN_GATHER_US overrides (0 = never wait).
static uint32_t gather_window_us(const nmn_index* idx) {
    static const int forced = [] {
        const char* e = getenv("NMN_GATHER_US");
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0) return (uint32_t)forced;
    if (!coalesce_enabled() || lead_limit(idx) != 1) return 0;
    const double sweep_us = (double)idx->rows * idx->ld * 2.0 / 5.0e6;  // bytes / (5 TB/s) in microseconds
    return (uint32_t)std::min(200.0, std::max(30.0, 0.08 * sweep_us));
}

// caller holds idx->mu and has checked slots_busy < kHostSlots
static int slot_take(nmn_index* idx) {
    for (int i = 0; i < nmn_index::kHostSlots; i++) {
        if (idx->slot_busy[i]) continue;
        idx->slot_busy[i] = true;
        idx->slots_busy++;
        return i;
    }
    return -1;
}
// caller holds idx->mu: hand free slots to the oldest waiting requests; each becomes the leader of a batch
static void designate_leaders(nmn_index* idx) {
    while (!idx->host_queue.empty() && idx->writers_waiting == 0 &&
           idx->slots_busy < lead_limit_for(idx, *idx->host_queue.front())) {
        HostReq* r = idx->host_queue.front();
        idx->host_queue.erase(idx->host_queue.begin());
        r->slot = slot_take(idx);
        std::lock_guard<std::mutex> g(r->m);
        r->lead = true;
        r->cv.notify_one();  // under r->m: the sleeper cannot return (and destroy *r) before this call is over
    }
}
// Before changing the shard (or using the host stream's workspace exclusively): no host-buffer search in flight and
// none starting.  Declared right after the unique_lock on idx->mu, so it is destroyed while the lock is still held.
struct IdleGuard {
    nmn_index* idx;
    IdleGuard(nmn_index* i, std::unique_lock<std::mutex>& lk) : idx(i) {
        idx->writers_waiting++;
        idx->cv.wait(lk, [&] { return idx->slots_busy == 0; });
    }
    ~IdleGuard() {
        idx->writers_waiting--;
        designate_leaders(idx);
    }
};

static void ws_release_core(Workspace* w);
static void ws_free(Workspace* w) {
    if (!w) return;
    ws_release_core(w);  // everything ws_alloc made (the list lives in ONE place: round 6 found the fallback selection's four buffers missing here)
    void* ptrs[] = {w->h_queries, w->h_mask, w->h_out_rows, w->h_out_scores, w->h_out_counts, w->h_rowlist, w->h_scorelist, w->lk_keys};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (w->h_pack) (void)hipFree(w->h_pack);
    if (w->h_qmasks) (void)hipFree(w->h_qmasks);
    for (void* p : {(void*)w->pred_block, (void*)w->pred_masks, (void*)w->pred_counts, (void*)w->pred_ticket})
        if (p) (void)hipFree(p);
    if (w->tiny_pool) (void)hipFree(w->tiny_pool);
    if (w->tiny_ticket) (void)hipFree(w->tiny_ticket);
    if (w->tiny_out) (void)hipHostFree(w->tiny_out);
    if (w->pin_pred) (void)hipHostFree(w->pin_pred);
    if (w->pin_in) (void)hipHostFree(w->pin_in);
    if (w->pin_out) (void)hipHostFree(w->pin_out);
    for (auto& e : w->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : w->hist)
        if (e) (void)hipEventDestroy(e);
    delete w;
}

template <typename T>
static hipError_t grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) {
        hipError_t e = hipFree(*p);
        *p = nullptr;
        *cap = 0;
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(need, 1) * sizeof(T));
    if (e == hipSuccess) *cap = need;
    return e;
}

static hipError_t grow_pinned(uint8_t** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = std::max<size_t>(need + need / 2, 4096);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(p), want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

constexpr uint32_t kSampleStep = 32;
// (scan_mfma_kernel's skip_sampled arithmetic divides by S - 1 and keeps the sampling pass's tile maxima in groups of four: ADVICE r05)
static_assert(kSampleStep >= 4 && kSampleStep % 4 == 0, "the sampling step must be a multiple of 4");  // sampling pass of the batched sweep: every 32nd tile (3 % of the corpus; 64 / 128 measured the same)

// queries per pipeline pass: bound the score matrix to ~4 GiB
static uint32_t pass_queries(const nmn_index* idx, uint32_t nq) {
    const uint64_t per_q = std::max<uint64_t>(idx->cap_pad, 64) * 4ull;
    uint64_t m = (8ull << 30) / per_q;  // score matrix of one pass <= 8 GiB (128 queries x 10M rows = 5 GiB)
    m = std::max<uint64_t>(1, std::min<uint64_t>(m, 256));
    return (uint32_t)std::min<uint64_t>(m, nq);
}

static nmn_status ws_alloc_core(nmn_index* idx, Workspace* w);
static nmn_status ws_get(nmn_index* idx, hipStream_t stream, uint32_t nq, uint32_t k, Workspace** out) {
    Workspace* w = nullptr;
    auto it = idx->ws.find(stream);
    if (it != idx->ws.end()) w = it->second;
    uint32_t nqc = std::min(pass_queries(idx, nq), idx->ws_nq_limit);
    // k beyond NMN_MAX_TOP_K takes the large-k path, which needs no candidate lists
    uint32_t cand_cap = std::max<uint32_t>(std::min<uint32_t>(idx->cand_cap, NMN_MAX_TOP_K), std::min<uint32_t>(k, NMN_MAX_TOP_K));
    if (w && w->nq_cap >= nqc && w->cand_cap >= cand_cap) {
        *out = w;
        return NMN_OK;
    }
    if (w) {
        // grow: drain the stream, keep the larger of old/new sizes so alternating callers do not thrash
        HIP_TRY(hipStreamSynchronize(stream));
        nqc = std::max(nqc, w->nq_cap);
        cand_cap = std::max(cand_cap, w->cand_cap);
        if (w->allocated && w->cand_cap >= cand_cap) {
            // Only the query pass grows (a larger batch, or an out-of-memory verdict that expired): the larger workspace is tried
            // BEFORE the working one is given up (on a device that is still full the old order freed a multi-GB
            // workspace, failed to get the larger one and came back through the OOM loop — every time the verdict expired).
            Workspace* wn = new (std::nothrow) Workspace();
            if (!wn) return fail_arg(NMN_ERR_OUT_OF_MEMORY, "workspace alloc");
            wn->stream = stream;
            wn->nq_cap = nqc;
            wn->cand_cap = cand_cap;
            const nmn_status st = ws_alloc_core(idx, wn);
            if (st == NMN_OK) {
                wn->allocated = true;
                idx->ws.erase(stream);
                ws_free(w);
                idx->ws[stream] = wn;
                *out = wn;
                return NMN_OK;
            }
            ws_free(wn);
            if (st != NMN_ERR_OUT_OF_MEMORY) return st;
            idx->ws_nq_limit = std::min(idx->ws_nq_limit, w->nq_cap);  // the working one serves, in more passes
            *out = w;
            return NMN_OK;
        }
        idx->ws.erase(stream);
        ws_free(w);
    }
    w = new (std::nothrow) Workspace();
    if (!w) return fail_arg(NMN_ERR_OUT_OF_MEMORY, "workspace alloc");
    w->stream = stream;
    w->nq_cap = nqc;
    w->cand_cap = cand_cap;
    idx->ws[stream] = w;
    *out = w;
    return NMN_OK;
}

// Crowd path (nmn_select.hip, CrowdParams): shards of at least 2^18 rows — below, the exact scan of everything costs
// less than the three extra launches — with a pool of 8M (row, score) pairs per workspace (64 MiB).
constexpr uint64_t kCrowdMinRows = 1ull << 18;
constexpr uint64_t kCrowdPool = 8ull << 20;

// Frees what ws_alloc allocates (and only that), leaving the workspace as ws_get made it.
static void ws_release_core(Workspace* w) {
    void** ptrs[] = {(void**)&w->scores, (void**)&w->tmax, (void**)&w->wmax, (void**)&w->tsample, (void**)&w->skip_key,
                     (void**)&w->k_extra, (void**)&w->qpad, (void**)&w->qi8, (void**)&w->qinfo, (void**)&w->qinfo_f32, (void**)&w->qstate,
                     (void**)&w->cand_rows, (void**)&w->cand_scores, (void**)&w->split_sg, (void**)&w->split_ctr, (void**)&w->final_ticket, (void**)&w->done_ctr, (void**)&w->run_slots, (void**)&w->run_bound, (void**)&w->h_counts2, (void**)&w->crowd_ctr,
                     (void**)&w->crowd_rows, (void**)&w->crowd_scores, (void**)&w->fb_hist, (void**)&w->fb_list, (void**)&w->fb_count,
                     (void**)&w->fb_sync};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    w->crowd_cap = 0;
    for (auto& e : w->ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (auto& e : w->hist) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    w->hist_head = w->hist_read = 0;
    w->allocated = false;
}

static nmn_status ws_alloc_core(nmn_index* idx, Workspace* w) {
    w->ld = idx->ld;
    w->score_stride = idx->cap_pad;
    w->n_tiles_cap = (uint32_t)(idx->cap_pad / kTileRows);
    const size_t nq = w->nq_cap;
    {
        // test hook (tests/test_gpu_edge_cases.py): passes of more than n queries "do not fit" — what a nearly full device does to
        // a 64-query pass over 10M rows (2.6 GB of scores), without having to fill 288 GiB first
        static const long test_max_nq = [] { const char* e = getenv("NMN_WS_TEST_MAX_NQ"); return e ? atol(e) : 0l; }();
        if (test_max_nq > 0 && (long)nq > test_max_nq) return fail_hip(hipErrorOutOfMemory, "workspace (NMN_WS_TEST_MAX_NQ)");
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->scores), std::max<size_t>(nq * w->score_stride, 64) * 4));
    w->tmax_stride = ((uint64_t)w->n_tiles_cap + 3) & ~3ull;  // rows of tmax stay 16-B aligned (uint4 sweeps)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->tmax), std::max<size_t>(nq * w->tmax_stride, 4) * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->wmax), nq * kMaxScanWaves * 4));
    w->n_sample_cap = (w->n_tiles_cap + kSampleStep - 1) / kSampleStep + 4;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->tsample), nq * w->n_sample_cap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->skip_key), nq * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->k_extra), 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->qpad), nq * w->ld * sizeof(float)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->qi8), nq * w->ld * 2));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->qinfo), nq * sizeof(QInfo)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->qinfo_f32), nq * sizeof(QInfo)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->qstate), nq * sizeof(QState)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->cand_rows), nq * w->cand_cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->cand_scores), nq * w->cand_cap * sizeof(float)));
    {
        const size_t nqs = std::min<size_t>(nq, 4);  // (the split selection serves calls of <= 4 queries)
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->split_sg), nqs * 1024 * 4));
        HIP_TRY(hipMemset(w->split_sg, 0, nqs * 1024 * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->split_ctr), nqs * 16));
        HIP_TRY(hipMemset(w->split_ctr, 0, nqs * 16));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->final_ticket), nq * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->done_ctr), 4));
    HIP_TRY(hipMemset(w->done_ctr, 0, 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->run_slots), nq * 1024 * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->run_bound), ((nq + 127) & ~(size_t)127) * 4 + 512));  // (+ a 64-query group's LDS-DMA may start at any multiple of 64)
    HIP_TRY(hipMemset(w->run_bound, 0, ((nq + 127) & ~(size_t)127) * 4 + 512));
    HIP_TRY(hipMemset(w->final_ticket, 0, nq * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->h_counts2), 2 * sizeof(unsigned long long)));
    if (idx->cap_pad >= kCrowdMinRows) {
        // 8M entries, or 128K per query of the pass when that is more (a 128-query batch of crowded queries)
        // (never more than the pass can use: a crowd is at most an eighth of the shard per query)
        w->crowd_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(kCrowdPool, (uint64_t)nq << 17),
                                                    std::max<uint64_t>(idx->cap_pad, (uint64_t)nq * (idx->cap_pad / 8)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->crowd_ctr), (3 + (size_t)kCrowdMaxGrid) * nq * 4));  // count | offset | fill | per-workgroup counts
        HIP_TRY(hipMemset(w->crowd_ctr, 0, (3 + (size_t)kCrowdMaxGrid) * nq * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->crowd_rows), (size_t)w->crowd_cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->crowd_scores), (size_t)w->crowd_cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->fb_hist), (6 * 2048 + 2) * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->fb_list), nq * (size_t)NMN_MAX_TOP_K * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->fb_count), nq * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w->fb_sync), 16));  // arrival counter + abort flag
        HIP_TRY(hipMemset(w->fb_sync, 0, 16));
    }
    for (auto& e : w->ev) HIP_TRY(hipEventCreate(&e));
    return NMN_OK;
}

// All or nothing: a workspace counts as allocated only once EVERY buffer exists.  A failure half way (a nearly full
// HBM, a 128-query pass) frees what it got, so the next search on this stream retries — or fails again with
// NMN_ERR_OUT_OF_MEMORY — instead of launching kernels on null pointers.
static nmn_status ws_alloc(nmn_index* idx, Workspace* w) {
    if (w->allocated) return NMN_OK;
    for (;;) {
        const nmn_status st = ws_alloc_core(idx, w);
        if (st == NMN_OK) break;
        const std::string keep = g_last_error;
        ws_release_core(w);
        g_last_error = keep;
        // Not enough HBM for a pass of this many queries (the score matrix is nq x rows x 4 B: 2.6 GB for 64 queries over 10M
        // rows): the batch runs as more passes of fewer queries instead of failing — remembered for the shard, so that later
        // workspaces do not try the large size again.  One query per pass is the floor.
        if (st != NMN_ERR_OUT_OF_MEMORY || w->nq_cap <= 1) return st;
        w->nq_cap = std::max<uint32_t>(1, w->nq_cap / 4);
        idx->ws_nq_limit = std::min(idx->ws_nq_limit, w->nq_cap);
    }
    w->allocated = true;
    return NMN_OK;
}

// ---- basic entry points -----------------------------------------------------------------------
extern "C" nmn_status nmn_device_count(int32_t* n) {
    if (!n) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "n is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return NMN_OK;
}

extern "C" const char* nmn_status_str(nmn_status s) {
    switch (s) {
        case NMN_OK: return "ok";
        case NMN_ERR_NOT_FOUND: return "Embedding not found";
        case NMN_ERR_DIMENSION_MISMATCH: return "Dimension mismatch";
        case NMN_ERR_EMPTY_VECTOR: return "Empty vector provided";
        case NMN_ERR_INVALID_TOP_K: return "Invalid top_k value (must be > 0)";
        case NMN_ERR_STORAGE: return "Storage error";
        case NMN_ERR_CONFIGURATION: return "Configuration error";
        case NMN_ERR_COLLECTION_EXISTS: return "Collection already exists";
        case NMN_ERR_COLLECTION_NOT_FOUND: return "Collection not found";
        case NMN_ERR_SEARCH_TIMEOUT: return "search timeout";
        case NMN_ERR_IO: return "IO error";
        case NMN_ERR_SERIALIZATION: return "Serialization error";
        case NMN_ERR_INVALID_ARGUMENT: return "invalid argument";
        case NMN_ERR_NO_DEVICE: return "no usable HIP device (libneumann_gpu has no CPU fallback)";
        case NMN_ERR_OUT_OF_MEMORY: return "out of device memory";
        case NMN_ERR_TOP_K_TOO_LARGE: return "top_k too large";
        case NMN_ERR_CAPACITY: return "upload exceeds index capacity";
        case NMN_ERR_BUFFER_TOO_SMALL: return "buffer too small";
        default: return "unknown status";
    }
}

extern "C" const char* nmn_sweep_kind_str(uint32_t kind) {
    static const char* const names[] = {"none", "ring_f32", "valu_f32", "valu_bf16", "valu_i8", "mfma_f32", "mfma_bf16", "mfma_i8", "exact"};
    return kind <= NMN_SWEEP_EXACT ? names[kind] : "unknown";
}

extern "C" const char* nmn_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* nmn_version(void) { return "0.3.0"; }

extern "C" nmn_status nmn_index_create(const nmn_index_desc* d, nmn_index** out) {
    if (!d || !out) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (d->dim == 0) return fail_arg(NMN_ERR_EMPTY_VECTOR, "dim == 0");
    if (d->flags & ~(NMN_INDEX_WIDE_ROWS | NMN_INDEX_NO_SINGLE_LAUNCH)) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "unknown nmn_index_desc.flags bit");
    uint32_t ld = (d->dim + 7u) & ~7u;  // whole 16-byte chunks of the bf16 mirror too: every row length gets it
    {
        // a row length just short of a multiple of 128 (1000, 960, 1500, 3000, ...) is padded up to it when that costs at
        // most 1/8 more bytes per sweep: query batches then take the matrix-core sweep (64-128 queries per sweep instead
        // of 4); the padding is zero in the corpus, the mirror and the queries, and the exact kernels never read it
        // (the next multiple the sweep is built for: 896 -> 1024, 1152 -> 1280, 1408 -> 1536, 2560 -> 3072, ...)
        uint32_t ld128 = (d->dim + 127u) & ~127u;
        while (ld128 <= 4096u && !scan_mfma_supported(ld128, ld128, NMN_METRIC_COSINE)) ld128 += 128u;
        const uint64_t pad_div = (d->flags & NMN_INDEX_WIDE_ROWS) ? 2u : 8u;  // the caller trades bytes for batches
        if (ld128 <= 4096u && ld128 != ld && (uint64_t)(ld128 - d->dim) * pad_div <= d->dim) ld = ld128;
    }
    if ((uint64_t)ld * 4ull > 160ull * 1024ull)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "dim too large: one query must fit the 160 KiB LDS");
    if (d->capacity_rows >= 0xFFFFFFC0ull)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "capacity_rows must be < 2^32 - 64 per shard");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail_arg(NMN_ERR_NO_DEVICE, "no HIP device");
    }
    int dev = d->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail_arg(NMN_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));
    nmn_index* idx = new (std::nothrow) nmn_index();
    if (!idx) return fail_arg(NMN_ERR_OUT_OF_MEMORY, "index alloc");
    idx->dim = d->dim;
    idx->ld = ld;
    idx->cap = d->capacity_rows;
    idx->cap_pad = std::max<uint64_t>((d->capacity_rows + 63) & ~63ull, 64);
    idx->row_base = d->row_base;
    idx->device = dev;
    idx->cand_cap = d->cand_cap ? d->cand_cap : kDefaultCandCap;
    idx->no_single_launch = (d->flags & NMN_INDEX_NO_SINGLE_LAUNCH) != 0;
    auto cleanup = [&](nmn_status st) {
        if (idx->corpus) (void)hipFree(idx->corpus);
        if (idx->norms) (void)hipFree(idx->norms);
        if (idx->inv_norms) (void)hipFree(idx->inv_norms);
        if (idx->max_norm_bits) (void)hipFree(idx->max_norm_bits);
        if (idx->host_stream) (void)hipStreamDestroy(idx->host_stream);
        delete idx;
        return st;
    };
    hipError_t e;
    const size_t corpus_bytes = (size_t)idx->cap_pad * ld * sizeof(float);
    if ((e = hipMalloc(reinterpret_cast<void**>(&idx->corpus), corpus_bytes)) != hipSuccess)
        return cleanup(fail_hip(e, "hipMalloc(corpus)"));
    if ((e = hipMalloc(reinterpret_cast<void**>(&idx->norms), idx->cap_pad * sizeof(float))) != hipSuccess)
        return cleanup(fail_hip(e, "hipMalloc(norms)"));
    if ((e = hipMalloc(reinterpret_cast<void**>(&idx->inv_norms), idx->cap_pad * sizeof(float))) != hipSuccess)
        return cleanup(fail_hip(e, "hipMalloc(inv_norms)"));
    if ((e = hipMalloc(reinterpret_cast<void**>(&idx->max_norm_bits), 4)) != hipSuccess)
        return cleanup(fail_hip(e, "hipMalloc(max_norm)"));
    if ((e = hipStreamCreateWithFlags(&idx->host_stream, hipStreamNonBlocking)) != hipSuccess)
        return cleanup(fail_hip(e, "hipStreamCreate"));
    if ((e = hipMemsetAsync(idx->corpus, 0, corpus_bytes, idx->host_stream)) != hipSuccess ||
        (e = hipMemsetAsync(idx->norms, 0, idx->cap_pad * sizeof(float), idx->host_stream)) != hipSuccess ||
        (e = hipMemsetAsync(idx->inv_norms, 0, idx->cap_pad * sizeof(float), idx->host_stream)) != hipSuccess ||
        (e = hipMemsetAsync(idx->max_norm_bits, 0, 4, idx->host_stream)) != hipSuccess ||
        (e = hipStreamSynchronize(idx->host_stream)) != hipSuccess)
        return cleanup(fail_hip(e, "memset"));
    *out = idx;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_destroy(nmn_index* idx) {
    if (!idx) return NMN_OK;
    (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : idx->ws) ws_free(kv.second);
    idx->ws.clear();
    if (idx->corpus) (void)hipFree(idx->corpus);
    if (idx->half) (void)hipFree(idx->half);
    if (idx->half_err_bits) (void)hipFree(idx->half_err_bits);
    if (idx->half_scratch) (void)hipFree(idx->half_scratch);
    if (idx->half_stats) (void)hipFree(idx->half_stats);
    for (void* p : {(void*)idx->q8, (void*)idx->q8_scale, (void*)idx->q8_vv, (void*)idx->q8_cos, (void*)idx->q8_err_bits, (void*)idx->q8_stats,
                    (void*)idx->q8_l2_hint})
        if (p) (void)hipFree(p);
    if (idx->upload_ev) (void)hipEventDestroy(idx->upload_ev);
    for (auto& e : idx->sweep_ev)
        if (e) (void)hipEventDestroy(e);
    if (idx->norms) (void)hipFree(idx->norms);
    if (idx->inv_norms) (void)hipFree(idx->inv_norms);
    if (idx->max_norm_bits) (void)hipFree(idx->max_norm_bits);
    for (int i = 1; i < nmn_index::kHostSlots; i++)
        if (idx->host_slots[i]) (void)hipStreamDestroy(idx->host_slots[i]);
    if (idx->host_stream) (void)hipStreamDestroy(idx->host_stream);
    delete idx;
    return NMN_OK;
}

extern "C" uint64_t nmn_index_rows(const nmn_index* idx) { return idx ? idx->rows : 0; }
extern "C" uint32_t nmn_index_dim(const nmn_index* idx) { return idx ? idx->dim : 0; }
extern "C" uint32_t nmn_index_row_stride(const nmn_index* idx) { return idx ? idx->ld : 0; }
extern "C" uint64_t nmn_index_row_base(const nmn_index* idx) { return idx ? idx->row_base : 0; }
extern "C" const float* nmn_index_corpus_device(const nmn_index* idx, uint32_t* ld_out) {
    if (!idx) return nullptr;
    if (ld_out) *ld_out = idx->ld;
    return idx->corpus;
}
extern "C" const float* nmn_index_norms_device(const nmn_index* idx) { return idx ? idx->norms : nullptr; }

extern "C" nmn_status nmn_index_set_rows(nmn_index* idx, uint64_t rows) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    if (rows > idx->cap) return fail_arg(NMN_ERR_CAPACITY, "rows > capacity");
    idx->rows = rows;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_set_mirror(nmn_index* idx, int32_t enabled) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    if (enabled < 0 || enabled > 2) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "nmn_index_set_mirror: enabled must be 0, 1 or 2");
    std::lock_guard<std::mutex> g(idx->mu);
    idx->mirror_off = enabled == 0;
    idx->i8_off = enabled == 2;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_hbm_bytes(nmn_index* idx, uint64_t* corpus_bytes, uint64_t* mirror_bytes, uint64_t* per_row_bytes) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    // (an accounting getter, free of side effects since round 6 — ADVICE r05: a monitor polling it used to clear the shard's
    //  out-of-memory verdicts, and every following batch then freed its working multi-GB workspace to retry the larger one on a device
    //  that was still full; nmn_index_retry_declined is the explicit form, the 4096-search expiry the automatic one)
    const uint64_t elems = idx->cap_pad * (uint64_t)idx->ld;
    if (corpus_bytes) *corpus_bytes = elems * 4ull;
    if (mirror_bytes) *mirror_bytes = (idx->q8 ? elems + idx->cap_pad * 12ull : 0ull) + (idx->half ? elems * 2ull : 0ull);
    if (per_row_bytes) *per_row_bytes = idx->cap_pad * 8ull;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_retry_declined(nmn_index* idx) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    idx->q8_failed = idx->half_failed = false;  // the next search that wants a declined mirror asks the device again (one hipMemGetInfo)
    idx->ws_nq_limit = 0xFFFFFFFFu;             // ... and the next batch tries the full pass again (ws_get: the working workspace is kept
    return NMN_OK;                              //     until the larger one exists)
}

extern "C" nmn_status nmn_index_set_timing(nmn_index* idx, int32_t enabled) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    idx->timing = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    return NMN_OK;
}

static hipError_t half_scratch_get(nmn_index* idx, uint64_t rows, float** out);
static void half_scratch_trim(nmn_index* idx);

// A mirror is an optimisation: it must never take the HBM the next workspace, shard or upload needs.  Before allocating one
// the shard asks the driver how much memory is free and leaves the mirror alone unless `bytes` fit with this much to spare
// (1/16 of the device, at least 2 GiB: workspaces of every shard sharing the device — 2.6 GB each for 64-query passes over 10M rows —, crowd pools, result blocks).  Eight
// logical 10M x 768 shards on one 288-GB device (BASELINE config 4 on a single GPU: 246 GB of f32 rows) thus give the first
// few shards their mirror and leave the rest on the f32 sweep.  NMN_MIRROR_RESERVE_MB overrides the reserve.
static bool mirror_fits(size_t bytes) {
    static const long long forced_mb = [] {
        const char* e = getenv("NMN_MIRROR_RESERVE_MB");
        return e ? atoll(e) : -1ll;
    }();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return true;  // (no answer: let hipMalloc decide)
    }
    const size_t reserve = forced_mb >= 0 ? (size_t)forced_mb << 20 : std::max<size_t>(total_b / 16, (size_t)2 << 30);
    const bool fits = free_b >= bytes && free_b - bytes >= reserve;
    static const bool trace = env_set("NMN_TRACE_MIRROR");
    if (trace)
        fprintf(stderr, "[nmn] mirror of %.2f GiB: %.2f of %.2f GiB free, reserve %.2f -> %s\n", bytes / 1073741824.0, free_b / 1073741824.0,
                total_b / 1073741824.0, reserve / 1073741824.0, fits ? "built" : "declined");
    return fits;
}

// The bf16 mirror of the corpus (nmn_index::half) and its bookkeeping.  Not enough HBM is not an error: the shard then
// stays on the f32 sweep (half_failed).  All or nothing: a failure half way frees what it got.
static nmn_status mirror_alloc(nmn_index* idx, hipStream_t stream) {
    if (idx->half || idx->half_failed) return NMN_OK;
    const size_t bytes = (size_t)idx->cap_pad * idx->ld * 2;
    hipError_t e = mirror_fits(bytes) ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->half_err_bits), 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->half_stats), 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->half), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(idx->half_err_bits, 0, 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->half_stats, 0, 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->half, 0, bytes, stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (void** p : {(void**)&idx->half, (void**)&idx->half_err_bits, (void**)&idx->half_stats}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        idx->half_failed = true;
        return NMN_OK;
    }
    idx->half_rows = 0;
    return NMN_OK;
}

// The 8-bit mirror (nmn_index::q8).  Not enough HBM is not an error: the shard stays on the bf16 mirror / the f32 corpus
// (q8_failed).  All or nothing — the small buffers first: a shard whose q8 is set has EVERY q8_* buffer (the sweeps
// dereference all of them), a failure at any step frees all seven and sets q8_failed.
static void q8_release(nmn_index* idx) {
    for (void** p : {(void**)&idx->q8, (void**)&idx->q8_scale, (void**)&idx->q8_vv, (void**)&idx->q8_cos, (void**)&idx->q8_err_bits,
                     (void**)&idx->q8_stats, (void**)&idx->q8_l2_hint}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    idx->q8_rows = 0;
}
static nmn_status q8_alloc(nmn_index* idx, hipStream_t stream) {
    if (idx->q8 || idx->q8_failed) return NMN_OK;
    const size_t bytes = (size_t)idx->cap_pad * idx->ld, per_row = (size_t)idx->cap_pad * 4;
    hipError_t e = mirror_fits(bytes + 3 * per_row) ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_err_bits), 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_l2_hint), 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_stats), 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_scale), per_row);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_vv), per_row);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&idx->q8_cos), per_row);
    int8_t* codes = nullptr;  // (idx->q8 is what "the mirror exists" means: set last)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&codes), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_err_bits, 0, 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_l2_hint, 0, 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_stats, 0, 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(codes, 0, bytes, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_scale, 0, per_row, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_vv, 0, per_row, stream);
    if (e == hipSuccess) e = hipMemsetAsync(idx->q8_cos, 0, per_row, stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (codes) (void)hipFree(codes);
        q8_release(idx);
        idx->q8_failed = true;
        return NMN_OK;
    }
    idx->q8 = codes;
    idx->q8_rows = 0;
    return NMN_OK;
}

// Rows were written: keep the 8-bit mirror current.  Rows it already holds are re-quantized in place; a write that starts inside
// or right behind the rows it holds extends it when `extend` says so (bulk writes: the first search then finds the mirror
// built); otherwise rows beyond it are converted lazily, by the first search that wants them (search_enqueue).
static nmn_status q8_patch(nmn_index* idx, uint64_t row0, uint64_t n, hipStream_t stream, bool extend = false) {
    if (!idx->q8 || row0 > idx->q8_rows || n == 0) return NMN_OK;
    const uint64_t end = extend ? row0 + n : std::min(row0 + n, idx->q8_rows);
    if (end <= row0) return NMN_OK;
    const uint64_t cnt = end - row0;
    float* scratch = nullptr;
    HIP_TRY(half_scratch_get(idx, cnt, &scratch));
    HIP_TRY(launch_q8_rows(idx->corpus, idx->q8, idx->q8_scale, idx->q8_vv, idx->q8_cos, idx->ld, row0, cnt, idx->norms, scratch, idx->q8_err_bits, stream));
    idx->q8_rows = std::max(idx->q8_rows, end);
    if (idx->half_scratch_cap > (1u << 20)) {
        HIP_TRY(hipStreamSynchronize(stream));
        half_scratch_trim(idx);
    }
    return NMN_OK;
}

// Which mirror a shard builds while its rows arrive.  ONE: the 8-bit one (1 B per element: with the f32 rows 5 B per element
// resident) where the row stride lets the 8-bit sweeps serve the shard's searches, else the bf16 one (2 B).  The other is
// built on demand only — by the first search that needs it (a batch on a row length the 8-bit matrix-core sweep does not
// cover; a shard whose 8-bit margin proved useless and that went back to the bf16 mirror, search_enqueue).
static bool ingest_wants_q8(const nmn_index* idx) {
    return !idx->mirror_off && !idx->i8_off && !no_i8() && !idx->q8_failed && idx->cap >= i8_min_rows() &&
           scan_i8_supported(idx->ld, idx->dim, NMN_METRIC_COSINE);
}

// rows the bf16 mirror already holds were overwritten: re-round them in place (re-deriving the mirror "from row0 on" made one
// overwritten row near the top cost a conversion of the whole shard at the next search); rows beyond it are converted lazily
static nmn_status half_patch(nmn_index* idx, uint64_t row0, uint64_t n, hipStream_t stream) {
    if (!idx->half || row0 >= idx->half_rows) return NMN_OK;
    const uint64_t cnt = std::min(row0 + n, idx->half_rows) - row0;
    float* scratch = nullptr;
    HIP_TRY(half_scratch_get(idx, cnt, &scratch));
    HIP_TRY(launch_half_rows(idx->corpus, idx->half, idx->ld, row0, cnt, idx->norms, scratch, idx->half_err_bits, stream));
    if (idx->half_scratch_cap > (1u << 20)) {
        HIP_TRY(hipStreamSynchronize(stream));
        half_scratch_trim(idx);
    }
    return NMN_OK;
}

// What every writer of rows runs behind the copy: magnitudes in reference order, and the shard's mirror of the same rows — in
// ONE read of the new rows where their shape allows it (nmn_ingest.hip): ingest_q8_kernel for a shard whose mirror is the 8-bit
// one (magnitudes, codes, scales, error maxima; rows of whole 128-element halves up to 2048), ingest_kernel for the bf16 one
// (rows of whole 32-float stages).  A bulk write (>= 4096 rows from row 0) allocates the mirror right away so that the first
// search finds it built.  Other shapes: norms_kernel now, the mirror by its own kernels (here for rows it already holds, else
// lazily in search_enqueue).
static nmn_status rows_written(nmn_index* idx, uint64_t row0, uint64_t n, hipStream_t stream) {
    static const bool no_ingest = env_set("NMN_NO_INGEST");  // measurement knob: the separate passes of rounds 1-3
    const bool bulk = n >= 4096 && row0 == 0;
    const bool q8_first = ingest_wants_q8(idx);
    if (bulk && q8_first && !idx->q8) {
        nmn_status st = q8_alloc(idx, stream);
        if (st != NMN_OK) return st;
    }
    if (idx->q8 && row0 <= idx->q8_rows && ingest_q8_supported(idx->ld, idx->dim) && !no_ingest) {
        // the 8-bit mirror stays a prefix of the rows: a write inside or right behind it is quantized by the same read
        HIP_TRY(launch_ingest_q8(idx->corpus, idx->ld, row0, n, idx->norms, idx->inv_norms, idx->max_norm_bits, idx->q8, idx->q8_scale, idx->q8_vv,
                                 idx->q8_cos, idx->q8_err_bits, stream));
        idx->q8_rows = std::max(idx->q8_rows, row0 + n);
        return half_patch(idx, row0, n, stream);  // (a bf16 mirror built on demand beside it)
    }
    // (a write of >= 4096 rows that starts inside or right behind the rows the 8-bit mirror holds extends it)
    const bool q8_extend = idx->q8 && n >= 4096 && row0 <= idx->q8_rows;
    if (ingest_supported(idx->ld, idx->dim) && !no_ingest) {
        const bool want_half = !q8_first && !idx->mirror_off && !no_half() && !idx->half_failed;
        if (want_half && !idx->half && bulk) {
            nmn_status st = mirror_alloc(idx, stream);
            if (st != NMN_OK) return st;
        }
        const bool with_half = idx->half && row0 <= idx->half_rows;  // the mirror stays a prefix of the rows
        HIP_TRY(launch_ingest(idx->corpus, idx->ld, row0, n, idx->norms, idx->inv_norms, idx->max_norm_bits, with_half ? idx->half : nullptr,
                              idx->half_err_bits, stream));
        if (with_half) idx->half_rows = std::max(idx->half_rows, row0 + n);
        return q8_patch(idx, row0, n, stream, q8_extend);
    }
    HIP_TRY(launch_norms(idx->corpus, idx->ld, idx->dim, row0, n, idx->norms, idx->inv_norms, idx->max_norm_bits, stream));
    nmn_status st = half_patch(idx, row0, n, stream);
    if (st != NMN_OK) return st;
    return q8_patch(idx, row0, n, stream, q8_extend);
}

static nmn_status upload_common(nmn_index* idx, const float* src, bool src_is_host, uint64_t row0, uint64_t n,
                                hipStream_t stream) {
    if (!idx || (!src && n)) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (row0 > idx->rows) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "row0 leaves a gap (row0 > rows)");
    if (n > idx->cap || row0 > idx->cap - n) return fail_arg(NMN_ERR_CAPACITY, "row0 + n > capacity_rows");  // (no u64 wrap)
    if (n == 0) return NMN_OK;
    HIP_TRY(hipSetDevice(idx->device));
    float* dst = idx->corpus + row0 * (uint64_t)idx->ld;
    const hipMemcpyKind kind = src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (idx->ld == idx->dim) {
        HIP_TRY(hipMemcpyAsync(dst, src, n * (size_t)idx->dim * sizeof(float), kind, stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)idx->ld * sizeof(float), src, (size_t)idx->dim * sizeof(float),
                                 (size_t)idx->dim * sizeof(float), n, kind, stream));
    }
    nmn_status st = rows_written(idx, row0, n, stream);
    if (st != NMN_OK) return st;
    idx->rows = std::max(idx->rows, row0 + n);
    return NMN_OK;
}

extern "C" nmn_status nmn_index_upload(nmn_index* idx, const float* rows_host, uint64_t row0, uint64_t n) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    nmn_status st = upload_common(idx, rows_host, true, row0, n, idx->host_stream);
    if (st != NMN_OK) return st;
    HIP_TRY(hipStreamSynchronize(idx->host_stream));
    return NMN_OK;
}

// Asynchronous uploads are ordered against later searches on ANY stream by one event: every upload first makes its
// stream wait for the previous upload's event (so the latest event covers all earlier ones), records the event anew
// behind its own work, and search_enqueue makes a stream that has not yet seen that upload wait for it.
static nmn_status upload_fence_record(nmn_index* idx, hipStream_t stream) {
    if (!idx->upload_ev) HIP_TRY(hipEventCreateWithFlags(&idx->upload_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(idx->upload_ev, stream));
    idx->upload_seq++;
    return NMN_OK;
}
static nmn_status upload_fence_wait(nmn_index* idx, Workspace* w, hipStream_t stream) {
    if (w->seen_upload_seq == idx->upload_seq) return NMN_OK;
    if (idx->upload_ev) HIP_TRY(hipStreamWaitEvent(stream, idx->upload_ev, 0));
    w->seen_upload_seq = idx->upload_seq;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_upload_device(nmn_index* idx, const float* rows_dev, uint64_t row0, uint64_t n,
                                              void* stream) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(idx->device));
    if (idx->upload_ev && idx->upload_seq) HIP_TRY(hipStreamWaitEvent(s, idx->upload_ev, 0));  // chain behind earlier uploads
    nmn_status st = upload_common(idx, rows_dev, false, row0, n, s);
    if (st != NMN_OK || n == 0) return st;
    return upload_fence_record(idx, s);
}

// ---- the search pipeline ------------------------------------------------------------------------
// k > NMN_MAX_TOP_K: exact score of every row, full descending sort of the composite keys, first k
// (nmn_sortk.hip).  One query at a time; the reference does the same amount of work for every k.
static nmn_status search_large_k(nmn_index* idx, Workspace* w, const float* queries_dev, uint32_t nq, uint32_t k,
                                 nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows, float* out_scores,
                                 uint32_t* out_counts, hipStream_t stream) {
    const uint64_t n_rows = idx->rows;
    const uint64_t n_sort = largek_sort_len(n_rows);
    if (n_sort > w->lk_keys_cap) HIP_TRY(hipStreamSynchronize(stream));  // the old buffer may still be in use
    HIP_TRY(grow(&w->lk_keys, &w->lk_keys_cap, (size_t)n_sort));
    w->timed = idx->timing;
    w->scan_ev_in_hist = false;
    w->last_nq = nq;
    w->last_rows_scanned = n_rows;
    w->last_elem_bytes = 4;
    w->last_sweep_kind = n_rows ? NMN_SWEEP_EXACT : NMN_SWEEP_NONE;
    w->last_sweep_launches = n_rows ? 1u : 0u;
    w->last_masked = mask_dev != nullptr;
    if (w->timed) HIP_TRY(hipEventRecord(w->ev[0], stream));
    for (uint32_t qa = 0; qa < nq; qa += w->nq_cap) {
        const uint32_t nqc = std::min(w->nq_cap, nq - qa);
        HIP_TRY(launch_qprep(queries_dev + (size_t)qa * idx->dim, nqc, idx->dim, idx->ld, (int)metric, idx->max_norm_bits,
                             w->qpad, w->qinfo, w->qstate, 0, stream));
        for (uint32_t q = 0; q < nqc; q++) {
            const bool first = qa == 0 && q == 0;
            if (w->timed && first) HIP_TRY(hipEventRecord(w->ev[1], stream));
            if (n_rows > 0) {
                ExactScanParams ep{};
                ep.corpus = idx->corpus;
                ep.norms = idx->norms;
                ep.qpad = w->qpad + (size_t)q * idx->ld;
                ep.qinfo = w->qinfo + q;
                ep.qstate = nullptr;
                ep.mask = mask_dev;
                ep.scores = w->scores;  // plain row order: score_at(row, 0, 1) == row
                ep.n_rows = n_rows;
                ep.nql = 1;
                ep.ld = idx->ld;
                ep.dim = idx->dim;
                ep.nq = 1;
                ep.metric = (int)metric;
                HIP_TRY(launch_exact_scan(ep, stream));
            }
            if (w->timed && first) HIP_TRY(hipEventRecord(w->ev[2], stream));
            const size_t o = (size_t)(qa + q);
            HIP_TRY(launch_largek(w->scores, n_rows, w->lk_keys, k, idx->row_base, out_rows + o * k, out_scores + o * k,
                                  out_counts + o, stream, w->fb_hist, w->fb_sync));
        }
    }
    if (w->timed) HIP_TRY(hipEventRecord(w->ev[3], stream));
    return NMN_OK;
}

// scratch of the mirror conversion: persistent up to 1M rows (4 MiB); a bulk build's larger buffer is returned afterwards.
// Caller holds idx->mu and waits for the conversion before anybody else can ask.
static hipError_t half_scratch_get(nmn_index* idx, uint64_t rows, float** out) {
    if (rows > idx->half_scratch_cap) {
        if (idx->half_scratch) (void)hipFree(idx->half_scratch);
        idx->half_scratch = nullptr;
        idx->half_scratch_cap = 0;
        const size_t want = (size_t)std::max<uint64_t>(rows, 4096);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&idx->half_scratch), want * sizeof(float));
        if (e != hipSuccess) return e;
        idx->half_scratch_cap = want;
    }
    *out = idx->half_scratch;
    return hipSuccess;
}
static void half_scratch_trim(nmn_index* idx) {
    if (idx->half_scratch_cap > (1u << 20)) {
        (void)hipFree(idx->half_scratch);
        idx->half_scratch = nullptr;
        idx->half_scratch_cap = 0;
    }
}

// Smallest batch that takes the matrix-core sweep.  Its time does not depend on the number of queries (<= 64) while the
// VALU sweep slows with every query it adds: measured crossover at 3 queries for rows of >= 768 dimensions (10M x 768:
// 3.00 vs 2.69 ms; 5M x 1536: 2.55 vs 2.33 ms), at 5 (= a second VALU sweep) for short rows (10M x 128: 1.30 vs
// 1.62 ms at 3 queries).  NMN_MFMA_MIN_NQ overrides, for measurements.
static uint32_t mfma_min_queries(const nmn_index* idx) {
    static const uint32_t forced = [] {
        const char* e = getenv("NMN_MFMA_MIN_NQ");
        const int x = e ? atoi(e) : 0;
        return (uint32_t)(x >= 1 ? x : 0);
    }();
    if (forced) return forced;
    return idx->dim >= 768 ? 3u : 5u;
}

// qmasks_dev / qmasks_host (both or neither): one bitmap pointer per query — the same nq device pointers as an array in
// device memory (what the matrix-core sweep reads) and in host memory (for the passes that cannot take that sweep:
// those run query by query, each with its own bitmap).  `mask_dev` must be null when they are given.
static nmn_status search_enqueue(nmn_index* idx, Workspace* w, const float* queries_dev, uint32_t nq, uint32_t k,
                                 nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows,
                                 float* out_scores, uint32_t* out_counts, hipStream_t stream,
                                 const uint64_t* const* qmasks_dev = nullptr, const uint64_t* const* qmasks_host = nullptr,
                                 bool short_chain = false, bool nested = false) {
    // nested: a part of a pass that search_enqueue itself split (pairs of a 3-4-query pass on the 8-bit mirror, the queries of a
    // pass with per-query bitmaps that cannot take the matrix-core sweep): the timing events and the history entry belong to the
    // OUTER call — one entry per search, total_ms over the whole of it (every part used to re-record them).
    // short_chain (the host-buffer API, which waits for the answer anyway — host_batch_body): only the launches every search needs,
    // qprep -> sweep -> select -> rescore -> final.  A query whose candidate list overflows is not followed up on the device (crowd
    // kernels, f32 retry sweep, second selection, exact scan of everything, device-wide selection: six launches that return at once
    // in the common case, ~2.5 us each in a dependent chain, more under another stream's sweep): final_kernel reports it as
    // out_counts[q] = 0xFFFFFFFF and the host runs the whole chain for the call again.
    nmn_status st = ws_alloc(idx, w);
    if (st != NMN_OK) return st;
    st = upload_fence_wait(idx, w, stream);  // rows uploaded asynchronously on another stream must have landed
    if (st != NMN_OK) return st;
    if (k > NMN_MAX_TOP_K) {
        if (qmasks_host) {
            for (uint32_t i = 0; i < nq; i++) {
                st = search_large_k(idx, w, queries_dev + (size_t)i * idx->dim, 1, k, metric, qmasks_host[i],
                                    out_rows + (size_t)i * k, out_scores + (size_t)i * k, out_counts + i, stream);
                if (st != NMN_OK) return st;
            }
            return NMN_OK;
        }
        return search_large_k(idx, w, queries_dev, nq, k, metric, mask_dev, out_rows, out_scores, out_counts, stream);
    }
    // A mirror declined (mirror_fits / hipMalloc) or a pass shrunk (ws_alloc) for lack of HBM is a verdict on the device's memory at
    // THAT moment — a sibling shard building, a large workspace alive.  It is re-examined every 4096 searches (and by
    // nmn_index_hbm_bytes): the flags are cleared, the next call that wants the mirror / the larger pass simply tries again, and
    // a device that is still full declines again at the cost of one hipMemGetInfo (the verdicts used to be for life).
    if ((idx->q8_failed || idx->half_failed || idx->ws_nq_limit != 0xFFFFFFFFu) && (++idx->squeeze_calls & 4095u) == 0) {
        idx->q8_failed = idx->half_failed = false;
        idx->ws_nq_limit = 0xFFFFFFFFu;
    }
    const uint64_t n_rows = idx->rows;
    const uint32_t n_tiles = (uint32_t)((n_rows + kTileRows - 1) / kTileRows);
    if (!nested) {
        w->timed = idx->timing;
        w->last_nq = nq;
        w->last_rows_scanned = n_rows;
        w->last_masked = mask_dev != nullptr || qmasks_dev != nullptr;
        w->last_sweep_kind = NMN_SWEEP_NONE;
        w->last_sweep_launches = 0;
        w->scan_ev_in_hist = false;
        if (w->timed == 1) HIP_TRY(hipEventRecord(w->ev[0], stream));
    }
    // the parts of a split pass, timed as ONE sweep entry of the history ring
    auto hist_mark = [&](int which) -> nmn_status {
        hipEvent_t& h = w->hist[2 * (w->hist_head % Workspace::kTimingHistory) + which];
        if (!h) HIP_TRY(hipEventCreate(&h));
        HIP_TRY(hipEventRecord(h, stream));
        if (which == 1) {
            w->hist_head++;
            w->scan_ev_in_hist = true;
        }
        return NMN_OK;
    };
    for (uint32_t qa = 0; qa < nq; qa += w->nq_cap) {
        const uint32_t nqc = std::min(w->nq_cap, nq - qa);
        bool fused_tail = false;   // this pass ends in rescore_final_kernel (set where the rescore is enqueued)
        RescoreParams rp_tail{};
        // The approximate sweep reads a MIRROR of the shard where one serves the call (its measured rounding error is part of
        // the candidate margin; every candidate is re-scored from the f32 rows): the 8-BIT mirror (1 B per element,
        // nmn_scan_i8.hip / nmn_scan_mfma.hip) for 1-2 queries on rows of whole 128-element halves and for batches on the row
        // lengths its matrix-core sweep covers; else the bf16 mirror (2 B; VALU sweep, or the matrix cores from 3-5 queries on);
        // else the f32 corpus.  A shard keeps ONE mirror by default — the one built while its rows arrived (rows_written) — and
        // builds the other only when a call needs it: here, on first use, extended when rows were uploaded since.  Each has
        // its own on/off switch: data whose margin keeps overflowing the candidate lists pays a mirror pass AND an f32 retry
        // per query, so every 256th search the host reads the counters select_kernel keeps and, if more than half of the recent
        // queries were retried, leaves that mirror alone for the next 8192 searches.
        const bool mfma_shape = nqc >= mfma_min_queries(idx) && n_rows > 0 && scan_mfma_supported(idx->ld, idx->dim, (int)metric) &&
                                !no_mfma();
        const bool mirrors_on = n_rows > 0 && !idx->mirror_off;
        static const bool no_i8_masked_mfma = getenv("NMN_NO_I8_MASKED_MFMA") != nullptr;  // (A/B: bitmap batches on the bf16 mirror, as until round 3)
        const bool i8_shape = mfma_shape ? (!((mask_dev || qmasks_dev) && no_i8_masked_mfma) && !no_i8_mfma() &&
                                            scan_mfma_i8_supported(idx->ld, idx->dim, (int)metric))
                                         : (nqc <= 2 && !qmasks_dev && scan_i8_supported(idx->ld, idx->dim, (int)metric));
        const bool i8_enabled = mirrors_on && n_rows >= i8_min_rows() && !idx->i8_off && !no_i8() && !idx->q8_failed;
        // 3-4 queries on rows too short for the matrix cores to pay (mfma_min_queries): one bf16 VALU sweep of four reads 2 B per
        // element — and so do two 8-bit sweeps of two.  A shard that holds the 8-bit mirror and no bf16 one runs the pass as
        // pairs instead of building (and keeping: 2 more bytes per element) a second mirror for it.
        if (!mfma_shape && nqc > 2 && nqc <= 4 && i8_enabled && !idx->half && idx->q8 && !qmasks_dev && !qmasks_host &&
            idx->q8_calls >= idx->q8_off_until && scan_i8_supported(idx->ld, idx->dim, (int)metric)) {
            const bool mark = w->timed && qa == 0 && !nested;
            if (mark && (st = hist_mark(0)) != NMN_OK) return st;
            for (uint32_t i = 0; i < nqc; i += 2) {
                st = search_enqueue(idx, w, queries_dev + (size_t)(qa + i) * idx->dim, std::min(2u, nqc - i), k, metric, mask_dev,
                                    out_rows + (size_t)(qa + i) * k, out_scores + (size_t)(qa + i) * k, out_counts + qa + i, stream, nullptr,
                                    nullptr, short_chain, true);
                if (st != NMN_OK) return st;
            }
            if (mark && (st = hist_mark(1)) != NMN_OK) return st;
            w->last_elem_bytes = 1u;
            continue;
        }
        bool use_i8 = i8_enabled && i8_shape && idx->q8_calls >= idx->q8_off_until;
        if (i8_enabled && i8_shape && idx->q8_stats && (++idx->q8_calls & 255u) == 0 && idx->q8_calls >= idx->q8_off_until) {
            uint32_t now[2] = {0, 0};  // a plain read of two counters other streams may still be adding to: good enough
            if (hipMemcpy(now, idx->q8_stats, 8, hipMemcpyDeviceToHost) == hipSuccess) {
                const uint32_t total = now[0] - idx->q8_seen[0], retried = now[1] - idx->q8_seen[1];
                idx->q8_seen[0] = now[0];
                idx->q8_seen[1] = now[1];
                // (batches on ONE query plane carry a wider margin: when more than an eighth of the recent queries overflowed under
                //  it, the shard goes back to both planes first — and leaves the 8-bit mirror only if that overflows as well)
                if (total >= 64 && idx->one_plane_recent && retried * 8 > total) {
                    idx->one_plane_off_until = idx->q8_calls + 8192;
                } else if (total >= 64 && retried * 2 > total) {
                    idx->q8_off_until = idx->q8_calls + 8192;
                    use_i8 = false;
                }
                idx->one_plane_recent = false;
            } else {
                (void)hipGetLastError();
            }
        }
        if (use_i8) {
            if (!idx->q8) {
                st = q8_alloc(idx, stream);
                if (st != NMN_OK) return st;
                if (!idx->q8) use_i8 = false;  // not enough HBM: the bf16 mirror or the f32 corpus serves
            }
            if (use_i8 && idx->q8_rows < n_rows) {
                const uint64_t cnt = n_rows - idx->q8_rows;
                float* scratch = nullptr;
                HIP_TRY(half_scratch_get(idx, cnt, &scratch));
                hipError_t ce = launch_q8_rows(idx->corpus, idx->q8, idx->q8_scale, idx->q8_vv, idx->q8_cos, idx->ld, idx->q8_rows, cnt, idx->norms, scratch,
                                               idx->q8_err_bits, stream);
                // rare (first search, or rows uploaded since): wait here so that searches enqueued on OTHER streams
                // afterwards may rely on the mirror without cross-stream events
                if (ce == hipSuccess) ce = hipStreamSynchronize(stream);
                half_scratch_trim(idx);
                if (ce != hipSuccess) return fail_hip(ce, "8-bit mirror");
                idx->q8_rows = n_rows;
            }
        }
        bool use_half = !use_i8 && mirrors_on && scan_half_supported(idx->ld, (int)metric) && !idx->half_failed &&
                        (mfma_shape || (!no_half() && idx->half_calls >= idx->half_off_until));
        if (!use_i8 && !mfma_shape && idx->half_stats && (++idx->half_calls & 255u) == 0 && idx->half_calls >= idx->half_off_until) {
            uint32_t now[2] = {0, 0};
            if (hipMemcpy(now, idx->half_stats, 8, hipMemcpyDeviceToHost) == hipSuccess) {
                const uint32_t total = now[0] - idx->half_seen[0], retried = now[1] - idx->half_seen[1];
                idx->half_seen[0] = now[0];
                idx->half_seen[1] = now[1];
                if (total >= 64 && retried * 2 > total) {
                    idx->half_off_until = idx->half_calls + 8192;
                    use_half = false;
                }
            } else {
                (void)hipGetLastError();
            }
        }
        if (use_half) {
            if (!idx->half) {
                st = mirror_alloc(idx, stream);
                if (st != NMN_OK) return st;
                if (!idx->half) use_half = false;  // not enough HBM for the mirror: f32 VALU sweeps serve everything
            }
            if (use_half && idx->half_rows < n_rows) {
                const uint64_t cnt = n_rows - idx->half_rows;
                float* scratch = nullptr;
                HIP_TRY(half_scratch_get(idx, cnt, &scratch));
                hipError_t ce = launch_half_rows(idx->corpus, idx->half, idx->ld, idx->half_rows, cnt, idx->norms, scratch,
                                                 idx->half_err_bits, stream);
                if (ce == hipSuccess) ce = hipStreamSynchronize(stream);  // (as for the 8-bit mirror: other streams may rely on it from now on)
                half_scratch_trim(idx);
                if (ce != hipSuccess) return fail_hip(ce, "bf16 mirror");
                idx->half_rows = n_rows;
            }
        }
        // The matrix-core sweep reads whichever matrix serves the pass: a mirror, or — none built, none allowed, none fitting — the f32
        // rows themselves, rounded to bf16 in registers (nmn_scan_mfma_f32.hip: rows*dim*4 bytes once per 64-128 queries, where the VALU
        // sweep of four queries would read them 16-32 times).  NMN_NO_F32_MFMA=1: the A/B (VALU sweeps of four, as until round 4).
        static const bool no_f32_mfma = env_set("NMN_NO_F32_MFMA");
        const bool use_mfma = mfma_shape && (use_half || use_i8 || !no_f32_mfma);
        const bool mfma_f32 = use_mfma && !use_half && !use_i8;
        if (qmasks_host && !use_mfma) {
            // per-query bitmaps need the matrix-core sweep: this pass runs query by query instead
            const bool mark = w->timed && qa == 0 && !nested;
            if (mark && (st = hist_mark(0)) != NMN_OK) return st;
            for (uint32_t i = 0; i < nqc; i++) {
                st = search_enqueue(idx, w, queries_dev + (size_t)(qa + i) * idx->dim, 1, k, metric, qmasks_host[qa + i],
                                    out_rows + (size_t)(qa + i) * k, out_scores + (size_t)(qa + i) * k, out_counts + qa + i,
                                    stream, nullptr, nullptr, short_chain, true);
                if (st != NMN_OK) return st;
            }
            if (mark && (st = hist_mark(1)) != NMN_OK) return st;
            continue;
        }
        w->last_elem_bytes = use_i8 ? 1u : use_half ? 2u : 4u;  // (nested parts too: what the sweeps of this pass read)
        // A mirror pass whose margin admits more than cand_cap rows must not fall into the exact scan of everything (85 ms
        // for 10M x 1536 Euclidean): large shards get an f32 retry sweep that only runs for the queries that overflowed.
        // 8-bit matrix-core batches under cosine / dot product multiply ONE int8 plane of every query (the sweep is instruction-bound at
        // the package power limit: the second plane's MFMAs were 0.18 of its 1.53 ms at 10M x 768 x 64); the query's coarser rounding is
        // measured and widens the margin like the rows' own.  NMN_I8_TWO_PLANES=1: both planes, the A/B.
        static const bool two_planes = env_set("NMN_I8_TWO_PLANES");
        // Where it pays (profiles/r06p_*): cosine, k <= 128 — 10M x 768: 64 queries 1.53 -> 1.38-1.48 ms, 128 queries 2.29 -> 1.80; the
        // candidate lists grow (~900 -> ~4 000 rows for the worst query of a batch at 10M rows), which k = 1000 (1.75 -> 1.90 ms) and the
        // dot product's absolute margin (5M x 1536: 1.38 -> 1.51) do not repay.
        // ... nor does a 64-query batch of a PIPELINED caller: with two batches in flight on two streams the longer candidate lists
        // (~590 MB of exact re-scoring reads per batch instead of ~170) run under the next batch's sweep and cost it what the sweep
        // gained (bench.py's c3_i8_*: 39.1 k -> 37.7 k q/s; 128-query passes still gain 11 %: profiles/r06t_*).  "Pipelined" = the shard's
        // previous large sweep was enqueued on another stream (the sweep chain's own bookkeeping).
        const bool pipelined = idx->sweep_seq != 0 && idx->sweep_stream != stream;
        const bool one_plane = use_i8 && use_mfma && !two_planes && metric == NMN_METRIC_COSINE && k <= 128u && (nqc > 64u || !pipelined) &&
                               scan_mfma_i8_one_plane_supported((int)metric) && idx->q8_calls >= idx->one_plane_off_until;
        if (one_plane) idx->one_plane_recent = true;
        const bool f32_retry = (use_half || use_i8) && !use_mfma && n_rows >= (1u << 18) && !short_chain;
        // The batched sweep as ONE launch (round 6; ScanParams::run_*): the bound that gates its score stores rises inside the sweep
        // itself — no sampling pass, no bound kernels, no split into two launches.  Unmasked batches on shards large enough for the
        // sampled bound it replaces (>= 32 768 tiles), k <= 256 (slots per query: the power of two >= k, at least 128).
        // NMN_NO_RUN_BOUND=1: the sampling pass + refinement, the A/B.
        static const bool no_run_bound = env_set("NMN_NO_RUN_BOUND");
        static const bool wgs_pinned = getenv("NMN_MFMA_WGS") != nullptr;  // (the running maxima are one per workgroup of the DEFAULT 1024)
        const bool run_bound = use_mfma && !no_run_bound && !no_sample() && !mask_dev && !qmasks_dev && k <= 256u && nqc <= 64u && !wgs_pinned &&
                               idx->ld <= 1536u &&  // (longer rows: wave pairs per query group, whose LDS has no room for the refresh buffers)
                               n_tiles >= 1024u * kSampleStep && n_tiles / kSampleStep >= 4u * k;
        const uint32_t run_S = run_bound ? k : 0u;  // (the rank of the bound among the <= 1024 workgroups' running maxima)
        // (one launch prepares the query for the mirror sweep and, in qinfo_f32, for the f32 retry behind it)
        HIP_TRY(launch_qprep(queries_dev + (size_t)qa * idx->dim, nqc, idx->dim, idx->ld, (int)metric,
                             idx->max_norm_bits, w->qpad, w->qinfo, w->qstate,
                             use_i8 ? (1 | 2 | 4 | (use_mfma ? (one_plane ? 16 : 0) : 8)) : ((use_mfma ? 1 : 0) | ((use_half || mfma_f32) ? 2 : 0)), stream,
                             // (f32 rows rounded to bf16 on the fly: the rounding is the mirror's — its measured error norms where a
                             //  complete bf16 mirror happens to exist, else the a-priori bound |e_r| <= 2^-8 |v_r| of qprep_kernel)
                             use_i8 ? idx->q8_err_bits : (use_half || (mfma_f32 && idx->half && idx->half_rows >= n_rows)) ? idx->half_err_bits : nullptr,
                             use_i8 ? w->qi8 : nullptr,
                             (use_i8 && !use_mfma) ? idx->q8_l2_hint : nullptr, f32_retry ? w->qinfo_f32 : nullptr,
                             run_bound ? w->run_slots : nullptr, run_S, run_bound ? w->run_bound : nullptr));
        if (n_rows > 0) {
            ScanParams sp{};
            sp.corpus = idx->corpus;
            sp.corpus_half = use_half ? idx->half : nullptr;
            sp.corpus_i8 = use_i8 ? idx->q8 : nullptr;
            sp.i8_one_plane = one_plane ? 1u : 0u;
            sp.i8_scale = idx->q8_scale;
            sp.i8_vv = idx->q8_vv;
            sp.i8_cos = idx->q8_cos;
            sp.qi8 = w->qi8;
            sp.norms = idx->norms;
            sp.inv_norms = idx->inv_norms;
            sp.qpad = w->qpad;
            sp.qinfo = w->qinfo;
            sp.mask = mask_dev;
            sp.qmasks = qmasks_dev ? qmasks_dev + qa : nullptr;
            sp.scores = w->scores;
            sp.tmax = w->tmax;
            sp.wmax = w->wmax;
            sp.tmax_stride = w->tmax_stride;
            sp.wmax_stride = kMaxScanWaves;
            sp.n_rows = n_rows;
            sp.nql = nqc;
            sp.ld = idx->ld;
            sp.n_tiles = n_tiles;
            sp.nq = nqc;
            // ~16 waves per CU; every wave gets the same number of tiles
            static const uint32_t wave_target = [] {  // tuning knob: NMN_SCAN_WAVES in [256, kMaxScanWaves]
                const char* e = getenv("NMN_SCAN_WAVES");
                long v = e ? atol(e) : (long)kMaxScanWaves;
                return (uint32_t)std::min<long>(std::max<long>(v, 256), (long)kMaxScanWaves);
            }();
            sp.tiles_per_wave = std::max<uint32_t>(1, (n_tiles + wave_target - 1) / wave_target);
            // Sweeps of 0.125 .. 2 GiB (1M x 768 on the 8-bit mirror: 0.13 ms) run on 768 waves — 192 workgroups, three quarters of
            // the CUs — instead of 4096: the sweep is HBM-bound either way (0.1279 -> 0.1298 ms), and the CUs and wave slots it
            // leaves free are where the OTHER stream's selection / rescore tail runs while it streams.  Under 4096 persistent
            // waves that tail waited for slots (profiles/r04z_gantt_2streams.txt) and a pipelined caller got 0.158 ms per step out
            // of a 0.128-ms sweep; with 768: 0.1335 ms, 6 330 -> 7 490 q/s at 1M x 768, 3 700 -> 4 255 at 2M
            // (profiles/r04s_scan_waves_by_rows.txt; from 4M rows on 4096 waves win again — there the sweep chain hides the tail; the
            //  band's edges: 200 k x 768 +2 %, 300 k +5 %, 2.5M +11 %, 3M +2.5 %, 100 k -3 %).
            static const bool waves_pinned = getenv("NMN_SCAN_WAVES") != nullptr;
            static const uint32_t small_waves = [] {  // (A/B knob: NMN_SCAN_WAVES_SMALL, 0 = off)
                const char* e = getenv("NMN_SCAN_WAVES_SMALL");
                const long v = e ? atol(e) : 768;
                return v <= 0 ? 0u : (uint32_t)std::min<long>(std::max<long>(v, 256), (long)kMaxScanWaves);
            }();
            {
                const uint64_t sweep_bytes = (uint64_t)n_rows * idx->ld * w->last_elem_bytes;
                if (!waves_pinned && small_waves && !use_mfma && !mask_dev && sweep_bytes >= (128ull << 20) && sweep_bytes < (2ull << 30))
                    sp.tiles_per_wave = std::max<uint32_t>(1, (n_tiles + small_waves - 1) / small_waves);
            }
            // workgroups of the MFMA sweep: 4 per CU in sequence (one resident at a time) evens out the CUs' finish
            // times; measured 256 -> 1024: -2 % sweep time, 2048: worse (ring ramp-up per workgroup).  Knob: NMN_MFMA_WGS
            static const uint32_t mfma_wgs = [] {
                const char* e = getenv("NMN_MFMA_WGS");
                long v = e ? atol(e) : 1024;
                return (uint32_t)std::min<long>(std::max<long>(v, 64), (long)kMaxScanWaves);
            }();
            if (use_mfma) sp.tiles_per_wave = std::max<uint32_t>(1, (n_tiles + mfma_wgs - 1) / mfma_wgs);  // per workgroup
            auto launch_batch_sweep = [&](const ScanParams& x) -> hipError_t { return launch_scan_mfma(x, stream); };
            // One unmasked query over the f32 rows of a large shard (no mirror serves the call — the headline sweep of SURVEY §8(d)):
            // the ring sweep (nmn_scan_ring.hip: f32 arithmetic, the rows through the LDS-DMA ring) instead of scan_kernel's register
            // loads.  Workgroups as on the matrix-core path.  NMN_NO_RING=1: the A/B.
            static const bool no_ring = getenv("NMN_NO_RING") != nullptr;
            const bool use_ring = !no_ring && !use_mfma && !use_i8 && !use_half && nqc == 1 && !mask_dev && !qmasks_dev && n_tiles >= 4096u &&
                                  metric != NMN_METRIC_SPARSE_COSINE_F64 && scan_ring_supported(idx->ld, idx->dim, (int)metric);
            // (4096 workgroups, sixteen per CU in sequence: 0.836 of peak against 0.824-0.829 with 1024 and 0.80 with 256 at 10M x 768,
            //  profiles/r05z9_*; knob NMN_RING_WGS)
            static const uint32_t ring_wgs = [] {
                const char* e = getenv("NMN_RING_WGS");
                long v = e ? atol(e) : (long)kMaxScanWaves;
                return (uint32_t)std::min<long>(std::max<long>(v, 64), (long)kMaxScanWaves);
            }();
            if (use_ring) sp.tiles_per_wave = std::max<uint32_t>(1, (n_tiles + ring_wgs - 1) / ring_wgs);  // per workgroup
            sp.metric = (int)metric;
            sp.strided = (!use_mfma && mask_dev) ? 1u : 0u;  // masked VALU sweeps: a wave takes every W-th tile (runs of selected rows spread over all waves)
            static const bool no_walk = getenv("NMN_NO_WALK") != nullptr;  // (A/B switch of the survivor walk)
            sp.walk = (sp.strided && !no_walk) ? 1u : 0u;
            sp.tile_step = 1;
            sp.skip_key = nullptr;
            // Batched sweep on a large shard: a sampling pass over every 32nd tile (tile maxima only) bounds
            // the k-th best score of each query from below, so the main sweep writes scores only for the few
            // tiles that can still hold a candidate.
            // (NMN_SAMPLE_STEP = 64 / 128: a coarser sample for the A/B — never finer than kSampleStep, which sizes the buffers)
            // (Tried in round 4: the pass — 3 % of the mirror's bytes and a one-workgroup-per-query bound kernel, 51 + 24 us of mostly
            //  ramp-up and latency — enqueued AHEAD of the sweep chain's wait, to run under the previous batch's main sweep.  It takes
            //  from that sweep what it saves its own: 64 queries 1.678 vs 1.679 ms per step, 128 queries 2.55-2.59 vs 2.49-2.50
            //  (profiles/r04h_sampling_pass_ahead_of_chain_ab.txt).  NMN_SAMPLE_AHEAD_OF_CHAIN=1 keeps the A/B.)
            static const uint32_t sample_step = [] {
                const char* e = getenv("NMN_SAMPLE_STEP");
                const long v = e ? atol(e) : 0;
                return (v == 64 || v == 128 || v == 256) ? (uint32_t)v : kSampleStep;
            }();
            static const bool sample_behind = getenv("NMN_SAMPLE_AHEAD_OF_CHAIN") == nullptr;
            const uint32_t n_sample = (n_tiles + sample_step - 1) / sample_step;
            static const uint32_t run_dbg = [] { const char* e = getenv("NMN_RUN_BOUND_DEBUG"); return e ? (uint32_t)atol(e) : 0u; }();  // (measurement only)
            const bool sample = use_mfma && (!run_bound || (run_dbg & 1u)) && n_sample >= 4u * k && n_sample >= 1024u && !no_sample();
            if (run_bound) {
                sp.run_slots = w->run_slots;
                sp.run_bound = w->run_bound;
                sp.run_S = run_S;
                sp.run_dbg = run_dbg;
            }
            auto sampling_pass = [&]() -> nmn_status {
                ScanParams ss = sp;
                ss.run_S = 0;
                ss.tile_step = sample_step;
                ss.n_tiles = n_sample;
                ss.tiles_per_wave = std::max<uint32_t>(1, (n_sample + 255) / 256);
                ss.tmax = w->tsample;
                ss.tmax_stride = w->n_sample_cap;
                // the sampled tiles are FINISHED by this pass (maxima into the sweep's tmax, scores written: 1/32 of the score matrix,
                // 80 MB at 10M rows x 64 queries) and the main sweep does not stream them again: 3 % fewer bytes per batch
                // (NMN_SAMPLE_REREAD=1: the A/B — tile maxima only, every tile read again by the main sweep, as until round 4)
                // Not on the 8-bit mirror: there the pass reads 0.24 GB and the stores of its 80 MB cost as much as reading them again saves
                // (10M x 768, 64 queries: f32 rows 5.28 -> 5.12 ms, bf16 mirror 2.77 -> 2.65, 8-bit 1.538 -> 1.548; profiles/r05d_*).
                static const bool reread_env = getenv("NMN_SAMPLE_REREAD") != nullptr;
                const bool reread = reread_env || use_i8;
                if (!reread) {
                    ss.tmax_main = w->tmax;
                    ss.tmax_main_stride = w->tmax_stride;
                }
                HIP_TRY(launch_batch_sweep(ss));
                HIP_TRY(launch_sample_bound(w->tsample, w->n_sample_cap, n_sample, w->qinfo, nqc, k, w->skip_key, stream));
                sp.skip_key = w->skip_key;
                sp.skip_sampled = reread ? 0u : sample_step;
                return NMN_OK;
            };
            if (sample && !sample_behind) {
                st = sampling_pass();
                if (st != NMN_OK) return st;
            }
            // the sweep chain (nmn_index.h): on a large shard this sweep starts when the previous search's sweep — enqueued on
            // another stream — has ended; its own tail then runs under the next sweep.  NMN_NO_SWEEP_CHAIN=1: the A/B.
            static const bool no_chain = getenv("NMN_NO_SWEEP_CHAIN") != nullptr;
            const bool chain = !no_chain && (uint64_t)n_rows * idx->ld * w->last_elem_bytes >= (2ull << 30);  // (sweeps of >= ~0.35 ms: below, the two
            // event packets of the chain cost more than the overlap of two short sweeps — 1M x 768: 5.8 k q/s without, 5.6 k with)
            if (chain && idx->sweep_seq && idx->sweep_stream != stream)
                HIP_TRY(hipStreamWaitEvent(stream, idx->sweep_ev[(idx->sweep_seq - 1) & 3u], 0));
            if (w->timed && qa == 0 && !nested) {
                hipEvent_t& h = w->hist[2 * (w->hist_head % Workspace::kTimingHistory)];
                if (!h) HIP_TRY(hipEventCreate(&h));
                HIP_TRY(hipEventRecord(h, stream));
            }
            if (sample && sample_behind) {
                st = sampling_pass();
                if (st != NMN_OK) return st;
            }
            // More than 64 queries per pass: the score stores of the sweep are what its epilogue costs (a written tile is four
            // store instructions per query group with one query's lanes active, each holding the wave's issue slot behind the
            // DMA pieces: 0.24 of 2.98 ms at 128 queries, nothing at 64).  The sampled bound sits at about rank 32 k (every 32nd
            // tile); the tile maxima of the sweep's own first quarter give one at about rank 4 k.  So the sweep runs as two
            // launches over workgroup ranges with that refinement in between: ~8x fewer written tiles in the last three quarters
            // for one more launch (measured: DESIGN.md 3.2).  NMN_NO_REFINE=1: one launch (A/B).
            static const bool no_refine = getenv("NMN_NO_REFINE") != nullptr;
            const uint32_t mfma_blocks = use_mfma ? (n_tiles + sp.tiles_per_wave - 1) / sp.tiles_per_wave : 0;
            // the first launch is ONE round of workgroups (one per CU: the sweep's 1024 workgroups are four rounds, and a launch
            // that ends with a nearly empty round pays for a whole one: 248 + 774 workgroups measured +0.4 ms)
            static const uint32_t n_cu = [] {
                int dev = 0, n = 256;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
                return (uint32_t)n;
            }();
            const uint32_t first_blocks = mfma_blocks >= 3u * n_cu ? (n_cu & ~7u) : 0u;  // (a multiple of 8: the folded grid's tile ranges)
            static const uint32_t refine_min_nq = [] {  // (A/B knob: NMN_REFINE_MIN_NQ)
                const char* e = getenv("NMN_REFINE_MIN_NQ");
                return e ? (uint32_t)atol(e) : 65u;
            }();
            // (the 8-bit sweep's wider margin makes ~11 % of the (tile, query) pairs write their scores under the sampled bound —
            //  0.18 of its 1.55 ms at 64 queries — so it takes the refinement from 3 queries on: 10M x 768 1.53 -> 1.50 ms,
            //  5M x 1536 Euclidean 1.55 -> 1.39 ms)
            // (the f32-rows sweep rounds rows AND margin a priori — 2.5x the bf16 mirror's measured margin — and takes it from 3 on as well:
            //  10M x 768, 64 queries 5.24-5.29 -> 4.96 ms, profiles/r05c_f32_mfma_knobs_ab.txt)
            const uint32_t refine_from = ((use_i8 || mfma_f32) && !getenv("NMN_REFINE_MIN_NQ")) ? 3u : refine_min_nq;
            const bool refine = use_mfma && sample && nqc >= refine_from && !no_refine && first_blocks >= 64 && (uint64_t)first_blocks * sp.tiles_per_wave >= 4ull * k;
            if (qa == 0) {  // what nmn_search_stats reports: the dispatch decision itself (first pass of the call; nested parts: the parts' kernel)
                w->last_sweep_kind = use_mfma ? (use_i8 ? NMN_SWEEP_MFMA_I8 : use_half ? NMN_SWEEP_MFMA_BF16 : NMN_SWEEP_MFMA_F32)
                                              : use_i8 ? NMN_SWEEP_VALU_I8 : use_ring ? NMN_SWEEP_RING_F32 : use_half ? NMN_SWEEP_VALU_BF16 : NMN_SWEEP_VALU_F32;
                w->last_sweep_launches = (sample ? 2u : 0u) + (refine ? 3u : 1u);
            }
            if (refine) {
                ScanParams sa = sp;
                sa.bx_base = 0;
                sa.bx_count = first_blocks;
                HIP_TRY(launch_batch_sweep(sa));
                HIP_TRY(launch_sample_bound(w->tmax, w->tmax_stride, first_blocks * sp.tiles_per_wave, w->qinfo, nqc, k, w->skip_key, stream, 1));
                ScanParams sb = sp;
                sb.bx_base = first_blocks;
                sb.bx_count = 0;
                HIP_TRY(launch_batch_sweep(sb));
            } else {
                HIP_TRY(use_mfma ? launch_batch_sweep(sp) : use_i8 ? launch_scan_i8(sp, stream) : use_ring ? launch_scan_ring(sp, stream) : launch_scan(sp, stream));  // (use_mfma && use_i8: sp.corpus_i8 selects the 8-bit form)
            }
            if (w->timed && qa == 0 && !nested) {
                hipEvent_t& h = w->hist[2 * (w->hist_head % Workspace::kTimingHistory) + 1];
                if (!h) HIP_TRY(hipEventCreate(&h));
                HIP_TRY(hipEventRecord(h, stream));
                w->hist_head++;
                w->scan_ev_in_hist = true;
            }
            if (chain) {
                hipEvent_t& ce = idx->sweep_ev[idx->sweep_seq & 3u];
                if (!ce) HIP_TRY(hipEventCreateWithFlags(&ce, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(ce, stream));
                idx->sweep_seq++;
                idx->sweep_stream = stream;
            }

            SelectParams sel{};
            sel.scores = w->scores;
            sel.tmax = w->tmax;
            sel.wmax = w->wmax;
            sel.tmax_stride = w->tmax_stride;
            sel.wmax_stride = kMaxScanWaves;
            sel.tiles_per_wave = sp.tiles_per_wave;
            sel.strided = sp.strided;
            sel.n_waves = (n_tiles + sp.tiles_per_wave - 1) / sp.tiles_per_wave;
            sel.qinfo = w->qinfo;
            sel.qstate = w->qstate;
            sel.cand_rows = w->cand_rows;
            sel.n_rows = n_rows;
            sel.nql = nqc;
            sel.n_tiles = n_tiles;
            sel.nq = nqc;
            sel.k = k;
            sel.cand_cap = w->cand_cap;
            sel.skip_key = (run_bound && !(run_dbg & 1u)) ? w->run_bound : sp.skip_key;  // (the FINAL value of the running bound: every tile at or above it was written)
            sel.k_extra = nullptr;
            sel.retry = 0;
            sel.retry_follows = f32_retry ? 1 : 0;
            sel.half_stats = f32_retry ? (use_i8 ? idx->q8_stats : idx->half_stats) : (use_i8 && use_mfma) ? idx->q8_stats : nullptr;
            sel.count_overflows = (use_i8 && use_mfma) ? 1 : 0;
            if (short_chain && !use_mfma && (use_i8 || use_half) && n_rows >= (1u << 18)) {
                // (no retry selection whose launches could be counted: the mirror's on/off switch counts this selection's overflows)
                sel.half_stats = use_i8 ? idx->q8_stats : idx->half_stats;
                sel.count_overflows = 1;
            }
            sel.l2_hint = (use_i8 && !use_mfma && metric == NMN_METRIC_EUCLIDEAN) ? idx->q8_l2_hint : nullptr;
            sel.split_sg = w->split_sg;    // (launch_select decides whether the query's selection is split over several workgroups)
            sel.split_ctr = w->split_ctr;
            sel.fb_sync_reset = w->fb_sync;  // (nullable) zeroed for the device-wide fallback selection further down this stream
            if (metric == NMN_METRIC_SPARSE_COSINE_F64) {
                HIP_TRY(launch_count_untrusted(idx->norms, n_rows, w->k_extra, stream));
                sel.k_extra = w->k_extra;
            }
            const bool crowd = w->crowd_cap != 0 && n_rows >= kCrowdMinRows && !no_crowd() && !short_chain;
            // (short chain: the selection still hands over early when thousands of tiles are within the margin — nobody follows on
            //  the device, final_kernel flags the query and the host's second pass brings the crowd kernels)
            sel.crowd_follows = (crowd || (short_chain && n_rows >= kCrowdMinRows)) ? 1 : 0;
            sel.crowd_count_reset = crowd ? w->crowd_ctr : nullptr;
            HIP_TRY(launch_select(sel, stream));
            if (crowd) {  // three launches that return at once unless a candidate list overflowed
                CrowdParams cp{};
                cp.qstate = w->qstate;
                cp.tmax = w->tmax;
                cp.tmax_stride = w->tmax_stride;
                cp.scores = w->scores;
                cp.nql = nqc;
                cp.nq = nqc;
                cp.n_tiles = n_tiles;
                cp.n_rows = n_rows;
                cp.count = w->crowd_ctr;
                cp.offset = w->crowd_ctr + w->nq_cap;
                cp.fill = w->crowd_ctr + 2 * (size_t)w->nq_cap;
                cp.wg_count = w->crowd_ctr + 3 * (size_t)w->nq_cap;
                cp.pool_rows = w->crowd_rows;
                cp.pool_scores = w->crowd_scores;
                cp.pool_cap = w->crowd_cap;
                HIP_TRY(launch_crowd_collect(cp, stream));
            }
            if (f32_retry) {  // both launches return at once for queries whose first selection did not overflow
                ScanParams sr = sp;
                sr.corpus_half = nullptr;
                sr.qinfo = w->qinfo_f32;
                sr.retry_state = w->qstate;
                HIP_TRY(launch_scan(sr, stream));
                SelectParams sel2 = sel;
                sel2.qinfo = w->qinfo_f32;
                sel2.retry = 1;
                sel2.retry_follows = 0;
                sel2.fb_sync_reset = nullptr;
                sel2.l2_hint = nullptr;
                sel2.crowd_follows = 0;
                sel2.crowd_count_reset = nullptr;
                HIP_TRY(launch_select(sel2, stream));
            }

            RescoreParams rp{};
            rp.corpus = idx->corpus;
            rp.norms = idx->norms;
            rp.qpad = w->qpad;
            rp.qinfo = w->qinfo;
            rp.qstate = w->qstate;
            rp.cand_rows = w->cand_rows;
            rp.cand_scores = w->cand_scores;
            rp.mask = mask_dev;       // fallback duty of the same launch
            rp.qmasks = qmasks_dev ? qmasks_dev + qa : nullptr;
            rp.crowd_offset = w->crowd_ctr ? w->crowd_ctr + w->nq_cap : nullptr;
            rp.crowd_rows = w->crowd_rows;
            rp.crowd_scores = w->crowd_scores;
            rp.scores = w->scores;
            rp.n_rows = n_rows;
            rp.nql = nqc;
            rp.ld = idx->ld;
            rp.dim = idx->dim;
            rp.nq = nqc;
            rp.cand_cap = w->cand_cap;
            rp.metric = (int)metric;
            // the exact score of EVERY row for flagged queries: a lane per row streaming through LDS (nmn_ingest.hip) instead of the
            // eight-lanes-per-row form of rescore_kernel's fallback duty — Euclidean's sequential sum 11 -> 4.7 ms per 10M x 768,
            // dot / cosine 5.2 -> 4.7 (NMN_NO_EXACT_ROWS=1: the old form, for the A/B)
            static const bool no_exact_rows = getenv("NMN_NO_EXACT_ROWS") != nullptr;
            const bool exact_rows = !no_exact_rows && n_rows >= (1u << 18) && exact_rows_supported(idx->ld, idx->dim, (int)metric) && !short_chain;  // (small shards: not worth a launch)
            rp.skip_fallback = (exact_rows || short_chain) ? 1 : 0;  // (short chain: flagged queries are the host's to follow up)
            // The short chain CAN end in one launch — the last workgroup to finish a query's candidates sorts and emits them
            // (rescore_final_kernel, nmn_exact.hip; NMN_FUSED_TAIL=1) — but it measured SLOWER than the two launches it replaces: the
            // exact kernel needs 214 registers, so the fused workgroup is 512 threads and ranks two entries per thread — 27 us against
            // 8 + 9 at 1M x 768 (~600 candidates; 33 us on 1024 threads with spills): profiles/r06d_*.  Off by default; exact either way.
            static const bool fused_tail_on = env_set("NMN_FUSED_TAIL");
            fused_tail = short_chain && fused_tail_on;
            if (!fused_tail) HIP_TRY(launch_rescore(rp, stream));
            rp_tail = rp;
            if (exact_rows)  // (returns at once unless a query of the pass is flagged)
                HIP_TRY(launch_exact_rows(idx->corpus, idx->norms, idx->ld, n_rows, w->qpad, w->qinfo, w->qstate, 1, mask_dev,
                                          qmasks_dev ? qmasks_dev + qa : nullptr, w->scores, nqc, nqc, (int)metric, stream));
            if (crowd && w->fb_hist && k <= NMN_MAX_TOP_K && !no_grid_select()) {
                // queries still flagged now hold the exact score of every row: select their top-k with the whole device
                // (returns at once when none is flagged)
                FallbackParams fb{};
                fb.qstate = w->qstate;
                fb.scores = w->scores;
                fb.nql = nqc;
                fb.nq = nqc;
                fb.k = k;
                fb.n_rows = n_rows;
                fb.ghist = w->fb_hist;
                fb.list = w->fb_list;
                fb.list_count = w->fb_count;
                fb.sync = w->fb_sync;
                HIP_TRY(launch_fallback_select(fb, stream));
            }
        }
        if (n_rows == 0) fused_tail = false;
        FinalParams fp{};
        fp.fb_list = w->fb_list;
        fp.fb_count = w->fb_count;
        fp.crowd_offset = w->crowd_ctr ? w->crowd_ctr + w->nq_cap : nullptr;
        fp.crowd_rows = w->crowd_rows;
        fp.crowd_scores = w->crowd_scores;
        fp.cand_rows = w->cand_rows;
        fp.cand_scores = w->cand_scores;
        fp.qstate = w->qstate;
        fp.scores = w->scores;
        fp.nql = nqc;
        fp.n_rows = n_rows;
        fp.row_base = idx->row_base;
        fp.nq = nqc;
        fp.k = k;
        fp.cand_cap = w->cand_cap;
        fp.short_chain = short_chain ? 1 : 0;
        fp.out_rows = out_rows + (size_t)qa * k;
        fp.out_scores = out_scores + (size_t)qa * k;
        fp.out_counts = out_counts + qa;
        if (w->poll_word_dev && !nested && qa + w->nq_cap >= nq && !fused_tail) {  // (the call's LAST launch publishes: host_batch_body polls)
            fp.done_word = w->poll_word_dev;
            fp.done_ctr = w->done_ctr;
            fp.done_seq = w->done_seq;
        }
        if (fused_tail) HIP_TRY(launch_rescore_final(rp_tail, fp, w->final_ticket, stream));
        else HIP_TRY(launch_final(fp, stream));
    }
    if (w->timed == 1 && !nested) HIP_TRY(hipEventRecord(w->ev[3], stream));
    return NMN_OK;
}

static nmn_status check_search_args(const nmn_index* idx, const void* queries, uint32_t nq, uint32_t k,
                                    nmn_metric metric, const void* out_rows, const void* out_scores,
                                    const void* out_counts) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    if (k == 0) return fail_arg(NMN_ERR_INVALID_TOP_K, "k == 0");
    if (nq == 0 || nq > NMN_MAX_QUERIES) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "nq out of range");
    if (!queries || !out_rows || !out_scores || !out_counts)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null buffer");
    if ((int)metric < 0 || (int)metric > 3) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "bad metric");
    return NMN_OK;
}

extern "C" nmn_status nmn_index_search_device(nmn_index* idx, const float* queries_dev, uint32_t nq, uint32_t k,
                                              nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows_dev,
                                              float* out_scores_dev, uint32_t* out_counts_dev, void* stream) {
    if ((int)metric < 0 || (int)metric > 3) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "bad metric");
    return index_search_device(idx, queries_dev, nq, k, (int)metric, mask_dev, out_rows_dev, out_scores_dev,
                               out_counts_dev, static_cast<hipStream_t>(stream));
}

nmn_status nmn::index_search_device(nmn_index* idx, const float* queries_dev, uint32_t nq, uint32_t k, int metric_i,
                                    const uint64_t* mask_dev, uint64_t* out_rows_dev, float* out_scores_dev,
                                    uint32_t* out_counts_dev, hipStream_t stream, bool short_chain) {
    const nmn_metric metric = (nmn_metric)metric_i;
    nmn_status st = check_search_args(idx, queries_dev, nq, k, metric_i == kMetricNegL2 ? NMN_METRIC_EUCLIDEAN : metric,
                                      out_rows_dev, out_scores_dev, out_counts_dev);
    if (st != NMN_OK) return st;
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> g(idx->mu);
    Workspace* w = nullptr;
    st = ws_get(idx, s, nq, k, &w);
    if (st != NMN_OK) return st;
    static const bool no_short = env_set("NMN_NO_SHORT_CHAIN");
    // (MEASUREMENT ONLY: what the six rare-path launches cost the asynchronous API — with this set an overflowed query is NOT
    //  followed up by anybody and comes back with count 0xFFFFFFFF)
    static const bool force_short = env_set("NMN_MEASURE_DEVICE_SHORT_CHAIN");
    if (short_chain) idx->short_calls++;
    return search_enqueue(idx, w, queries_dev, nq, k, metric, mask_dev, out_rows_dev, out_scores_dev,
                          out_counts_dev, s, nullptr, nullptr,
                          ((short_chain && idx->short_calls > idx->short_off_until) || force_short) && !no_short && k <= NMN_MAX_TOP_K &&
                              idx->rows >= (1u << 18));
}

void nmn::index_short_chain_flagged(nmn_index* idx) {
    std::lock_guard<std::mutex> g(idx->mu);
    idx->short_off_until = idx->short_calls + 256;
}

static nmn_status stats_collect(nmn_index* idx, Workspace* w, nmn_search_stats* stats) {
    if (!stats) return NMN_OK;
    memset(stats, 0, sizeof *stats);
    stats->scan_ms = -1.f;
    stats->total_ms = -1.f;
    if (!w || !w->qstate) return NMN_OK;
    const uint32_t nqc = std::min(w->last_nq, w->nq_cap);
    std::vector<QState> qs(nqc);
    if (nqc) HIP_TRY(hipMemcpy(qs.data(), w->qstate, nqc * sizeof(QState), hipMemcpyDeviceToHost));
    for (auto& q : qs) {
        stats->candidates_rescored = std::max(stats->candidates_rescored, q.cand_count);
        stats->fallback_queries += q.overflow == 1 ? 1 : 0;  // 2 = crowd list: re-scored like candidates, just more of them
    }
    stats->rows_scanned = w->last_rows_scanned;  // upper bound when masked (excluded rows are skipped)
    stats->bytes_scanned = w->last_rows_scanned * (uint64_t)idx->dim * (uint64_t)w->last_elem_bytes;
    stats->sweep_kind = w->last_sweep_kind;
    stats->sweep_launches = w->last_sweep_launches;
    if (w->timed) {
        float a = 0.f, b = 0.f;
        hipEvent_t e1 = w->ev[1], e2 = w->ev[2];
        if (w->scan_ev_in_hist && w->hist_head) {  // (the pipeline keeps the sweep's events in the history ring)
            const uint32_t slot = (uint32_t)((w->hist_head - 1) % Workspace::kTimingHistory);
            e1 = w->hist[2 * slot];
            e2 = w->hist[2 * slot + 1];
        }
        if (w->last_rows_scanned && e1 && e2 && hipEventElapsedTime(&a, e1, e2) == hipSuccess) stats->scan_ms = a;
        if (w->timed == 1 && hipEventElapsedTime(&b, w->ev[0], w->ev[3]) == hipSuccess) stats->total_ms = b;
        (void)hipGetLastError();
    }
    return NMN_OK;
}

extern "C" nmn_status nmn_index_last_stats(nmn_index* idx, void* stream, nmn_search_stats* stats) {
    if (!idx || !stats) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipStreamSynchronize(s));
    std::lock_guard<std::mutex> g(idx->mu);
    auto it = idx->ws.find(s);
    return stats_collect(idx, it == idx->ws.end() ? nullptr : it->second, stats);
}

extern "C" nmn_status nmn_index_scan_history(nmn_index* idx, void* stream, float* scan_ms, uint32_t cap, uint32_t* n_out) {
    if (!idx || !n_out || (!scan_ms && cap)) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *n_out = 0;
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipStreamSynchronize(s));
    std::lock_guard<std::mutex> g(idx->mu);
    auto it = idx->ws.find(s);
    if (it == idx->ws.end()) return NMN_OK;
    Workspace* w = it->second;
    const uint64_t avail = std::min<uint64_t>(w->hist_head - w->hist_read, Workspace::kTimingHistory);
    const uint64_t take = std::min<uint64_t>(avail, cap);
    for (uint64_t i = w->hist_head - take; i < w->hist_head; i++) {  // the most recent `take`, oldest first
        float ms = -1.f;
        const uint32_t slot = (uint32_t)(i % Workspace::kTimingHistory);
        if (hipEventElapsedTime(&ms, w->hist[2 * slot], w->hist[2 * slot + 1]) != hipSuccess) {
            (void)hipGetLastError();
            ms = -1.f;
        }
        scan_ms[(*n_out)++] = ms;
    }
    w->hist_read = w->hist_head;
    return NMN_OK;
}

// ---- host-buffer searches: coalescing of concurrent callers into query batches ------------------------
// Queries one batch may carry: what ONE corpus sweep serves — 128 (rows of <= 768, 1024, 1280 elements) or 64 stationary queries
// on the matrix-core sweep, 4 on the VALU sweep (a longer batch there would only make every rider wait for the
// later sweeps of the others).
static uint32_t batch_queries(const nmn_index* idx, int metric) {
    if (!scan_mfma_supported(idx->ld, idx->dim, metric) || no_mfma() || idx->mirror_off) return 4;
    // 128 callers per sweep for every row length: either 128 stationary queries per workgroup (rows <= 768, 1024, 1280)
    // or several query blocks whose workgroups share the streamed tiles through their XCD's L2 (nmn_scan_mfma.hip);
    // 2M x 3072: 64 queries 2.9 ms, 128 queries 4.3 ms.
    return 2u * nmn_index::kCoalesceQueries;
}
// requests that may share a batch: the candidate pipeline (k <= NMN_MAX_TOP_K), at most one sweep's worth of queries
static bool mergeable(const nmn_index* idx, const HostReq& r) {
    return coalesce_enabled() && !r.own_batch && r.k <= NMN_MAX_TOP_K && r.nq <= batch_queries(idx, r.metric);
}
// Same metric, and filters that can share a sweep.  Filters are compared by address: two calls blocked in here with
// the same mask pointer necessarily mean the same bits (a caller changing them under a running search races with its
// own call).  DIFFERENT filters share a sweep too when both are bitmaps in device memory (or absent) and the shard's
// batches take the matrix-core sweep, which reads one bitmap per query.
static bool same_mask(const HostReq& a, const HostReq& b) {
    if (a.pred_cols || b.pred_cols) return false;  // a predicate yields its own bitmap
    return a.mask == b.mask && (a.mask == nullptr || a.mask_on_device == b.mask_on_device);
}
static bool same_batch_key(const nmn_index* idx, const HostReq& a, const HostReq& b) {
    if (a.metric != b.metric) return false;
    if (a.pred_cols && b.pred_cols && a.pred_cols != b.pred_cols) return false;  // one set of columns per batch
    if (same_mask(a, b)) return true;
    const bool dev_a = a.pred_cols || a.mask == nullptr || a.mask_on_device;
    const bool dev_b = b.pred_cols || b.mask == nullptr || b.mask_on_device;
    return dev_a && dev_b && batch_queries(idx, a.metric) > 4;
}

// One packed search for `reqs` (queries concatenated, k = the largest asked for) on host slot `slot`.
// `lk` holds idx->mu on entry; it is released once everything is enqueued (may still be held on an error return).
static nmn_status host_batch_body(nmn_index* idx, std::unique_lock<std::mutex>& lk, int slot, HostReq* const* reqs,
                                  size_t n_reqs) {
    if (!idx->host_slots[slot]) {
        if (slot == 0) idx->host_slots[0] = idx->host_stream;
        else HIP_TRY(hipStreamCreateWithFlags(&idx->host_slots[slot], hipStreamNonBlocking));
    }
    hipStream_t s = idx->host_slots[slot];
    uint32_t nq = 0, k = 0;
    bool want_stats = false;
    for (size_t i = 0; i < n_reqs; i++) {
        nq += reqs[i]->nq;
        k = std::max(k, reqs[i]->k);
        want_stats |= reqs[i]->stats != nullptr;
    }
    const HostReq& first = *reqs[0];
    Workspace* w = nullptr;
    nmn_status st = ws_get(idx, s, nq, k, &w);
    if (st != NMN_OK) return st;
    // ---- a small shard, one query: the whole search is ONE launch (tiny_search_kernel) — the query rides in the kernel
    // arguments, the kernel writes the result into pinned host memory; no H2D, no D2H, none of the pipeline's buffers
    constexpr size_t kTinyK = 1024, kOffScores = kTinyK * 8, kOffCount = kOffScores + kTinyK * 4;
    bool tiny = n_reqs == 1 && nq == 1 && !first.pred_cols && (first.mask == nullptr || first.mask_on_device) &&
                tiny_supported(idx->rows, idx->ld, idx->dim, k) && !no_tiny() && !idx->no_single_launch;
    if (tiny && !w->tiny_pool) {
        // its three buffers appear together or not at all (a workspace with the pool but no ticket would pass this test next
        // time and launch with null pointers); if one cannot be had the search simply takes the general pipeline
        unsigned long long* pool = nullptr;
        uint32_t* ticket = nullptr;
        uint8_t *out = nullptr, *out_dev = nullptr;
        bool ok = hipMalloc(reinterpret_cast<void**>(&pool), 128 * 512 * sizeof(unsigned long long)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&ticket), 4) == hipSuccess && hipMemsetAsync(ticket, 0, 4, s) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void**>(&out), kOffCount + 16, hipHostMallocMapped) == hipSuccess &&
                  hipHostGetDevicePointer(reinterpret_cast<void**>(&out_dev), out, 0) == hipSuccess;
        if (ok) {
            memset(out, 0, kOffCount + 16);
            w->tiny_pool = pool;
            w->tiny_ticket = ticket;
            w->tiny_out = out;
            w->tiny_out_dev = out_dev;
        } else {
            (void)hipGetLastError();
            if (out) (void)hipHostFree(out);
            if (ticket) (void)hipFree(ticket);
            if (pool) (void)hipFree(pool);
            tiny = false;
        }
    }
    if (tiny) {
        st = upload_fence_wait(idx, w, s);
        if (st != NMN_OK) return st;
        HIP_TRY(launch_tiny_search(idx->corpus, idx->norms, first.mask, idx->rows, idx->row_base, idx->ld, idx->dim, k, first.metric,
                                   first.queries, w->tiny_pool, w->tiny_ticket, reinterpret_cast<uint64_t*>(w->tiny_out_dev),
                                   reinterpret_cast<float*>(w->tiny_out_dev + kOffScores),
                                   reinterpret_cast<uint32_t*>(w->tiny_out_dev + kOffCount),
                                   reinterpret_cast<uint32_t*>(w->tiny_out_dev + kOffCount + 4), ++w->tiny_seq, s));
        const uint64_t rows_now = idx->rows;
        const uint32_t want_seq = w->tiny_seq;
        lk.unlock();
        // The kernel's last act is to echo the sequence number into the pinned block: spinning on that word returns a few
        // microseconds before hipStreamSynchronize would (1000 x 128: 28 us per call with the runtime's wait).  A kernel
        // that faults never writes it: after 2 ms the runtime's wait takes over and reports the error.
        {
            volatile const uint32_t* flag = reinterpret_cast<volatile const uint32_t*>(w->tiny_out + kOffCount + 4);
            const auto t_give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            uint32_t spins = 0;
            while (*flag != want_seq) {
                __builtin_ia32_pause();
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() > t_give_up) {
                    HIP_TRY(hipStreamSynchronize(s));
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        memcpy(first.out_rows, w->tiny_out, (size_t)k * 8);
        memcpy(first.out_scores, w->tiny_out + kOffScores, (size_t)k * 4);
        first.out_counts[0] = *reinterpret_cast<const uint32_t*>(w->tiny_out + kOffCount);
        if (first.stats) {
            memset(first.stats, 0, sizeof *first.stats);
            first.stats->rows_scanned = rows_now;
            first.stats->bytes_scanned = rows_now * (uint64_t)idx->dim * 4ull;  // exact scores straight from the f32 rows
            first.stats->scan_ms = -1.f;
            first.stats->total_ms = -1.f;
            first.stats->sweep_kind = rows_now ? NMN_SWEEP_EXACT : NMN_SWEEP_NONE;  // (tiny_search_kernel: exact scores, one launch)
            first.stats->sweep_launches = rows_now ? 1u : 0u;
        }
        return NMN_OK;
    }
    const size_t dim = idx->dim;
    const size_t qn = (size_t)nq * dim, on = (size_t)nq * k;
    const size_t words = (size_t)((idx->rows + 63) / 64);
    // packed result block: rows (8-byte aligned) | scores | counts
    const size_t off_scores = on * sizeof(uint64_t), off_counts = off_scores + on * sizeof(float);
    const size_t res_bytes = off_counts + (size_t)nq * sizeof(uint32_t);
    // ... | the predicates' row counters (8-byte aligned): they come back with the results in ONE copy (a second small D2H behind
    // the first cost a filtered search ~12 us of its 0.31 ms: profiles/r04y_*)
    size_t n_pred = 0;
    for (size_t i = 0; i < n_reqs; i++) n_pred += reqs[i]->pred_cols ? 1 : 0;
    const size_t pred_count_stride = (n_pred && words) ? 1 + (size_t)pred_batch_blocks(idx->rows, (uint32_t)n_pred) : 0;
    const size_t off_pred = (res_bytes + 7) & ~(size_t)7;
    const size_t pack_bytes = off_pred + n_pred * pred_count_stride * 8;
    HIP_TRY(grow(&w->h_queries, &w->h_queries_cap, qn));
    HIP_TRY(grow(&w->h_pack, &w->h_pack_cap, pack_bytes));
    {
        const uint8_t *in0 = w->pin_in, *out0 = w->pin_out;
        HIP_TRY(grow_pinned(&w->pin_in, &w->pin_in_cap, qn * sizeof(float)));
        HIP_TRY(grow_pinned(&w->pin_out, &w->pin_out_cap, pack_bytes + 16));  // (+ the polled sequence word, in the block's LAST 8 bytes)
        if (w->pin_in != in0 && hipHostGetDevicePointer(reinterpret_cast<void**>(&w->pin_in_dev), w->pin_in, 0) != hipSuccess) {
            (void)hipGetLastError();
            w->pin_in_dev = nullptr;
        }
        if (w->pin_out != out0) {
            memset(w->pin_out, 0, w->pin_out_cap);  // (the polled sequence word must not hold a stale match)
            if (hipHostGetDevicePointer(reinterpret_cast<void**>(&w->pin_out_dev), w->pin_out, 0) != hipSuccess) {
                (void)hipGetLastError();
                w->pin_out_dev = nullptr;
            }
        }
    }
    // Zero-copy I/O (round 6): the query block and the result block are pinned host memory the device can address, so the chain's
    // first kernel (qprep) READS the queries from it and its last kernel WRITES rows | scores | counts into it — no H2D and no D2H
    // copy packets around the chain (two blit kernels and their dependency gaps: ~10 us of a lone call's ~200,
    // profiles/r05j_search_launch_chains.txt).  Small calls only (a batch's 200 KB of queries are better off as one DMA burst than as
    // qprep's loads over PCIe); NMN_NO_ZERO_COPY=1: the staged copies, the A/B.
    static const bool no_zero_copy = env_set("NMN_NO_ZERO_COPY");
    const bool zero_copy = !no_zero_copy && w->pin_in_dev && w->pin_out_dev && qn * sizeof(float) <= (64u << 10) && res_bytes <= (256u << 10);
    // ... and a lone caller POLLS a word of that block instead of synchronising the stream (as the single-launch path of small shards
    // does): final_kernel's last workgroup stores the call's sequence number behind the results (system-scope release), the host
    // reads it a few microseconds before hipStreamSynchronize would return.  Bounded: 400 us, then the runtime's wait.
    // NMN_NO_POLL=1: the A/B.
    static const bool no_poll = env_set("NMN_NO_POLL") || env_set("NMN_FUSED_TAIL");  // (the fused tail's kernel does not publish)
    // (the word lives in the LAST 8 bytes of the pinned block, where no call's results ever reach: placed right behind the results it
    //  moved with k, and a row id an earlier, larger call had left there matched a sequence number — a k = 1 search returned before
    //  its kernels had run; tests/test_gpu_parity_basic.py caught it)
    const size_t off_done = (w->pin_out_cap - 8) & ~(size_t)7;
    const bool poll = zero_copy && !no_poll && n_reqs == 1 && nq <= 4 && k <= NMN_MAX_TOP_K && pack_bytes + 16 <= w->pin_out_cap;
    const float* const q_dev = zero_copy ? reinterpret_cast<const float*>(w->pin_in_dev) : w->h_queries;
    uint8_t* const pack_dev = zero_copy ? w->pin_out_dev : w->h_pack;
    unsigned long long* const d_pred_counts = reinterpret_cast<unsigned long long*>(w->h_pack + off_pred);
    const unsigned long long* const h_pred_counts = reinterpret_cast<const unsigned long long*>(w->pin_out + off_pred);
    {
        float* dst = reinterpret_cast<float*>(w->pin_in);
        for (size_t i = 0; i < n_reqs; i++) {
            memcpy(dst, reqs[i]->queries, (size_t)reqs[i]->nq * dim * sizeof(float));
            dst += (size_t)reqs[i]->nq * dim;
        }
    }
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(w->h_queries, w->pin_in, qn * sizeof(float), hipMemcpyHostToDevice, s));
    // ---- predicates of the batch: one launch on this stream, each into its own bitmap -------------------------
    std::vector<const uint64_t*> eff_mask(n_reqs);  // the bitmap each request's queries are searched with
    size_t pred_words = 0;
    bool pred_ticket_on = false;
    if (n_pred && words) {
        const nmn_columns* cols = nullptr;
        for (size_t i = 0; i < n_reqs; i++)
            if (reqs[i]->pred_cols) cols = reqs[i]->pred_cols;
        pred_words = (size_t)columns_words(cols);
        if (pred_words < words || columns_device(cols) != idx->device)
            return fail_arg(NMN_ERR_INVALID_ARGUMENT, "metadata columns do not cover the shard's rows (or live on another device)");
        HIP_TRY(grow(&w->pred_masks, &w->pred_masks_cap, n_pred * pred_words));
        // (NMN_PRED_TICKET=1: the last block of each program sums the partial counts itself — no count_reduce launch.  Built and
        //  measured SLOWER: a release fence + one atomic per block, 2 400 blocks, in a streaming kernel: filtered SIMILAR 0.269 ->
        //  0.383 ms (0.61 with the fence in every thread); the 5.8-us launch it saves stays.  Off.)
        static const bool pred_ticket_env = env_set("NMN_PRED_TICKET");
        pred_ticket_on = pred_ticket_env;
        if (pred_ticket_on && n_pred > w->pred_ticket_cap) {  // (zeroed once: the kernel leaves every counter at zero)
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(grow(&w->pred_ticket, &w->pred_ticket_cap, std::max<size_t>(n_pred, 64)));
            HIP_TRY(hipMemset(w->pred_ticket, 0, w->pred_ticket_cap * sizeof(uint32_t)));
        }
        size_t off = n_pred * pred_desc_bytes();
        std::vector<uint32_t> ops_off(n_pred), consts_off(n_pred);
        size_t j = 0;
        for (size_t i = 0; i < n_reqs; i++) {
            if (!reqs[i]->pred_cols) continue;
            ops_off[j] = (uint32_t)off;
            off += reqs[i]->pred_ops.size();
            consts_off[j] = (uint32_t)off;
            off += (size_t)reqs[i]->pred_n_consts * 8 + 8;
            j++;
        }
        const size_t block_bytes = off;
        HIP_TRY(grow(&w->pred_block, &w->pred_block_cap, block_bytes));
        HIP_TRY(grow_pinned(&w->pin_pred, &w->pin_pred_cap, block_bytes));
        j = 0;
        for (size_t i = 0; i < n_reqs; i++) {
            if (!reqs[i]->pred_cols) continue;
            pred_desc_write(w->pin_pred + j * pred_desc_bytes(), ops_off[j], (uint32_t)(reqs[i]->pred_ops.size() / pred_op_bytes()),
                            consts_off[j], w->pred_masks + j * pred_words, d_pred_counts + j * pred_count_stride,
                            // (zero copy: the selected-row total of each predicate lands in the pinned result block by itself)
                            zero_copy ? reinterpret_cast<unsigned long long*>(w->pin_out_dev + off_pred) + j * pred_count_stride : nullptr,
                            pred_ticket_on ? w->pred_ticket + j : nullptr);
            memcpy(w->pin_pred + ops_off[j], reqs[i]->pred_ops.data(), reqs[i]->pred_ops.size());
            if (reqs[i]->pred_n_consts) memcpy(w->pin_pred + consts_off[j], reqs[i]->pred_consts, (size_t)reqs[i]->pred_n_consts * 8);
            j++;
        }
        HIP_TRY(hipMemcpyAsync(w->pred_block, w->pin_pred, block_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(launch_pred_batch(cols, w->pred_block, (uint32_t)n_pred, idx->rows, s, pred_ticket_on));
    }
    {
        size_t j = 0;
        for (size_t i = 0; i < n_reqs; i++) {
            if (reqs[i]->pred_cols) eff_mask[i] = (n_pred && words) ? w->pred_masks + (j++) * pred_words : nullptr;
            else eff_mask[i] = reqs[i]->mask;
        }
    }
    bool one_mask = true;
    for (size_t i = 1; i < n_reqs; i++)
        one_mask = one_mask && !reqs[i]->pred_cols && !first.pred_cols && same_mask(first, *reqs[i]);
    const uint64_t* mask_dev = nullptr;
    std::vector<const uint64_t*> qmasks;  // one device bitmap (or null) per query when the batch mixes filters
    if (n_reqs == 1 && first.pred_cols) {
        mask_dev = eff_mask[0];  // a lone filtered search: the ordinary masked sweep (skips what the bitmap excludes)
    } else if (!one_mask) {
        if (words) {
            for (size_t i = 0; i < n_reqs; i++) qmasks.insert(qmasks.end(), reqs[i]->nq, eff_mask[i]);
            HIP_TRY(grow(&w->h_qmasks, &w->h_qmasks_cap, (size_t)nq));
            // pageable -> device is a staged copy: the vector may go out of scope once the call returns
            HIP_TRY(hipMemcpyAsync(w->h_qmasks, qmasks.data(), (size_t)nq * sizeof(uint64_t*), hipMemcpyHostToDevice, s));
        }
    } else if (first.mask && words && first.mask_on_device) {
        mask_dev = first.mask;
    } else if (first.mask && words) {
        HIP_TRY(grow(&w->h_mask, &w->h_mask_cap, words));
        HIP_TRY(hipMemcpyAsync(w->h_mask, first.mask, words * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        mask_dev = w->h_mask;
    }
    uint64_t* d_rows = reinterpret_cast<uint64_t*>(pack_dev);
    float* d_scores = reinterpret_cast<float*>(pack_dev + off_scores);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(pack_dev + off_counts);
    // A mixed batch reads every row once; its members one by one read what their bitmaps select.  With predicates in
    // the batch the selectivities are known only now: fetch their counts (the launches above are tens of microseconds)
    // and serve the members separately when that is the cheaper way (few, selective filters).
    bool separately = false;
    if (!qmasks.empty() && n_pred && n_reqs <= 16) {
        HIP_TRY(hipMemcpyAsync(w->pin_out + off_pred, d_pred_counts, n_pred * pred_count_stride * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const double sweep_us = (double)idx->rows * idx->ld * 2.0 / 5.5e6, fixed_us = 100.0;
        double sum = 0.0;
        size_t j = 0;
        for (size_t i = 0; i < n_reqs; i++) {
            double sel = selectivity_of(idx, *reqs[i]);
            if (reqs[i]->pred_cols)
                sel = (double)h_pred_counts[(j++) * pred_count_stride] / (double)std::max<uint64_t>(idx->rows, 1);
            sum += std::max(fixed_us, sel * sweep_us);
        }
        separately = sum < 1.1 * sweep_us + fixed_us;
    }
    // The chain of a search on a shard of >= 2^18 rows is 5 launches that do the work and 6 that only act when a candidate list
    // overflowed.  This call waits for its answer anyway, so it enqueues the five (search_enqueue: short_chain), looks at the
    // counts that come back, and only when final_kernel flagged a query runs the call again with the whole chain.
    static const bool no_short = env_set("NMN_NO_SHORT_CHAIN");  // (A/B switch)
    idx->short_calls++;
    const bool try_short = !no_short && k <= NMN_MAX_TOP_K && idx->rows >= (1u << 18) && idx->short_calls > idx->short_off_until;
    auto enqueue_all = [&](bool short_chain) -> nmn_status {
        nmn_status e = NMN_OK;
        if (separately) {
            size_t q0 = 0;
            for (size_t i = 0; i < n_reqs && e == NMN_OK; i++) {
                e = search_enqueue(idx, w, q_dev + q0 * dim, reqs[i]->nq, k, (nmn_metric)first.metric, eff_mask[i],
                                   d_rows + q0 * k, d_scores + q0 * k, d_counts + q0, s, nullptr, nullptr, short_chain);
                q0 += reqs[i]->nq;
            }
        } else if (!qmasks.empty())
            e = search_enqueue(idx, w, q_dev, nq, k, (nmn_metric)first.metric, nullptr, d_rows, d_scores, d_counts, s,
                               w->h_qmasks, qmasks.data(), short_chain);
        else
            e = search_enqueue(idx, w, q_dev, nq, k, (nmn_metric)first.metric, mask_dev, d_rows, d_scores, d_counts, s, nullptr,
                               nullptr, short_chain);
        if (e != NMN_OK) return e;
        if (!zero_copy) HIP_TRY(hipMemcpyAsync(w->pin_out, w->h_pack, pack_bytes, hipMemcpyDeviceToHost, s));
        // (zero copy: the results and the predicates' totals are already where the host reads them — count_reduce_batch_kernel wrote
        //  the totals into the pinned block when it formed them, ahead of the sweep)
        return NMN_OK;
    };
    volatile uint32_t* const done_host = reinterpret_cast<volatile uint32_t*>(w->pin_out + off_done);
    auto arm_poll = [&](bool on) {
        w->poll_word_dev = on ? reinterpret_cast<uint32_t*>(w->pin_out_dev + off_done) : nullptr;
        if (on) w->done_seq++;
    };
    const bool poll_this = poll && !separately && qmasks.empty() && nq <= w->nq_cap;
    arm_poll(poll_this);
    st = enqueue_all(try_short);
    arm_poll(false);
    if (st != NMN_OK) return st;
    std::vector<unsigned long long> pred_selected(n_pred, 0ull);
    // (the predicates' counts — word 0 of each counter block — came back with the results: the tail of the packed block)
    lk.unlock();  // everything is enqueued: other threads may enqueue on their slots while this one waits
    bool seen = false;
    if (poll_this) {
        const uint32_t want = w->done_seq;
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
        uint32_t spins = 0;
        while (!(seen = (*done_host == want))) {
            __builtin_ia32_pause();
            if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() > give_up) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(s));
    if (try_short) {
        const uint32_t* hc = reinterpret_cast<const uint32_t*>(w->pin_out + off_counts);
        bool flagged = false;
        for (uint32_t q = 0; q < nq; q++) flagged = flagged || hc[q] == 0xFFFFFFFFu;
        if (flagged) {  // (this slot is still ours: no writer can have changed the shard)
            lk.lock();
            idx->short_off_until = idx->short_calls + 256;  // (searches that keep overflowing: the whole chain at once for a while)
            st = enqueue_all(false);
            lk.unlock();
            if (st != NMN_OK) return st;
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    if (n_pred && words)
        for (size_t j = 0; j < n_pred; j++)
            pred_selected[j] = h_pred_counts[j * pred_count_stride];
    // hand the results out: the first k_i of each query's k-list are that request's answer (same total order)
    const uint64_t* h_rows = reinterpret_cast<const uint64_t*>(w->pin_out);
    const float* h_scores = reinterpret_cast<const float*>(w->pin_out + off_scores);
    const uint32_t* h_counts = reinterpret_cast<const uint32_t*>(w->pin_out + off_counts);
    size_t q0 = 0;
    for (size_t i = 0; i < n_reqs; i++) {
        HostReq& r = *reqs[i];
        for (uint32_t j = 0; j < r.nq; j++) {
            memcpy(r.out_rows + (size_t)j * r.k, h_rows + (q0 + j) * k, (size_t)r.k * sizeof(uint64_t));
            memcpy(r.out_scores + (size_t)j * r.k, h_scores + (q0 + j) * k, (size_t)r.k * sizeof(float));
            r.out_counts[j] = std::min(h_counts[q0 + j], r.k);
        }
        q0 += r.nq;
    }
    {
        size_t j = 0;
        for (size_t i = 0; i < n_reqs; i++)
            if (reqs[i]->pred_cols) {
                if (reqs[i]->selected_out) *reqs[i]->selected_out = (n_pred && words) ? pred_selected[j] : 0;
                j++;
            }
    }
    if (want_stats) {
        nmn_search_stats batch_stats;
        st = stats_collect(idx, w, &batch_stats);
        if (st != NMN_OK) return st;
        for (size_t i = 0; i < n_reqs; i++)
            if (reqs[i]->stats) *reqs[i]->stats = batch_stats;
    }
    return NMN_OK;
}

static nmn_status host_submit(nmn_index* idx, HostReq& me, HostReq* const* extras = nullptr, size_t n_extras = 0);

nmn_status nmn::index_search_hostio(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k, int metric,
                                    const uint64_t* mask, bool mask_on_device, uint64_t* out_rows, float* out_scores,
                                    uint32_t* out_counts, nmn_search_stats* stats, uint64_t mask_rows) {
    nmn_status st = check_search_args(idx, queries, nq, k, metric == kMetricNegL2 ? NMN_METRIC_EUCLIDEAN : (nmn_metric)metric,
                                      out_rows, out_scores, out_counts);
    if (st != NMN_OK) return st;
    HIP_TRY(hipSetDevice(idx->device));
    HostReq me;
    me.queries = queries;
    me.nq = nq;
    me.k = k;
    me.metric = metric;
    me.mask = mask;
    me.mask_on_device = mask_on_device;
    me.mask_rows = mask ? mask_rows : UINT64_MAX;
    me.out_rows = out_rows;
    me.out_scores = out_scores;
    me.out_counts = out_counts;
    me.stats = stats;
    return host_submit(idx, me);
}

extern "C" nmn_status nmn_index_search_pred(nmn_index* idx, nmn_columns* cols, const nmn_pred_op* prog, uint32_t n_ops,
                                            const uint64_t* consts, uint64_t n_consts, const float* queries, uint32_t nq,
                                            uint32_t k, nmn_metric metric, uint64_t* out_rows, float* out_scores,
                                            uint32_t* out_counts, uint64_t* selected_out, nmn_search_stats* stats) {
    nmn_status st = check_search_args(idx, queries, nq, k, metric, out_rows, out_scores, out_counts);
    if (st != NMN_OK) return st;
    if (!cols || !prog || n_ops == 0 || (n_consts && !consts)) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null predicate");
    if (k > NMN_MAX_TOP_K) return fail_arg(NMN_ERR_TOP_K_TOO_LARGE, "nmn_index_search_pred serves k <= NMN_MAX_TOP_K");
    // checked per REQUEST, before it can ride in anybody's batch: a bad set of columns fails its own caller only
    if (columns_device(cols) != idx->device || columns_words(cols) < (idx->rows + 63) / 64)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "metadata columns do not cover the shard's rows (or live on another device)");
    HIP_TRY(hipSetDevice(idx->device));
    HostReq me;
    me.queries = queries;
    me.nq = nq;
    me.k = k;
    me.metric = (int)metric;
    me.mask = nullptr;
    me.mask_on_device = true;
    me.mask_rows = UINT64_MAX;
    me.out_rows = out_rows;
    me.out_scores = out_scores;
    me.out_counts = out_counts;
    me.stats = stats;
    me.pred_cols = cols;
    me.pred_consts = consts;
    me.pred_n_consts = n_consts;
    me.selected_out = selected_out;
    if (selected_out) *selected_out = 0;
    st = columns_compile(cols, prog, n_ops, n_consts, idx->rows, &me.pred_ops);  // rows: read without the lock; re-checked by the leader
    if (st != NMN_OK) return st;
    return host_submit(idx, me);
}

uint32_t nmn::index_hostio_many_capacity(const nmn_index* idx, int metric, uint32_t k) {
    if (!idx || k > NMN_MAX_TOP_K || !coalesce_enabled()) return 0;
    const uint32_t cap = batch_queries(idx, metric);
    return cap > 4 ? cap : 0;  // (4: the VALU sweep, one bitmap for all its queries)
}

nmn_status nmn::index_search_hostio_many(nmn_index* idx, const HostSearchSpec* specs, uint32_t n, uint32_t k, int metric,
                                         nmn_search_stats* stats) {
    if (!idx || !specs || n == 0) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (n > index_hostio_many_capacity(idx, metric, k)) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "more searches than one batch carries");
    HIP_TRY(hipSetDevice(idx->device));
    std::vector<HostReq> reqs(n);
    std::vector<HostReq*> extras;
    for (uint32_t i = 0; i < n; i++) {
        HostReq& r = reqs[i];
        nmn_status st = check_search_args(idx, specs[i].query, 1, k, metric == kMetricNegL2 ? NMN_METRIC_EUCLIDEAN : (nmn_metric)metric,
                                          specs[i].out_rows, specs[i].out_scores, specs[i].out_count);
        if (st != NMN_OK) return st;
        r.queries = specs[i].query;
        r.nq = 1;
        r.k = k;
        r.metric = metric;
        r.mask = specs[i].mask;
        r.mask_on_device = true;
        r.mask_rows = specs[i].mask ? specs[i].mask_rows : UINT64_MAX;
        r.out_rows = specs[i].out_rows;
        r.out_scores = specs[i].out_scores;
        r.out_counts = specs[i].out_count;
        r.stats = i + 1 == n ? stats : nullptr;
        r.local = i != 0;
        r.own_batch = i == 0 && n > 1;
        if (i) extras.push_back(&r);
    }
    return host_submit(idx, reqs[0], extras.data(), extras.size());
}

// queue / lead / ride: the coalescer proper
static nmn_status host_submit(nmn_index* idx, HostReq& me, HostReq* const* extras, size_t n_extras) {
    const uint32_t nq = me.nq;
    const int metric = me.metric;
    nmn_status st = NMN_OK;
    std::unique_lock<std::mutex> lk(idx->mu);
    if (idx->host_queue.empty() && idx->writers_waiting == 0 && idx->slots_busy < lead_limit_for(idx, me)) {
        me.slot = slot_take(idx);  // the shard can take another search right now: lead a batch of one
    } else {
        idx->host_queue.push_back(&me);
        const bool tell = idx->gathering > 0;
        lk.unlock();
        if (tell) idx->gather_cv.notify_all();
        {
            std::unique_lock<std::mutex> ml(me.m);
            me.cv.wait(ml, [&] { return me.done || me.lead; });
            if (me.done) {  // rode in somebody's batch
                ml.unlock();
                if (me.st != NMN_OK) return fail_arg(me.st, me.err.c_str());  // the leader's failure text, on this thread
                return NMN_OK;
            }
        }
        lk.lock();  // told to lead: already out of the queue, me.slot is ours
    }
    // lead a batch: this request plus every queued one that can share its sweep
    std::vector<HostReq*> batch{&me};
    if (n_extras) {  // the caller's own batch: nobody else rides in it
        batch.insert(batch.end(), extras, extras + n_extras);
    } else if (mergeable(idx, me)) {
        const uint32_t limit = batch_queries(idx, metric);
        const uint32_t gather_us = gather_window_us(idx);
        if (gather_us && idx->last_batch_requests > 1) {
            // 3/4 of the previous batch is "everybody is back" (the rest may have left for good)
            const size_t want = (size_t)idx->last_batch_requests - idx->last_batch_requests / 4;
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(gather_us);
            idx->gathering++;
            while (1 + idx->host_queue.size() < want &&
                   idx->gather_cv.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
            idx->gathering--;
        }
        uint32_t total = nq;
        auto& qu = idx->host_queue;
        // Differently filtered searches share a sweep that reads EVERY row (one bitmap per query), while a sweep with
        // one bitmap skips what it excludes: mixing pays once the filters waiting here would cost more than a whole
        // sweep when served one by one.  Otherwise each filter is led separately.
        bool mix = false;
        {
            // separately: each distinct filter costs max(a fixed launch-bound search, its share of a sweep);
            // together: one whole sweep (+ the fixed part)
            const double sweep_us = (double)idx->rows * idx->ld * 2.0 / 5.5e6, fixed_us = 100.0;
            auto alone_us = [&](const HostReq& r) { return std::max(fixed_us, selectivity_of(idx, r) * sweep_us); };
            double sum = alone_us(me);
            std::vector<const uint64_t*> seen{me.mask};
            for (HostReq* r : qu) {
                if (!mergeable(idx, *r) || !same_batch_key(idx, me, *r) || same_mask(me, *r)) continue;
                if (!r->pred_cols) {  // (a predicate is a filter of its own, whatever its bitmap will be)
                    if (std::find(seen.begin(), seen.end(), r->mask) != seen.end()) continue;
                    seen.push_back(r->mask);
                }
                sum += alone_us(*r);
            }
            mix = sum >= 1.1 * sweep_us + fixed_us;
        }
        for (auto it = qu.begin(); it != qu.end();) {
            HostReq* r = *it;
            if (mergeable(idx, *r) && same_batch_key(idx, me, *r) && (mix || same_mask(me, *r)) && total + r->nq <= limit) {
                total += r->nq;
                batch.push_back(r);
                it = qu.erase(it);
            } else {
                ++it;
            }
        }
    }
    idx->last_batch_requests = (uint32_t)batch.size();
    if (batch.size() > 1) {
        idx->coalesced_batches++;
        idx->coalesced_requests += batch.size();
    }
    const int slot = me.slot;
    try {
        st = host_batch_body(idx, lk, slot, batch.data(), batch.size());
    } catch (const std::exception& ex) {  // (host allocation failure: the riders must still be released)
        st = fail_arg(NMN_ERR_OUT_OF_MEMORY, ex.what());
    }
    const std::string err = st == NMN_OK ? std::string() : std::string(nmn_last_error());
    // wake the riders first (their results are in place), then pass the slot on
    for (HostReq* r : batch) {
        if (r == &me || r->local) continue;
        std::lock_guard<std::mutex> g(r->m);
        r->st = st;
        r->err = err;
        r->done = true;
        r->cv.notify_one();  // under r->m, see designate_leaders
    }
    if (!lk.owns_lock()) lk.lock();
    idx->slot_busy[slot] = false;
    idx->slots_busy--;
    designate_leaders(idx);
    const bool idle = idx->slots_busy == 0;
    lk.unlock();
    if (idle) idx->cv.notify_all();  // writers waiting for the slots to drain
    return st;
}

extern "C" nmn_status nmn_index_callers_probe(nmn_index* idx, const float* queries, uint32_t threads, uint32_t k,
                                              nmn_metric metric, double seconds, double* calls_per_s,
                                              uint64_t* merged_batches, uint64_t* merged_calls, uint64_t* mismatches) {
    if (!idx || !queries || !calls_per_s) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (threads == 0 || threads > 1024 || k == 0 || k > NMN_MAX_TOP_K || !(seconds > 0.0))
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "threads in 1..1024, k in 1..4096, seconds > 0");
    const size_t dim = idx->dim;
    const uint32_t check_every = 16;  // every 16th thread compares each of its answers with the reference
    std::vector<uint64_t> ref_rows((size_t)threads * k);
    std::vector<float> ref_scores((size_t)threads * k);
    uint32_t cnt = 0;
    for (uint32_t t = 0; t < threads; t += check_every) {
        nmn_status st = nmn_index_search(idx, queries + t * dim, 1, k, metric, nullptr, ref_rows.data() + (size_t)t * k,
                                         ref_scores.data() + (size_t)t * k, &cnt, nullptr);
        if (st != NMN_OK) return st;
    }
    uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
    {
        std::lock_guard<std::mutex> g(idx->mu);
        b0 = idx->coalesced_batches;
        r0 = idx->coalesced_requests;
    }
    std::atomic<uint64_t> calls{0}, bad{0};
    std::atomic<int> failed{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint64_t> rows(k);
            std::vector<float> scores(k);
            uint32_t c = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                if (nmn_index_search(idx, queries + t * dim, 1, k, metric, nullptr, rows.data(), scores.data(), &c,
                                     nullptr) != NMN_OK) {
                    failed = 1;
                    break;
                }
                if (t % check_every == 0 && (memcmp(rows.data(), ref_rows.data() + (size_t)t * k, (size_t)k * 8) != 0 ||
                                             memcmp(scores.data(), ref_scores.data() + (size_t)t * k, (size_t)k * 4) != 0))
                    bad++;
                calls++;
            }
        });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        std::lock_guard<std::mutex> g(idx->mu);
        b1 = idx->coalesced_batches;
        r1 = idx->coalesced_requests;
    }
    if (failed) return fail_arg(NMN_ERR_STORAGE, "a search of the callers probe failed");
    *calls_per_s = (double)calls.load() / dt;
    if (merged_batches) *merged_batches = b1 - b0;
    if (merged_calls) *merged_calls = r1 - r0;
    if (mismatches) *mismatches = bad.load();
    return NMN_OK;
}

extern "C" nmn_status nmn_index_coalesce_stats(nmn_index* idx, uint64_t* batches, uint64_t* requests) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (batches) *batches = idx->coalesced_batches;
    if (requests) *requests = idx->coalesced_requests;
    return NMN_OK;
}

extern "C" nmn_status nmn_index_search(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                       nmn_metric metric, const uint64_t* mask, uint64_t* out_rows,
                                       float* out_scores, uint32_t* out_counts, nmn_search_stats* stats) {
    const nmn_status st = index_search_hostio(idx, queries, nq, k, (int)metric, mask, false, out_rows, out_scores, out_counts, stats);
    if (st == NMN_OK && stats && mask && stats->rows_scanned && idx->dim) {
        // a host bitmap can be counted here: rows_scanned / bytes_scanned are the KEPT rows (excluded rows are never read), not
        // the upper bound the device-mask entry points report
        const uint64_t n = stats->rows_scanned, eb = stats->bytes_scanned / (n * idx->dim);
        uint64_t kept = 0;
        for (uint64_t wd = 0; wd < n / 64; wd++) kept += (uint64_t)__builtin_popcountll(mask[wd]);
        if (n % 64) kept += (uint64_t)__builtin_popcountll(mask[n / 64] & ((1ull << (n % 64)) - 1));
        stats->rows_scanned = kept;
        stats->bytes_scanned = kept * idx->dim * eb;
    }
    return st;
}

extern "C" nmn_status nmn_index_search_dmask(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                             nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows,
                                             float* out_scores, uint32_t* out_counts, nmn_search_stats* stats) {
    return index_search_hostio(idx, queries, nq, k, (int)metric, mask_dev, true, out_rows, out_scores, out_counts, stats);
}

// Pure read sweep over the shard's rows (no arithmetic): the bandwidth a read-only kernel reaches on this device
// with the scan's own access pattern.  *gbps_out = rows * ld * 4 bytes / best-of-`reps` kernel time.
extern "C" nmn_status nmn_index_search_dmask_hint(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                                  nmn_metric metric, const uint64_t* mask_dev, uint64_t mask_rows,
                                                  uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                                                  nmn_search_stats* stats) {
    return index_search_hostio(idx, queries, nq, k, (int)metric, mask_dev, true, out_rows, out_scores, out_counts, stats,
                               mask_rows);
}

extern "C" nmn_status nmn_index_read_probe(nmn_index* idx, uint32_t reps, double* gbps_out) {
    if (!idx || !gbps_out) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *gbps_out = 0.0;
    if (idx->rows == 0) return NMN_OK;
    HIP_TRY(hipSetDevice(idx->device));
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = idx->host_stream;
    float* sink = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sink), 4));
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    float best = 0.f;
    // Where the headline sweep is the ring kernel (scan_ring_supported, >= 4096 tiles) the probe is THAT kernel's data movement with
    // nothing behind it — same workgroups, stages and LDS-DMA pieces (the register-load probe below reads slower
    // than the ring sweep itself and is no ceiling for it); elsewhere scan_kernel's access pattern with the arithmetic removed.
    const uint64_t n_tiles = idx->rows / kTileRows;
    const bool ring = n_tiles >= 4096u && scan_ring_supported(idx->ld, idx->dim, NMN_METRIC_DOT_PRODUCT) && !getenv("NMN_NO_RING");
    const uint32_t ring_tpw = (uint32_t)std::max<uint64_t>(1, (n_tiles + kMaxScanWaves - 1) / kMaxScanWaves);
    uint64_t probe_rows = ring ? n_tiles * kTileRows : idx->rows;
    auto probe = [&]() { return ring ? launch_ring_probe(idx->corpus, idx->rows, idx->ld, ring_tpw, s) : launch_read_probe(idx->corpus, idx->rows, idx->ld, sink, s); };
    if (e == hipSuccess) e = probe();  // warm-up
    for (uint32_t i = 0; i < std::max(reps, 1u) && e == hipSuccess; i++) {
        e = hipEventRecord(a, s);
        if (e == hipSuccess) e = probe();
        if (e == hipSuccess) e = hipEventRecord(b, s);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e == hipSuccess && ms > 0.f && (best == 0.f || ms < best)) best = ms;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(sink);
    if (e != hipSuccess) return fail_hip(e, "nmn_index_read_probe");
    if (best > 0.f) *gbps_out = (double)probe_rows * idx->ld * 4.0 / (best * 1e-3) / 1e9;
    return NMN_OK;
}

// ---- exact helpers ------------------------------------------------------------------------------
extern "C" nmn_status nmn_index_score_rows(nmn_index* idx, const float* queries, uint32_t nq, nmn_metric metric,
                                           const uint64_t* local_rows, uint32_t n_rows, float* out_scores) {
    if (!idx || !queries || !local_rows || !out_scores) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (nq == 0 || n_rows == 0) return NMN_OK;
    if (nq > NMN_MAX_QUERIES) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "nq out of range");
    for (uint32_t i = 0; i < n_rows; i++)
        if (local_rows[i] >= idx->rows) return fail_arg(NMN_ERR_NOT_FOUND, "row out of range");
    HIP_TRY(hipSetDevice(idx->device));
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = idx->host_stream;
    // private small buffers: nq may exceed the per-pass query count of the search workspace
    float *dq = nullptr, *dqpad = nullptr, *dout = nullptr;
    QInfo* dqi = nullptr;
    QState* dqs = nullptr;
    uint64_t* drows = nullptr;
    auto done = [&](nmn_status st) {
        for (void* p : {(void*)dq, (void*)dqpad, (void*)dout, (void*)dqi, (void*)dqs, (void*)drows})
            if (p) (void)hipFree(p);
        return st;
    };
    hipError_t e;
#define TRY2(x) if ((e = (x)) != hipSuccess) return done(fail_hip(e, #x))
    TRY2(hipMalloc(reinterpret_cast<void**>(&dq), (size_t)nq * idx->dim * 4));
    TRY2(hipMalloc(reinterpret_cast<void**>(&dqpad), (size_t)nq * idx->ld * 4));
    TRY2(hipMalloc(reinterpret_cast<void**>(&dout), (size_t)nq * n_rows * 4));
    TRY2(hipMalloc(reinterpret_cast<void**>(&dqi), (size_t)nq * sizeof(QInfo)));
    TRY2(hipMalloc(reinterpret_cast<void**>(&dqs), (size_t)nq * sizeof(QState)));
    TRY2(hipMalloc(reinterpret_cast<void**>(&drows), (size_t)n_rows * 8));
    TRY2(hipMemcpyAsync(dq, queries, (size_t)nq * idx->dim * 4, hipMemcpyHostToDevice, s));
    TRY2(hipMemcpyAsync(drows, local_rows, (size_t)n_rows * 8, hipMemcpyHostToDevice, s));
    TRY2(launch_qprep(dq, nq, idx->dim, idx->ld, (int)metric, idx->max_norm_bits, dqpad, dqi, dqs, 0, s));
    TRY2(launch_score_rows(idx->corpus, idx->norms, dqpad, dqi, drows, n_rows, nq, idx->ld, idx->dim, (int)metric,
                           dout, s));
    TRY2(hipMemcpyAsync(out_scores, dout, (size_t)nq * n_rows * 4, hipMemcpyDeviceToHost, s));
    TRY2(hipStreamSynchronize(s));
#undef TRY2
    return done(NMN_OK);
}

extern "C" nmn_status nmn_index_count_exact(nmn_index* idx, const float* query, nmn_metric metric,
                                            const uint64_t* mask, float score, uint64_t* n_greater,
                                            uint64_t* n_equal) {
    if (!idx || !query || !n_greater || !n_equal) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = idx->host_stream;
    Workspace* w = nullptr;
    nmn_status st = ws_get(idx, s, 1, 1, &w);
    if (st != NMN_OK) return st;
    st = ws_alloc(idx, w);
    if (st != NMN_OK) return st;
    HIP_TRY(grow(&w->h_queries, &w->h_queries_cap, (size_t)idx->dim));
    HIP_TRY(hipMemcpyAsync(w->h_queries, query, (size_t)idx->dim * 4, hipMemcpyHostToDevice, s));
    const size_t words = (size_t)((idx->rows + 63) / 64);
    const uint64_t* mask_dev = nullptr;
    if (mask && words) {
        HIP_TRY(grow(&w->h_mask, &w->h_mask_cap, words));
        HIP_TRY(hipMemcpyAsync(w->h_mask, mask, words * 8, hipMemcpyHostToDevice, s));
        mask_dev = w->h_mask;
    }
    HIP_TRY(launch_qprep(w->h_queries, 1, idx->dim, idx->ld, (int)metric, idx->max_norm_bits, w->qpad, w->qinfo,
                         w->qstate, 0, s));
    ExactScanParams ex{};
    ex.corpus = idx->corpus;
    ex.norms = idx->norms;
    ex.qpad = w->qpad;
    ex.qinfo = w->qinfo;
    ex.qstate = nullptr;
    ex.mask = mask_dev;
    ex.scores = w->scores;
    ex.n_rows = idx->rows;
    ex.nql = 1;
    ex.ld = idx->ld;
    ex.dim = idx->dim;
    ex.nq = 1;
    ex.metric = (int)metric;
    HIP_TRY(launch_exact_scan(ex, s));
    HIP_TRY(hipMemsetAsync(w->h_counts2, 0, 16, s));
    HIP_TRY(launch_count_cmp(w->scores, idx->rows, score, w->h_counts2, s));
    unsigned long long host2[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(host2, w->h_counts2, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *n_greater = host2[0];
    *n_equal = host2[1];
    return NMN_OK;
}

// ---- shard merge --------------------------------------------------------------------------------
extern "C" nmn_status nmn_merge_topk_host(const uint64_t* rows, const float* scores, const uint32_t* counts,
                                          uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* out_rows,
                                          float* out_scores, uint32_t* out_counts) {
    if (!rows || !scores || !counts || !out_rows || !out_scores || !out_counts)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (k == 0) return fail_arg(NMN_ERR_INVALID_TOP_K, "k == 0");
    struct Hit {
        uint32_t key;
        uint64_t row;
        float score;
    };
    std::vector<Hit> all;
    all.reserve((size_t)n_lists * k);
    for (uint32_t q = 0; q < nq; q++) {
        all.clear();
        for (uint32_t l = 0; l < n_lists; l++) {
            const uint32_t c = std::min(counts[(size_t)l * nq + q], k);
            const size_t base = ((size_t)l * nq + q) * k;
            for (uint32_t i = 0; i < c; i++) all.push_back({score_to_key(scores[base + i]), rows[base + i], scores[base + i]});
        }
        // merge_top_k: sort by score descending (distributed.rs:424-429); ties by ascending row
        std::sort(all.begin(), all.end(), [](const Hit& a, const Hit& b) {
            return a.key > b.key || (a.key == b.key && a.row < b.row);
        });
        const uint32_t cnt = (uint32_t)std::min<size_t>(all.size(), k);
        for (uint32_t i = 0; i < k; i++) {
            out_rows[(size_t)q * k + i] = i < cnt ? all[i].row : UINT64_MAX;
            out_scores[(size_t)q * k + i] = i < cnt ? all[i].score : -INFINITY;
        }
        out_counts[q] = cnt;
    }
    return NMN_OK;
}

extern "C" nmn_status nmn_merge_topk_device(const uint64_t* rows_dev, const float* scores_dev,
                                            const uint32_t* counts_dev, uint32_t n_lists, uint32_t nq, uint32_t k,
                                            uint64_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                            void* stream) {
    if (!rows_dev || !scores_dev || !counts_dev || !out_rows_dev || !out_scores_dev || !out_counts_dev)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (k == 0) return fail_arg(NMN_ERR_INVALID_TOP_K, "k == 0");
    if (nq == 0 || n_lists == 0) return NMN_OK;
    HIP_TRY(launch_merge(rows_dev, scores_dev, counts_dev, 0, n_lists, nq, k, out_rows_dev, out_scores_dev,
                         out_counts_dev, static_cast<hipStream_t>(stream)));
    return NMN_OK;
}

extern "C" nmn_status nmn_merge_topk_device_strided(const uint64_t* rows_dev, const float* scores_dev,
                                                    const uint32_t* counts_dev, uint64_t list_stride_bytes,
                                                    uint32_t n_lists, uint32_t nq, uint32_t k,
                                                    uint64_t* out_rows_dev, float* out_scores_dev,
                                                    uint32_t* out_counts_dev, void* stream) {
    if (!rows_dev || !scores_dev || !counts_dev || !out_rows_dev || !out_scores_dev || !out_counts_dev)
        return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (k == 0) return fail_arg(NMN_ERR_INVALID_TOP_K, "k == 0");
    if (nq == 0 || n_lists == 0) return NMN_OK;
    HIP_TRY(launch_merge(rows_dev, scores_dev, counts_dev, list_stride_bytes, n_lists, nq, k, out_rows_dev,
                         out_scores_dev, out_counts_dev, static_cast<hipStream_t>(stream)));
    return NMN_OK;
}

// ---- synthetic data -----------------------------------------------------------------------------
extern "C" float nmn_synth_value(uint64_t seed, uint64_t row, uint32_t col) {
    return nmn::synth_value_host(seed, row, col);
}

extern "C" nmn_status nmn_synth_fill_host(float* out, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim) {
    if (!out) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    for (uint64_t i = 0; i < n; i++)
        for (uint32_t c = 0; c < dim; c++) out[i * dim + c] = nmn::synth_value_host(seed, row0 + i, c);
    return NMN_OK;
}

extern "C" nmn_status nmn_index_fill_synthetic(nmn_index* idx, uint64_t seed, uint64_t row0, uint64_t n) {
    if (!idx) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null index");
    if (row0 > idx->rows) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "row0 leaves a gap (row0 > rows)");
    if (n > idx->cap || row0 > idx->cap - n) return fail_arg(NMN_ERR_CAPACITY, "row0 + n > capacity_rows");  // (no u64 wrap)
    if (n == 0) return NMN_OK;
    HIP_TRY(hipSetDevice(idx->device));
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = idx->host_stream;
    HIP_TRY(launch_synth_fill(idx->corpus, idx->ld, idx->dim, seed, idx->row_base + row0, row0, n, s));
    idx->half_rows = std::min(idx->half_rows, row0);  // whatever the mirror held from row0 on is stale
    nmn_status st = rows_written(idx, row0, n, s);
    if (st != NMN_OK) return st;
    HIP_TRY(hipStreamSynchronize(s));
    idx->rows = std::max(idx->rows, row0 + n);
    return NMN_OK;
}

extern "C" nmn_status nmn_index_set_row(nmn_index* idx, uint64_t row, const float* vec_host) {
    if (!idx || !vec_host) return fail_arg(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (row >= idx->rows) return fail_arg(NMN_ERR_NOT_FOUND, "row out of range");
    HIP_TRY(hipSetDevice(idx->device));
    std::unique_lock<std::mutex> lk(idx->mu);
    IdleGuard idle(idx, lk);
    hipStream_t s = idx->host_stream;
    HIP_TRY(hipMemcpyAsync(idx->corpus + row * (uint64_t)idx->ld, vec_host, (size_t)idx->dim * 4,
                           hipMemcpyHostToDevice, s));
    nmn_status st = rows_written(idx, row, 1, s);
    if (st != NMN_OK) return st;
    HIP_TRY(hipStreamSynchronize(s));
    return NMN_OK;
}
