// nmn_sharded.hip — ONE process, N devices: the row-range sharded index behind the opaque `nmn_sharded` handle
// (include/neumann_gpu.h, "multi-GPU in one process").
//
// The reference host is one process holding an Arc<VectorEngine> (query_router/src/lib.rs:710); its model for fanning a
// SIMILAR out and merging the answers is the distributed planner's scatter-gather: every shard answers the same query,
// ResultMerger::merge_top_k concatenates, sorts by score descending and truncates (query_router/src/distributed.rs:173-180,
// 413-433).  Here a shard is a row range resident on one GPU (shard g owns rows [g*ceil(N/G), ...)); one
// call = queries replicated to every device -> the single-shard pipeline on each device's own stream, writing a PACKED
// block [rows u64 | scores f32 | counts u32] -> ONE collective that brings every block to the merging device ->
// merge_kernel (nmn_select.hip) -> one D2H.  The collective:
//   * RCCL (all devices distinct): ncclAllGather of the packed blocks inside one ncclGroupStart/End, one communicator per
//     device from ncclCommInitAll — over xGMI between the GPUs of a node; payload nq*k*12 B per shard, latency-bound.
//     librccl is dlopen'ed on first use (no link-time dependency: the library loads on hosts without it, and a process
//     that already carries a copy — PyTorch bundles one — keeps a single instance).
//   * peer copies (several LOGICAL shards on one device, which a communicator cannot express, or RCCL unavailable):
//     hipMemcpyPeerAsync of each block into the merging device's gather buffer, ordered by events.
// Global top-k is a subset of the union of the shards' top-k lists, so the answer is exactly the unsharded one.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "nmn_index.h"

using namespace nmn;

#define S_TRY(expr)                                                   \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return set_error_hip(_e, #expr);        \
    } while (0)

// ---- the slice of the RCCL API this file uses, resolved at run time ---------------------------------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;  // 0 = ncclSuccess
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.so) break;
        }
        if (!r.so) {
            r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
        auto sym = [&](const char* n) { return dlsym(r.so, n); };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd;
        if (!r.ok) r.why = "librccl lacks ncclCommInitAll / ncclAllGather / ncclGroupStart";
    });
    return r;
}
nmn_status fail_nccl(ncclResult_t e, const char* what) {
    Rccl& r = rccl();
    std::string m = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error") + " (" + std::to_string(e) + ")";
    return set_error(NMN_ERR_STORAGE, m.c_str());
}

// Every nmn_sharded handle owns communicators of its own over (possibly) the same devices, and an engine keeps one handle
// per mirror: two searches on two mirrors would issue grouped all-gathers on different communicators from different host
// threads with nothing ordering them across the devices — RCCL documents that as a potential deadlock (the devices may start
// the two collectives in opposite orders).  One process-wide lock is held from ncclGroupStart until every rank's stream has
// drained the collective: collectives of different handles never interleave.  (Their sweeps, enqueued before, still overlap.)
std::mutex& collective_mutex() {
    static std::mutex m;
    return m;
}

struct PackLayout {
    size_t size, off_scores, off_counts;
};
PackLayout pack_layout(uint32_t nq, uint32_t k) {
    PackLayout p;
    p.off_scores = (size_t)nq * k * 8;
    p.off_counts = p.off_scores + (size_t)nq * k * 4;
    p.size = (p.off_counts + (size_t)nq * 4 + 15) & ~(size_t)15;
    return p;
}
}  // namespace

struct ShardLane {  // what one shard needs for one search in flight
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float* queries = nullptr;      size_t queries_cap = 0;  // device, nq x dim
    uint8_t* block = nullptr;      size_t block_cap = 0;    // device, this shard's packed result block
    uint8_t* gathered = nullptr;   size_t gathered_cap = 0; // device, n_shards packed blocks (RCCL: every shard; peer: shard 0)
    uint64_t* mask = nullptr;      size_t mask_cap = 0;     // device, this shard's slice of the selection bitmap
};

// A caller of the handle while it is busy.  Callers that arrive during a search wait in `waiting`; when the running one ends,
// the oldest waiter is woken to lead and takes every waiting search of the same metric (no bitmap) along as ONE query
// batch — the request coalescing of nmn_index_search (nmn_api.hip), one level up: a second QUERY in a sweep of all the
// shards is nearly free, a second SWEEP costs a whole one.  Writers (upload, fill) queue as `solo` and run alone, in turn.
struct ShardedReq {
    const float* queries = nullptr;
    uint32_t nq = 0, k = 0;
    nmn_metric metric = NMN_METRIC_COSINE;
    const uint64_t* mask = nullptr;
    uint64_t* out_rows = nullptr;
    float* out_scores = nullptr;
    uint32_t* out_counts = nullptr;
    nmn_search_stats* stats = nullptr;
    bool solo = false;
    // filled by the leader of the batch this request rode in
    nmn_status status = NMN_OK;
    std::string err;
    bool done = false, lead = false;
    std::condition_variable cv;
};
constexpr uint32_t kShardedBatchQueries = 128;  // one pass of the matrix-core sweep

// One host thread per shard.  A search enqueues a dozen launches per shard; issued from ONE thread the shards of a node start
// one after the other (8 shards: the last begins ~0.4 ms after the first, longer than the sweep of a 1.25M-row shard), and a
// host upload that spans shards would cross one PCIe link at a time.  The crew runs `job(g)` for every shard at once and
// returns when all have finished ENQUEUEING (the streams carry on); errors come back with their text (the last-error string
// is thread-local).  Used when the shards sit on several devices (crew_start).
struct ShardCrew {
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t seq = 0;
    uint32_t pending = 0;
    bool quit = false;
    const std::function<nmn_status(uint32_t)>* job = nullptr;
    std::vector<nmn_status> st;
    std::vector<std::string> err;
    std::vector<std::thread> th;
};

struct nmn_sharded {
    std::deque<ShardedReq*> waiting;
    bool busy = false;
    uint64_t merged_batches = 0, merged_calls = 0;  // batches that carried >= 2 calls / calls in them
    std::condition_variable arrive_cv;               // a caller queued up (the leader may be waiting for the cohort to return)
    uint32_t last_batch_calls = 1;                   // calls the previous batch carried
    double last_batch_us = 0.0;                      // ... and how long it ran
    std::vector<float> cat_q;                        // a merged batch: the callers' queries back to back
    std::vector<uint64_t> cat_rows;
    std::vector<float> cat_scores;
    std::vector<uint32_t> cat_counts;
    uint32_t dim = 0, n_shards = 0;
    uint64_t cap = 0, per = 0;  // total capacity, rows per shard (= ceil(cap / n_shards); the last one may hold fewer)
    uint64_t row_base = 0;      // global id of row 0 (cyclic layout: added when the shards' local ids are mapped back)
    bool cyclic = false;        // NMN_SHARDED_LAYOUT_CYCLIC: 64-row blocks dealt round-robin (block b -> shard b % G, local block b / G)
    uint32_t rccl_ranks = 0;    // communicator ranks the create-time self-test saw answer (0: peer-copy gather)
    uint32_t gather = NMN_GATHER_PEER;
    std::vector<nmn_index*> shard;
    std::vector<int> device;
    std::vector<ShardLane> lane;
    std::vector<ncclComm_t> comm;  // RCCL mode: one per shard
    // merging device (= shard 0's): merged output, pinned staging
    uint8_t* out_block = nullptr;  size_t out_cap = 0;       // device
    uint8_t* pin_in = nullptr;     size_t pin_in_cap = 0;    // pinned: queries (+ per-shard mask slices)
    uint8_t* pin_out = nullptr;    size_t pin_out_cap = 0;   // pinned: merged block
    std::vector<uint64_t> mask_tmp;
    uint64_t searches = 0;
    float last_gather_ms = -1.f;
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;  // around the collective + merge on the merging device (when timed)
    bool timing = false;
    std::unique_ptr<ShardCrew> crew;  // null: the shards share one device (crew_start)
    std::mutex mu;  // guards `busy` / `waiting`; the lanes belong to whoever holds `busy` (one batch or one writer at a time)
};

template <typename T>
static hipError_t grow_dev(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(need, 16) * sizeof(T));
    if (e == hipSuccess) *cap = need;
    return e;
}
static hipError_t grow_pin(uint8_t** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = std::max<size_t>(need + need / 2, 4096);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(p), want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

static void shard_bounds(const nmn_sharded* s, uint32_t g, uint64_t total, uint64_t* r0, uint64_t* r1) {
    *r0 = std::min<uint64_t>((uint64_t)g * s->per, total);
    *r1 = std::min<uint64_t>(*r0 + s->per, total);
}

static void crew_worker(nmn_sharded* s, uint32_t g) {
    ShardCrew& c = *s->crew;
    (void)hipSetDevice(s->device[g]);
    uint64_t seen = 0;
    for (;;) {
        const std::function<nmn_status(uint32_t)>* job;
        {
            std::unique_lock<std::mutex> lk(c.m);
            c.cv_go.wait(lk, [&] { return c.quit || c.seq != seen; });
            if (c.quit) return;
            seen = c.seq;
            job = c.job;
        }
        const nmn_status st = (*job)(g);
        std::string why;
        if (st != NMN_OK) why = nmn_last_error();
        std::lock_guard<std::mutex> lk(c.m);
        c.st[g] = st;
        c.err[g] = std::move(why);
        if (--c.pending == 0) c.cv_done.notify_one();
    }
}
static void crew_start(nmn_sharded* s) {
    // Worth it where the shards sit on different GPUs.  Logical shards of ONE device gain nothing from parallel enqueueing (the
    // device serializes them anyway) and pay the hand-over: 8 shards on one MI355X, one caller, 0.94 -> 1.03 ms per search.
    // NMN_SHARDED_CREW=1 / 0 forces it on / off (tests run the crew on the one-GPU box that way).
    bool several_devices = false;
    for (int d : s->device) several_devices = several_devices || d != s->device[0];
    const char* force = getenv("NMN_SHARDED_CREW");
    if (s->n_shards < 2 || !(force ? atoi(force) != 0 : several_devices)) return;
    s->crew.reset(new ShardCrew());
    s->crew->st.assign(s->n_shards, NMN_OK);
    s->crew->err.resize(s->n_shards);
    for (uint32_t g = 0; g < s->n_shards; g++) s->crew->th.emplace_back(crew_worker, s, g);
}
static void crew_stop(nmn_sharded* s) {
    if (!s->crew) return;
    {
        std::lock_guard<std::mutex> lk(s->crew->m);
        s->crew->quit = true;
    }
    s->crew->cv_go.notify_all();
    for (std::thread& t : s->crew->th)
        if (t.joinable()) t.join();
    s->crew.reset();
}
// job(g) for every shard (side by side when there is a crew); the first failure in shard order is reported.  The caller holds
// s->busy, so one call is in here at a time.
static nmn_status for_each_shard(nmn_sharded* s, const std::function<nmn_status(uint32_t)>& job) {
    if (!s->crew) {
        for (uint32_t g = 0; g < s->n_shards; g++) {
            const nmn_status st = job(g);
            if (st != NMN_OK) return st;
        }
        return NMN_OK;
    }
    ShardCrew& c = *s->crew;
    std::unique_lock<std::mutex> lk(c.m);
    c.job = &job;
    c.pending = s->n_shards;
    c.seq++;
    c.cv_go.notify_all();
    c.cv_done.wait(lk, [&] { return c.pending == 0; });
    c.job = nullptr;
    for (uint32_t g = 0; g < s->n_shards; g++)
        if (c.st[g] != NMN_OK) return set_error(c.st[g], c.err[g].c_str());
    return NMN_OK;
}

// ---- the collective ----------------------------------------------------------------------------------------------------
// lane[g].block (`bytes` each) of every shard -> lane[0].gathered[g * bytes ...] (RCCL: -> every lane's `gathered`).  Enqueued
// on the lanes' streams behind whatever produced the blocks; on return the merging stream (lane 0) is ordered behind all of
// them.  RCCL: the caller holds collective_mutex() around this call (one issue order of the collectives on every device).
static nmn_status sharded_gather(nmn_sharded* s, size_t bytes) {
    const uint32_t G = s->n_shards;
    ShardLane& root = s->lane[0];
    if (s->gather == NMN_GATHER_RCCL) {
        Rccl& r = rccl();
        ncclResult_t e = r.GroupStart();
        if (e != 0) return fail_nccl(e, "ncclGroupStart");
        for (uint32_t g = 0; g < G && e == 0; g++) {
            // (grouped calls may be issued from one thread for all the devices it drives; RCCL sets the device itself)
            e = r.AllGather(s->lane[g].block, s->lane[g].gathered, bytes, kNcclChar, s->comm[g], s->lane[g].stream);
        }
        ncclResult_t e2 = r.GroupEnd();
        if (e != 0) return fail_nccl(e, "ncclAllGather");
        if (e2 != 0) return fail_nccl(e2, "ncclGroupEnd");
        return NMN_OK;
    }
    for (uint32_t g = 0; g < G; g++) {
        ShardLane& l = s->lane[g];
        S_TRY(hipSetDevice(s->device[g]));
        // enqueued on the PRODUCING shard's stream (right behind its pipeline); the merging stream waits for the event
        if (s->device[g] == s->device[0])
            S_TRY(hipMemcpyAsync(root.gathered + (size_t)g * bytes, l.block, bytes, hipMemcpyDeviceToDevice, l.stream));
        else
            S_TRY(hipMemcpyPeerAsync(root.gathered + (size_t)g * bytes, s->device[0], l.block, s->device[g], bytes, l.stream));
        if (g != 0) S_TRY(hipEventRecord(l.done, l.stream));
    }
    S_TRY(hipSetDevice(s->device[0]));
    for (uint32_t g = 1; g < G; g++) S_TRY(hipStreamWaitEvent(root.stream, s->lane[g].done, 0));
    return NMN_OK;
}

// wait for every lane's stream (after a failure: nothing may still be reading the staging buffers the next call rewrites)
static void sharded_drain(nmn_sharded* s) {
    for (uint32_t g = 0; g < s->lane.size(); g++) {
        if (!s->lane[g].stream) continue;
        (void)hipSetDevice(s->device[g]);
        (void)hipStreamSynchronize(s->lane[g].stream);
    }
    (void)hipGetLastError();
    if (!s->device.empty()) (void)hipSetDevice(s->device[0]);
}

// create-time self-test: ranks through the collective.  Shard g's block = 4 words {magic, g, G, ~g}; after the gather the
// merging device — with RCCL every device — must hold them in rank order.
static nmn_status sharded_selftest(nmn_sharded* s) {
    const uint32_t G = s->n_shards;
    constexpr size_t kBytes = 16;
    constexpr uint32_t kMagic = 0x4E4D4E53u;  // "NMNS"
    std::unique_lock<std::mutex> coll(collective_mutex(), std::defer_lock);
    if (s->gather == NMN_GATHER_RCCL) coll.lock();
    nmn_status st = NMN_OK;
    for (uint32_t g = 0; g < G && st == NMN_OK; g++) {
        ShardLane& l = s->lane[g];
        const uint32_t words[4] = {kMagic, g, G, ~g};
        hipError_t e = hipSetDevice(s->device[g]);
        if (e == hipSuccess) e = grow_dev(&l.block, &l.block_cap, kBytes);
        if (e == hipSuccess && (s->gather == NMN_GATHER_RCCL || g == 0)) e = grow_dev(&l.gathered, &l.gathered_cap, kBytes * G);
        if (e == hipSuccess && l.gathered) e = hipMemsetAsync(l.gathered, 0, kBytes * G, l.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(l.block, words, kBytes, hipMemcpyHostToDevice, l.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(l.stream);  // (`words` is a stack buffer)
        if (e != hipSuccess) st = set_error_hip(e, "self-test: staging a shard's rank");
    }
    if (st == NMN_OK) st = sharded_gather(s, kBytes);
    std::vector<uint32_t> got((size_t)4 * G);
    const uint32_t readers = s->gather == NMN_GATHER_RCCL ? G : 1u;
    for (uint32_t r = 0; r < readers && st == NMN_OK; r++) {
        ShardLane& l = s->lane[r];
        hipError_t e = hipSetDevice(s->device[r]);
        if (e == hipSuccess) e = hipMemcpyAsync(got.data(), l.gathered, kBytes * G, hipMemcpyDeviceToHost, l.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(l.stream);
        if (e != hipSuccess) {
            st = set_error_hip(e, "self-test: reading the gathered ranks back");
            break;
        }
        for (uint32_t g = 0; g < G; g++) {
            const uint32_t* w = got.data() + (size_t)4 * g;
            if (w[0] != kMagic || w[1] != g || w[2] != G || w[3] != ~g) {
                const std::string m = std::string("Storage error: multi-GPU self-test failed: device ") + std::to_string(s->device[r]) +
                                      " does not hold the block of shard " + std::to_string(g) + " (device " + std::to_string(s->device[g]) +
                                      ") after the " + (s->gather == NMN_GATHER_RCCL ? "RCCL all-gather" : "peer-copy gather");
                st = set_error(NMN_ERR_STORAGE, m.c_str());
                break;
            }
        }
    }
    if (st != NMN_OK) {
        const std::string why = nmn_last_error();
        sharded_drain(s);
        return set_error(st == NMN_ERR_STORAGE ? st : NMN_ERR_STORAGE, why.c_str());
    }
    sharded_drain(s);
    if (s->gather == NMN_GATHER_RCCL) {
        // ranks that answered = G (checked above on every device); the communicators must agree
        Rccl& r = rccl();
        uint32_t ranks = G;
        for (uint32_t g = 0; g < G && r.CommCount; g++) {
            int cnt = 0, me = -1;
            if (r.CommCount(s->comm[g], &cnt) != 0 || (uint32_t)cnt != G) ranks = 0;
            if (r.CommUserRank && (r.CommUserRank(s->comm[g], &me) != 0 || (uint32_t)me != g)) ranks = 0;
        }
        if (ranks != G) return set_error(NMN_ERR_STORAGE, "Storage error: multi-GPU self-test failed: ncclCommCount / ncclCommUserRank disagree with the shard list");
        s->rccl_ranks = G;
    }
    return NMN_OK;
}

// cyclic layout: a shard's pipeline reports LOCAL rows (its row_base is 0); global = base + ((local / 64) * G + g) * 64 + local % 64
__global__ void __launch_bounds__(256) cyclic_remap_kernel(uint64_t* __restrict__ rows, uint32_t n, uint32_t g, uint32_t G, uint64_t base) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    if (r != UINT64_MAX) rows[i] = base + (((r >> 6) * G + g) << 6) + (r & 63ull);
}

// rows [row0, row0 + n) of the global numbering that fall to shard g under the cyclic layout: ONE run of local rows
// [*local0, *local0 + *cnt) (only the first block of the run can start inside a block, only the last can end inside one)
static void cyclic_part(const nmn_sharded* s, uint32_t g, uint64_t row0, uint64_t n, uint64_t* local0, uint64_t* cnt) {
    const uint64_t G = s->n_shards;
    *local0 = 0;
    *cnt = 0;
    if (n == 0) return;
    const uint64_t b0 = row0 >> 6, bl = (row0 + n - 1) >> 6;
    const uint64_t bf = b0 + ((g + G - b0 % G) % G);  // first block >= b0 that belongs to shard g
    if (bf > bl) return;
    *local0 = (bf / G) * 64 + (bf == b0 ? (row0 & 63) : 0);
    for (uint64_t b = bf; b <= bl; b += G) *cnt += std::min(row0 + n, (b + 1) * 64) - std::max(row0, b * 64);
}
// the same rows, copied out of the caller's contiguous buffer src[(row - row0) * dim ...] into one contiguous run
static void cyclic_pack(const nmn_sharded* s, uint32_t g, const float* src, uint64_t row0, uint64_t n, float* dst) {
    const uint64_t G = s->n_shards;
    const uint64_t b0 = row0 >> 6, bl = (row0 + n - 1) >> 6;
    for (uint64_t b = b0 + ((g + G - b0 % G) % G); b <= bl; b += G) {
        const uint64_t a = std::max(row0, b * 64), e = std::min(row0 + n, (b + 1) * 64);
        memcpy(dst, src + (a - row0) * (uint64_t)s->dim, (size_t)(e - a) * s->dim * sizeof(float));
        dst += (e - a) * (uint64_t)s->dim;
    }
}

extern "C" nmn_status nmn_sharded_destroy(nmn_sharded* s) {
    if (!s) return NMN_OK;
    crew_stop(s);
    for (uint32_t g = 0; g < s->lane.size(); g++) {
        (void)hipSetDevice(s->device[g]);
        ShardLane& l = s->lane[g];
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        for (void* p : {(void*)l.queries, (void*)l.block, (void*)l.gathered, (void*)l.mask})
            if (p) (void)hipFree(p);
        if (l.done) (void)hipEventDestroy(l.done);
    }
    for (ncclComm_t c : s->comm)
        if (c && rccl().ok) (void)rccl().CommDestroy(c);
    if (!s->device.empty()) (void)hipSetDevice(s->device[0]);
    if (s->out_block) (void)hipFree(s->out_block);
    if (s->pin_in) (void)hipHostFree(s->pin_in);
    if (s->pin_out) (void)hipHostFree(s->pin_out);
    if (s->ev_g0) (void)hipEventDestroy(s->ev_g0);
    if (s->ev_g1) (void)hipEventDestroy(s->ev_g1);
    for (uint32_t g = 0; g < s->shard.size(); g++) {
        // the shards own their streams; lanes used them (no stream of our own to destroy)
        if (s->shard[g]) (void)nmn_index_destroy(s->shard[g]);
    }
    delete s;
    return NMN_OK;
}

extern "C" nmn_status nmn_sharded_create(const nmn_sharded_desc* d, nmn_sharded** out) {
    if (!d || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (d->dim == 0) return set_error(NMN_ERR_EMPTY_VECTOR, "dim == 0");
    if (d->n_shards == 0 || d->n_shards > NMN_MAX_SHARDS) return set_error(NMN_ERR_INVALID_ARGUMENT, "n_shards out of range (1..NMN_MAX_SHARDS)");
    if (d->gather > NMN_GATHER_PEER) return set_error(NMN_ERR_INVALID_ARGUMENT, "unknown gather mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return set_error(NMN_ERR_NO_DEVICE, "no HIP device");
    }
    nmn_sharded* s = new (std::nothrow) nmn_sharded();
    if (!s) return set_error(NMN_ERR_OUT_OF_MEMORY, "sharded alloc");
    if (d->layout > NMN_SHARDED_LAYOUT_CYCLIC) {
        delete s;
        return set_error(NMN_ERR_INVALID_ARGUMENT, "unknown nmn_sharded_desc.layout");
    }
    s->dim = d->dim;
    s->n_shards = d->n_shards;
    s->cap = d->capacity_rows;
    s->row_base = d->row_base;
    s->cyclic = d->layout == NMN_SHARDED_LAYOUT_CYCLIC && d->n_shards > 1;
    s->per = (d->capacity_rows + d->n_shards - 1) / d->n_shards;  // shard g owns [g*ceil(N/G), ...)
    if (s->cyclic) {  // whole 64-row blocks per shard: block b of the global numbering is local block b / G of shard b % G
        const uint64_t blocks = (d->capacity_rows + 63) / 64;
        s->per = (blocks + d->n_shards - 1) / d->n_shards * 64;
    }
    bool distinct = true;
    for (uint32_t g = 0; g < d->n_shards; g++) {
        int dev = d->devices ? d->devices[g] : (int)(g % (uint32_t)ndev);  // no list: round-robin over the node's GPUs
        if (dev < 0 || dev >= ndev) {
            nmn_sharded_destroy(s);
            return set_error(NMN_ERR_NO_DEVICE, "device ordinal out of range");
        }
        for (int e : s->device) distinct = distinct && e != dev;
        s->device.push_back(dev);
    }
    // which collective: RCCL needs one rank per DEVICE; logical shards sharing a device take the peer-copy gather
    if (d->gather == NMN_GATHER_RCCL && !distinct) {
        nmn_sharded_destroy(s);
        return set_error(NMN_ERR_INVALID_ARGUMENT, "NMN_GATHER_RCCL needs distinct devices (one communicator rank per GPU)");
    }
    s->gather = NMN_GATHER_PEER;
    if (d->gather == NMN_GATHER_RCCL || (d->gather == NMN_GATHER_AUTO && distinct && d->n_shards > 1)) {
        if (rccl().ok) s->gather = NMN_GATHER_RCCL;
        else if (d->gather == NMN_GATHER_RCCL) {
            nmn_sharded_destroy(s);
            return set_error(NMN_ERR_CONFIGURATION, rccl().why.c_str());
        }
    }
    for (uint32_t g = 0; g < d->n_shards; g++) {
        nmn_index_desc id{};
        id.dim = d->dim;
        id.flags = d->flags;
        uint64_t r0, r1;
        shard_bounds(s, g, s->cap, &r0, &r1);
        id.capacity_rows = s->cyclic ? s->per : r1 - r0;
        id.row_base = s->cyclic ? 0 : d->row_base + r0;  // (cyclic: local ids are mapped to global ones after the shard's search)
        id.device = s->device[g];
        id.cand_cap = d->cand_cap;
        nmn_index* idx = nullptr;
        nmn_status st = nmn_index_create(&id, &idx);
        if (st != NMN_OK) {
            nmn_sharded_destroy(s);
            return st;
        }
        s->shard.push_back(idx);
    }
    s->lane.resize(d->n_shards);
    for (uint32_t g = 0; g < d->n_shards; g++) {
        hipError_t e = hipSetDevice(s->device[g]);
        // the search of shard g runs on a stream of ITS device; logical shards on one device get a stream each, so their
        // pipelines overlap like those of different GPUs would
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->lane[g].stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->lane[g].done, hipEventDisableTiming);
        if (e != hipSuccess) {
            nmn_sharded_destroy(s);
            return set_error_hip(e, "stream / event of a shard");
        }
    }
    if (s->gather == NMN_GATHER_PEER) {
        // peer access between the merging device and the others makes hipMemcpyPeerAsync a direct xGMI copy
        (void)hipSetDevice(s->device[0]);
        for (uint32_t g = 1; g < d->n_shards; g++) {
            if (s->device[g] == s->device[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s->device[0], s->device[g]) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(s->device[g], 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled / not supported: the copy still works (staged)
            }
        }
    } else {
        s->comm.assign(d->n_shards, nullptr);
        ncclResult_t r = rccl().CommInitAll(s->comm.data(), (int)d->n_shards, s->device.data());
        if (r != 0) {
            nmn_status st = fail_nccl(r, "ncclCommInitAll");
            s->comm.clear();
            nmn_sharded_destroy(s);
            return st;
        }
    }
    (void)hipSetDevice(s->device[0]);
    (void)hipEventCreate(&s->ev_g0);
    (void)hipEventCreate(&s->ev_g1);
    crew_start(s);
    if (s->n_shards >= 2) {
        // first contact with the other devices happens HERE, not in somebody's search: every shard sends its rank through the
        // very collective a search uses and the merging device (RCCL: every device) must see all of them
        const nmn_status st = sharded_selftest(s);
        if (st != NMN_OK) {
            const std::string why = nmn_last_error();
            nmn_sharded_destroy(s);
            return set_error(st, why.c_str());
        }
    }
    *out = s;
    return NMN_OK;
}

extern "C" uint32_t nmn_sharded_shards(const nmn_sharded* s) { return s ? s->n_shards : 0; }
extern "C" nmn_index* nmn_sharded_shard(nmn_sharded* s, uint32_t g) { return (s && g < s->n_shards) ? s->shard[g] : nullptr; }
extern "C" int32_t nmn_sharded_device(const nmn_sharded* s, uint32_t g) { return (s && g < s->n_shards) ? s->device[g] : -1; }
extern "C" uint32_t nmn_sharded_gather_mode(const nmn_sharded* s) { return s ? s->gather : NMN_GATHER_AUTO; }
extern "C" uint64_t nmn_sharded_rows(const nmn_sharded* s) {
    uint64_t n = 0;
    if (s)
        for (nmn_index* i : s->shard) n += nmn_index_rows(i);
    return n;
}
extern "C" uint64_t nmn_sharded_global_row(const nmn_sharded* s, uint32_t g, uint64_t local_row) {
    if (!s || g >= s->n_shards) return UINT64_MAX;
    if (s->cyclic) return s->row_base + (((local_row >> 6) * s->n_shards + g) << 6) + (local_row & 63ull);
    return nmn_index_row_base(s->shard[g]) + local_row;
}
extern "C" uint32_t nmn_sharded_rccl_ranks(const nmn_sharded* s) { return s ? s->rccl_ranks : 0; }
extern "C" uint32_t nmn_sharded_layout(const nmn_sharded* s) { return (s && s->cyclic) ? NMN_SHARDED_LAYOUT_CYCLIC : NMN_SHARDED_LAYOUT_RANGES; }

// rows [row0, row0+n) of the GLOBAL numbering: each shard gets the part that falls into its range, all parts at once (a host
// upload then crosses every GPU's PCIe link at the same time).  Like nmn_index_upload rows must arrive without gaps, i.e. in
// global order (a shard fills up before the next one starts) — checked here, before any shard is touched.
template <typename F>
static nmn_status for_each_part(nmn_sharded* s, uint64_t row0, uint64_t n, F&& f) {
    if (n > s->cap || row0 > s->cap - n) return set_error(NMN_ERR_CAPACITY, "row0 + n > capacity_rows");
    if (row0 > nmn_sharded_rows(s)) return set_error(NMN_ERR_INVALID_ARGUMENT, "row0 leaves a gap (row0 > rows)");
    if (n == 0) return NMN_OK;
    const std::function<nmn_status(uint32_t)> job = [&](uint32_t g) -> nmn_status {
        uint64_t r0, r1;
        shard_bounds(s, g, s->cap, &r0, &r1);
        const uint64_t a = std::max(row0, r0), b = std::min(row0 + n, r1);
        if (a >= b) return NMN_OK;
        return f(g, a - r0, a - row0, b - a);
    };
    return for_each_shard(s, job);
}

// Take the handle for a writer or a lone search: returns with s->busy held by the caller (mu NOT held).
static void sharded_acquire_solo(nmn_sharded* s) {
    ShardedReq me;
    me.solo = true;
    std::unique_lock<std::mutex> lk(s->mu);
    if (!s->busy) {
        s->busy = true;
        return;
    }
    s->waiting.push_back(&me);
    me.cv.wait(lk, [&] { return me.lead; });
}
// Hand the handle to the oldest waiter (it leads the next batch), or mark it idle.
static void sharded_release(nmn_sharded* s) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->waiting.empty()) {
        ShardedReq* nx = s->waiting.front();
        s->waiting.pop_front();
        nx->lead = true;
        nx->cv.notify_one();  // busy stays set: ownership passes
    } else {
        s->busy = false;
    }
}
struct SoloGuard {
    nmn_sharded* s;
    explicit SoloGuard(nmn_sharded* s_) : s(s_) { sharded_acquire_solo(s); }
    ~SoloGuard() { sharded_release(s); }
};

// (the switches change what a running search reads: they take a turn like a writer)
extern "C" nmn_status nmn_sharded_set_timing(nmn_sharded* s, int32_t enabled) {
    if (!s) return set_error(NMN_ERR_INVALID_ARGUMENT, "null handle");
    SoloGuard turn(s);
    s->timing = enabled != 0;
    for (nmn_index* i : s->shard) (void)nmn_index_set_timing(i, enabled);
    return NMN_OK;
}
extern "C" nmn_status nmn_sharded_set_mirror(nmn_sharded* s, int32_t enabled) {
    if (!s) return set_error(NMN_ERR_INVALID_ARGUMENT, "null handle");
    if (enabled < 0 || enabled > 2) return set_error(NMN_ERR_INVALID_ARGUMENT, "nmn_sharded_set_mirror: enabled must be 0, 1 or 2");
    SoloGuard turn(s);
    for (nmn_index* i : s->shard) (void)nmn_index_set_mirror(i, enabled);
    return NMN_OK;
}

// the caller holds the handle's turn
static nmn_status sharded_upload_turn(nmn_sharded* s, const float* rows_host, uint64_t row0, uint64_t n) {
    if (!s->cyclic)
        return for_each_part(s, row0, n, [&](uint32_t sh, uint64_t local0, uint64_t src0, uint64_t cnt) {
            return nmn_index_upload(s->shard[sh], rows_host + src0 * (uint64_t)s->dim, local0, cnt);
        });
    // cyclic: every shard gets ONE run of local rows, gathered block by block out of the caller's buffer (by the shard's
    // crew thread when there is one: the packing and the PCIe copies of the shards run side by side)
    if (n > s->cap || row0 > s->cap - n) return set_error(NMN_ERR_CAPACITY, "row0 + n > capacity_rows");
    if (row0 > nmn_sharded_rows(s)) return set_error(NMN_ERR_INVALID_ARGUMENT, "row0 leaves a gap (row0 > rows)");
    if (n == 0) return NMN_OK;
    const std::function<nmn_status(uint32_t)> job = [&](uint32_t g) -> nmn_status {
        uint64_t local0, cnt;
        cyclic_part(s, g, row0, n, &local0, &cnt);
        if (cnt == 0) return NMN_OK;
        std::vector<float> run((size_t)cnt * s->dim);
        cyclic_pack(s, g, rows_host, row0, n, run.data());
        return nmn_index_upload(s->shard[g], run.data(), local0, cnt);
    };
    return for_each_shard(s, job);
}

extern "C" nmn_status nmn_sharded_upload(nmn_sharded* s, const float* rows_host, uint64_t row0, uint64_t n) {
    if (!s || (!rows_host && n)) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    SoloGuard turn(s);
    return sharded_upload_turn(s, rows_host, row0, n);
}

extern "C" nmn_status nmn_sharded_fill_synthetic(nmn_sharded* s, uint64_t seed, uint64_t row0, uint64_t n) {
    if (!s) return set_error(NMN_ERR_INVALID_ARGUMENT, "null handle");
    SoloGuard turn(s);
    if (s->cyclic) {
        // the device generator numbers a shard's rows row_base + local; under the cyclic layout the rows come from the host
        // generator instead (bit-identical values), a chunk at a time
        const uint64_t chunk = std::max<uint64_t>(64, ((64ull << 20) / ((uint64_t)s->dim * 4)) & ~63ull);
        std::vector<float> buf;
        for (uint64_t a = 0; a < n; a += chunk) {
            const uint64_t c = std::min(chunk, n - a);
            buf.resize((size_t)c * s->dim);
            nmn_status st = nmn_synth_fill_host(buf.data(), seed, s->row_base + row0 + a, c, s->dim);
            if (st == NMN_OK) st = sharded_upload_turn(s, buf.data(), row0 + a, c);
            if (st != NMN_OK) return st;
        }
        return NMN_OK;
    }
    return for_each_part(s, row0, n, [&](uint32_t sh, uint64_t local0, uint64_t, uint64_t cnt) {
        return nmn_index_fill_synthetic(s->shard[sh], seed, local0, cnt);  // value(seed, row_base + row, col): global ids
    });
}

// bits [b0, b0+nbits) of `src` (LSB first) -> dst[0 .. ceil(nbits/64)), bit 0 of dst = bit b0 of src
static void bitmap_slice(const uint64_t* src, uint64_t b0, uint64_t nbits, uint64_t* dst) {
    const uint64_t words = (nbits + 63) / 64, w0 = b0 >> 6;
    const unsigned sh = (unsigned)(b0 & 63);
    const uint64_t src_words = (b0 + nbits + 63) / 64;
    for (uint64_t i = 0; i < words; i++) {
        uint64_t lo = src[w0 + i] >> sh;
        if (sh && w0 + i + 1 < src_words) lo |= src[w0 + i + 1] << (64 - sh);
        dst[i] = lo;
    }
    if (nbits & 63) dst[words - 1] &= (~0ull) >> (64 - (nbits & 63));
}

// one query batch over every shard: the caller holds s->busy
static nmn_status sharded_run_body(nmn_sharded* s, const float* queries, uint32_t nq, uint32_t k, nmn_metric metric,
                              const uint64_t* mask, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                              nmn_search_stats* stats) {
    const uint32_t G = s->n_shards;
    const PackLayout pl = pack_layout(nq, k);
    const size_t qbytes = (size_t)nq * s->dim * sizeof(float);
    // ---- stage the replicated inputs once in pinned memory ------------------------------------------------------------
    std::vector<uint64_t> rows_of(G), base_of(G);
    for (uint32_t g = 0; g < G; g++) {
        rows_of[g] = nmn_index_rows(s->shard[g]);
        base_of[g] = (uint64_t)g * s->per;
    }
    size_t mask_words_total = 0;
    std::vector<size_t> mask_off(G, 0);
    if (mask)
        for (uint32_t g = 0; g < G; g++) {
            mask_off[g] = mask_words_total;
            mask_words_total += (size_t)((rows_of[g] + 63) / 64);
        }
    S_TRY(hipSetDevice(s->device[0]));
    S_TRY(grow_pin(&s->pin_in, &s->pin_in_cap, qbytes + mask_words_total * 8 + 16));
    S_TRY(grow_pin(&s->pin_out, &s->pin_out_cap, pl.size));
    memcpy(s->pin_in, queries, qbytes);
    uint64_t* pin_mask = reinterpret_cast<uint64_t*>(s->pin_in + ((qbytes + 15) & ~(size_t)15));
    if (mask)
        for (uint32_t g = 0; g < G; g++) {
            if (!rows_of[g]) continue;
            if (s->cyclic) {  // local word i of shard g is global word i * G + g (a block is one bitmap word)
                const size_t words = (size_t)((rows_of[g] + 63) / 64);
                for (size_t i = 0; i < words; i++) pin_mask[mask_off[g] + i] = mask[i * G + g];
            } else {
                bitmap_slice(mask, base_of[g], rows_of[g], pin_mask + mask_off[g]);
            }
        }
    // ---- every shard: H2D of the queries (and its bitmap slice), the single-shard pipeline, into its packed block --------
    const std::function<nmn_status(uint32_t)> enqueue = [&](uint32_t g) -> nmn_status {
        ShardLane& l = s->lane[g];
        S_TRY(hipSetDevice(s->device[g]));
        S_TRY(grow_dev(&l.queries, &l.queries_cap, (size_t)nq * s->dim));
        S_TRY(grow_dev(&l.block, &l.block_cap, pl.size));
        const bool holds_all = s->gather == NMN_GATHER_RCCL || g == 0;
        if (holds_all) S_TRY(grow_dev(&l.gathered, &l.gathered_cap, pl.size * G));
        S_TRY(hipMemcpyAsync(l.queries, s->pin_in, qbytes, hipMemcpyHostToDevice, l.stream));
        const uint64_t* mask_dev = nullptr;
        if (mask && rows_of[g]) {
            const size_t words = (size_t)((rows_of[g] + 63) / 64);
            S_TRY(grow_dev(&l.mask, &l.mask_cap, words));
            S_TRY(hipMemcpyAsync(l.mask, pin_mask + mask_off[g], words * 8, hipMemcpyHostToDevice, l.stream));
            mask_dev = l.mask;
        }
        nmn_status st = index_search_device(s->shard[g], l.queries, nq, k, (int)metric, mask_dev,
                                            reinterpret_cast<uint64_t*>(l.block), reinterpret_cast<float*>(l.block + pl.off_scores),
                                            reinterpret_cast<uint32_t*>(l.block + pl.off_counts), l.stream);
        if (st == NMN_OK && s->cyclic) {
            const uint32_t cnt = nq * k;
            hipLaunchKernelGGL(cyclic_remap_kernel, dim3((cnt + 255) / 256), dim3(256), 0, l.stream, reinterpret_cast<uint64_t*>(l.block), cnt,
                               g, G, s->row_base);
            S_TRY(hipGetLastError());
        }
        return st;
    };
    {
        const nmn_status st = for_each_shard(s, enqueue);
        if (st != NMN_OK) return st;
    }
    // ---- the collective: every packed block to the merging device (RCCL: to every device) ---------------------------------
    ShardLane& root = s->lane[0];
    if (s->timing) {
        S_TRY(hipSetDevice(s->device[0]));
        S_TRY(hipEventRecord(s->ev_g0, root.stream));
    }
    // (RCCL: collectives of different communicators must be ISSUED in the same order on every device — two handles searched from
    //  two threads — so the group call is made under a process-wide lock.  Only the issue: once every rank's all-gather sits in
    //  its stream the order is fixed, and the merge, the copy back and the drain of one handle do not hold up the searches of
    //  the others (it used to be held until every rank had drained).)
    //  with only the ISSUE under the lock, two communicators are in flight on the same devices while other host threads
    //  make implicitly synchronising HIP calls (a blocking hipMemcpy of a shard's counters, hipMalloc / hipFree of a workspace) —
    //  the RCCL guidance on concurrent communicators names that as a hang hazard, and no multi-GPU hardware has been available to
    //  test it on.  So by default the lock is kept until this handle's all-gathers have COMPLETED on every rank (its lanes drained:
    //  the gather is the last thing in each); NMN_RCCL_NARROW_LOCK=1 restores the issue-only lock for whoever can measure it.)
    {
        std::unique_lock<std::mutex> coll(collective_mutex(), std::defer_lock);
        if (s->gather == NMN_GATHER_RCCL) coll.lock();
        const nmn_status st = sharded_gather(s, pl.size);
        if (st != NMN_OK) return st;
        static const bool narrow = getenv("NMN_RCCL_NARROW_LOCK") != nullptr;
        if (s->gather == NMN_GATHER_RCCL && !narrow) {
            // (an event behind each lane's all-gather, and the wait on THOSE: what the lock protects is the collective, not whatever
            //  else a lane's stream may come to carry behind it — ADVICE r05)
            for (uint32_t g = 0; g < G; g++) {
                S_TRY(hipSetDevice(s->device[g]));
                S_TRY(hipEventRecord(s->lane[g].done, s->lane[g].stream));
            }
            for (uint32_t g = 0; g < G; g++) S_TRY(hipEventSynchronize(s->lane[g].done));
        }
    }
    // ---- merge_top_k on the merging device, one D2H ---------------------------------------------------------------------
    S_TRY(hipSetDevice(s->device[0]));
    S_TRY(grow_dev(&s->out_block, &s->out_cap, pl.size));
    S_TRY(launch_merge(reinterpret_cast<const uint64_t*>(root.gathered), reinterpret_cast<const float*>(root.gathered + pl.off_scores),
                       reinterpret_cast<const uint32_t*>(root.gathered + pl.off_counts), pl.size, G, nq, k,
                       reinterpret_cast<uint64_t*>(s->out_block), reinterpret_cast<float*>(s->out_block + pl.off_scores),
                       reinterpret_cast<uint32_t*>(s->out_block + pl.off_counts), root.stream));
    if (s->timing) S_TRY(hipEventRecord(s->ev_g1, root.stream));
    S_TRY(hipMemcpyAsync(s->pin_out, s->out_block, pl.size, hipMemcpyDeviceToHost, root.stream));
    S_TRY(hipStreamSynchronize(root.stream));
    if (s->gather == NMN_GATHER_RCCL)
        for (uint32_t g = 1; g < G; g++) {  // the other ranks of the collective must be done before their buffers are reused
            S_TRY(hipSetDevice(s->device[g]));
            S_TRY(hipStreamSynchronize(s->lane[g].stream));
        }
    memcpy(out_rows, s->pin_out, (size_t)nq * k * 8);
    memcpy(out_scores, s->pin_out + pl.off_scores, (size_t)nq * k * 4);
    memcpy(out_counts, s->pin_out + pl.off_counts, (size_t)nq * 4);
    s->searches++;
    if (s->timing) {
        float ms = -1.f;
        (void)hipSetDevice(s->device[0]);
        if (hipEventElapsedTime(&ms, s->ev_g0, s->ev_g1) == hipSuccess) {
            std::lock_guard<std::mutex> lk(s->mu);
            s->last_gather_ms = ms;
        }
        (void)hipGetLastError();
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->scan_ms = -1.f;
        stats->total_ms = -1.f;
        for (uint32_t g = 0; g < G; g++) {
            nmn_search_stats one;
            if (nmn_index_last_stats(s->shard[g], s->lane[g].stream, &one) != NMN_OK) continue;
            stats->rows_scanned += one.rows_scanned;
            stats->bytes_scanned += one.bytes_scanned;
            stats->candidates_rescored = std::max(stats->candidates_rescored, one.candidates_rescored);
            stats->fallback_queries += one.fallback_queries;
            stats->scan_ms = std::max(stats->scan_ms, one.scan_ms);  // the shards run side by side: the slowest counts
            stats->total_ms = std::max(stats->total_ms, one.total_ms);
            if (g == 0 || one.sweep_kind > stats->sweep_kind) stats->sweep_kind = one.sweep_kind;  // (shards of one handle take the same sweep unless a mirror did not fit on one)
            stats->sweep_launches = std::max(stats->sweep_launches, one.sweep_launches);
        }
    }
    return NMN_OK;
}

static nmn_status sharded_run(nmn_sharded* s, const float* queries, uint32_t nq, uint32_t k, nmn_metric metric,
                              const uint64_t* mask, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                              nmn_search_stats* stats) {
    const nmn_status st = sharded_run_body(s, queries, nq, k, metric, mask, out_rows, out_scores, out_counts, stats);
    if (st != NMN_OK) {
        // some lanes may still be running what was enqueued before the failure: nothing of it may outlive this call (the
        // pinned staging and the lanes' device buffers are rewritten by the next one)
        const std::string why = nmn_last_error();
        sharded_drain(s);
        return set_error(st, why.c_str());
    }
    return st;
}

extern "C" nmn_status nmn_sharded_search(nmn_sharded* s, const float* queries, uint32_t nq, uint32_t k, nmn_metric metric,
                                         const uint64_t* mask, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                                         nmn_search_stats* stats) {
    if (!s) return set_error(NMN_ERR_INVALID_ARGUMENT, "null handle");
    if (k == 0) return set_error(NMN_ERR_INVALID_TOP_K, "k == 0");
    if (nq == 0 || nq > NMN_MAX_QUERIES) return set_error(NMN_ERR_INVALID_ARGUMENT, "nq out of range");
    if (!queries || !out_rows || !out_scores || !out_counts) return set_error(NMN_ERR_INVALID_ARGUMENT, "null buffer");
    if ((int)metric < 0 || (int)metric > 3) return set_error(NMN_ERR_INVALID_ARGUMENT, "bad metric");
    ShardedReq me;
    me.queries = queries;
    me.nq = nq;
    me.k = k;
    me.metric = metric;
    me.mask = mask;
    me.out_rows = out_rows;
    me.out_scores = out_scores;
    me.out_counts = out_counts;
    me.stats = stats;
    std::vector<ShardedReq*> batch{&me};
    {
        std::unique_lock<std::mutex> lk(s->mu);
        if (s->busy) {
            s->waiting.push_back(&me);
            s->arrive_cv.notify_one();
            me.cv.wait(lk, [&] { return me.done || me.lead; });
            if (me.done) {  // rode in somebody's batch
                if (me.status != NMN_OK) return set_error(me.status, me.err.c_str());
                return NMN_OK;
            }
        } else {
            s->busy = true;
        }
        // I lead.  The callers of the batch that just ended are on their way back (their results are being copied out): a
        // leader that finds fewer waiters than that batch carried waits a moment for them — 8 % of the batch's time, 30-200 us —
        // instead of sweeping for half of the cohort now and the other half next (64 threads: 41 calls per batch without).
        if (!me.mask && s->last_batch_calls > 1 && s->waiting.size() + 1 < s->last_batch_calls) {
            const double us = std::min(200.0, std::max(30.0, 0.08 * s->last_batch_us));
            const uint32_t want = s->last_batch_calls;
            s->arrive_cv.wait_for(lk, std::chrono::microseconds((long)us), [&] { return s->waiting.size() + 1 >= want; });
        }
        // every waiting search that can share my sweeps comes along (same metric, no bitmap, up to one pass of queries; k within
        // the candidate-list path, which also bounds the merged result block: 128 x NMN_MAX_TOP_K entries)
        if (!me.mask && me.k <= NMN_MAX_TOP_K) {
            uint32_t total = me.nq;
            for (auto it = s->waiting.begin(); it != s->waiting.end();) {
                ShardedReq* r = *it;
                if (!r->solo && !r->mask && r->metric == me.metric && r->k <= NMN_MAX_TOP_K && total + r->nq <= kShardedBatchQueries) {
                    total += r->nq;
                    batch.push_back(r);
                    it = s->waiting.erase(it);
                } else {
                    ++it;
                }
            }
        }
    }
    nmn_status st;
    const auto t_batch = std::chrono::steady_clock::now();
    if (batch.size() == 1) {
        st = sharded_run(s, queries, nq, k, metric, mask, out_rows, out_scores, out_counts, stats);
    } else {
        uint32_t total = 0, kmax = 0;
        for (ShardedReq* r : batch) {
            total += r->nq;
            kmax = std::max(kmax, r->k);
        }
        s->cat_q.resize((size_t)total * s->dim);
        s->cat_rows.resize((size_t)total * kmax);
        s->cat_scores.resize((size_t)total * kmax);
        s->cat_counts.resize(total);
        size_t q0 = 0;
        for (ShardedReq* r : batch) {
            memcpy(s->cat_q.data() + q0 * s->dim, r->queries, (size_t)r->nq * s->dim * sizeof(float));
            q0 += r->nq;
        }
        nmn_search_stats bst{};
        st = sharded_run(s, s->cat_q.data(), total, kmax, metric, nullptr, s->cat_rows.data(), s->cat_scores.data(),
                         s->cat_counts.data(), &bst);
        const std::string err = st != NMN_OK ? std::string(nmn_last_error()) : std::string();
        // every caller gets the first k_i entries of its queries' lists: the same total order, so what it gets alone
        q0 = 0;
        for (ShardedReq* r : batch) {
            if (st == NMN_OK) {
                for (uint32_t q = 0; q < r->nq; q++) {
                    const size_t src = (q0 + q) * (size_t)kmax, dst = (size_t)q * r->k;
                    const uint32_t cnt = std::min(s->cat_counts[q0 + q], r->k);
                    memcpy(r->out_rows + dst, s->cat_rows.data() + src, (size_t)r->k * 8);
                    memcpy(r->out_scores + dst, s->cat_scores.data() + src, (size_t)r->k * 4);
                    for (uint32_t i = cnt; i < r->k; i++) {  // (entries past the list's end: the padding of a lone call)
                        r->out_rows[dst + i] = UINT64_MAX;
                        uint32_t ninf = 0xFF800000u;
                        memcpy(&r->out_scores[dst + i], &ninf, 4);
                    }
                    r->out_counts[q] = cnt;
                }
                if (r->stats) *r->stats = bst;
            }
            q0 += r->nq;
        }
        std::lock_guard<std::mutex> lk(s->mu);
        s->merged_batches++;
        s->merged_calls += batch.size();
        for (ShardedReq* r : batch) {
            if (r == &me) continue;
            r->status = st;
            r->err = err;
            r->done = true;
            r->cv.notify_one();
        }
        if (st != NMN_OK) set_error(st, err.c_str());
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->last_batch_calls = (uint32_t)batch.size();
        s->last_batch_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_batch).count();
    }
    sharded_release(s);
    return st;
}

extern "C" nmn_status nmn_sharded_coalesce_stats(nmn_sharded* s, uint64_t* batches, uint64_t* calls) {
    if (!s) return set_error(NMN_ERR_INVALID_ARGUMENT, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    if (batches) *batches = s->merged_batches;
    if (calls) *calls = s->merged_calls;
    return NMN_OK;
}

extern "C" nmn_status nmn_sharded_last_gather_ms(const nmn_sharded* s, float* ms) {
    if (!s || !ms) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(const_cast<nmn_sharded*>(s)->mu);
    *ms = s->last_gather_ms;
    return NMN_OK;
}
