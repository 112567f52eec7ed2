// nmn_index.h — the shard object behind the opaque `nmn_index` handle and the entry points other
// translation units of libneumann_gpu.so use (nmn_ivf.hip).  Internal: never installed.
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "nmn_internal.h"

using nmn::QInfo;
using nmn::QState;
using nmn::kDefaultCandCap;

// ---- workspace: everything one in-flight search on one stream needs ---------------------------
struct Workspace {
    hipStream_t stream = nullptr;
    uint32_t nq_cap = 0;       // queries per pipeline pass the buffers are sized for
    uint32_t cand_cap = 0;
    uint64_t score_stride = 0;
    uint32_t n_tiles_cap = 0;
    uint32_t ld = 0;
    uint32_t* scores = nullptr;
    uint32_t* tmax = nullptr;
    uint32_t* wmax = nullptr;
    uint32_t* tsample = nullptr;   // [nq][n_sample_cap] tile maxima of the sampling pass (batched sweep)
    uint32_t* skip_key = nullptr;  // [nq] score-write threshold of the batched sweep
    uint32_t* k_extra = nullptr;   // [1] rows forced into the candidates (f64 artifact similarity)
    // crowd path (CrowdParams): per-query counters [3][nq] and the shared pool of (row, exact score) pairs; only on
    // shards large enough for the exact scan of everything to hurt
    uint32_t* crowd_ctr = nullptr;
    uint32_t* crowd_rows = nullptr;
    float* crowd_scores = nullptr;
    uint32_t crowd_cap = 0;
    // device-wide exact-fallback selection (FallbackParams): histograms + counters, per-query lists, grid-barrier counter
    uint32_t* fb_hist = nullptr;
    unsigned long long* fb_list = nullptr;
    uint32_t* fb_count = nullptr;
    unsigned long long* fb_sync = nullptr;
    uint64_t n_sample_cap = 0;
    uint64_t tmax_stride = 0;
    float* qpad = nullptr;
    uint32_t* qi8 = nullptr;       // [nq][2][ld / 4]: int8 h / l planes of the queries (8-bit sweep, nmn_scan_i8.hip)
    QInfo* qinfo = nullptr;
    QInfo* qinfo_f32 = nullptr;    // margins of the f32 sweep, for the retry after an overflowing bf16 pass
    QState* qstate = nullptr;
    uint32_t* cand_rows = nullptr;
    float* cand_scores = nullptr;
    uint32_t* run_slots = nullptr;     // [nq_cap][256] the one-launch batched sweep's slot maxima (ScanParams::run_slots)
    uint32_t* run_bound = nullptr;     // [nq_cap rounded up to 128] ... and its published bounds
    uint32_t* split_sg = nullptr;      // [nq_cap][1024] super-group maxima of the split selection (SelectParams::split_sg), zero between launches
    uint32_t* split_ctr = nullptr;     // [nq_cap][4] its counters
    uint32_t* final_ticket = nullptr;  // [nq_cap] arrival counters of rescore_final_kernel (zero between launches)
    // staging for the host-buffer API
    float* h_queries = nullptr;  size_t h_queries_cap = 0;   // device copies of host inputs
    uint64_t* h_mask = nullptr;  size_t h_mask_cap = 0;
    uint64_t* h_out_rows = nullptr; float* h_out_scores = nullptr; uint32_t* h_out_counts = nullptr;
    size_t h_out_rows_cap = 0, h_out_scores_cap = 0, h_cnt_cap = 0;
    // host-buffer API: ONE packed device block [rows | scores | counts] and pinned host staging for the query and the
    // results, so a call is one true-async H2D, the pipeline, one D2H and one wait (pageable copies cost ~10 us each)
    uint8_t* h_pack = nullptr; size_t h_pack_cap = 0;        // device
    const uint64_t** h_qmasks = nullptr; size_t h_qmasks_cap = 0;  // device: per-query bitmap pointers of a merged batch
    // predicates of a batch: staged programs, result bitmaps [requests][words], counters [requests][1 + blocks]
    uint8_t* pred_block = nullptr; size_t pred_block_cap = 0;            // device
    uint64_t* pred_masks = nullptr; size_t pred_masks_cap = 0;           // device
    uint32_t* pred_ticket = nullptr; size_t pred_ticket_cap = 0;         // device, zero between launches: one arrival counter per predicate of a batch
    unsigned long long* pred_counts = nullptr; size_t pred_counts_cap = 0;  // (unused since round 4: the counters live in the tail of h_pack and come back with the results)
    uint8_t* pin_pred = nullptr; size_t pin_pred_cap = 0;                // pinned host staging of pred_block
    uint8_t* pin_in = nullptr; size_t pin_in_cap = 0;        // pinned host
    uint8_t* pin_out = nullptr; size_t pin_out_cap = 0;      // pinned host
    uint32_t done_seq = 0;            // sequence number of the last polled call on this workspace (FinalParams::done_*)
    uint32_t* done_ctr = nullptr;     // device: arrival counter of the polled final_kernel launch (zero between launches)
    uint32_t* poll_word_dev = nullptr;  // set by the host-buffer path around ONE search_enqueue call: the pinned word (device view) its final_kernel publishes into
    uint8_t* pin_in_dev = nullptr;    // the same blocks as the device sees them (zero-copy I/O of the host-buffer API: qprep reads the
    uint8_t* pin_out_dev = nullptr;   // queries from pinned host memory, the last kernel of the chain writes the results into it)
    // single-launch search of a small shard (tiny_search_kernel): candidate pool, ticket, and the result block the kernel
    // writes straight into pinned host memory ([rows 1024 x u64 | scores 1024 x f32 | count])
    unsigned long long* tiny_pool = nullptr;
    uint32_t* tiny_ticket = nullptr;
    uint8_t* tiny_out = nullptr;        // pinned host
    uint8_t* tiny_out_dev = nullptr;    // the same memory as the device sees it
    uint32_t tiny_seq = 0;              // sequence number of the last single-launch search (the kernel echoes it when done)
    uint64_t* h_rowlist = nullptr; size_t h_rowlist_cap = 0;
    float* h_scorelist = nullptr; size_t h_scorelist_cap = 0;
    unsigned long long* h_counts2 = nullptr;
    uint64_t* lk_keys = nullptr; size_t lk_keys_cap = 0;  // composite keys of the large-k path (k > NMN_MAX_TOP_K)
    // timing + stats of the last search
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // with timing on, the events around the sweep of the last kTimingHistory searches on this stream (nmn_index_scan_history: the
    // sweep's duration averaged over a timed loop, not just its last step)
    static constexpr uint32_t kTimingHistory = 64;
    hipEvent_t hist[2 * kTimingHistory] = {};
    uint64_t hist_head = 0;    // searches recorded so far
    uint64_t hist_read = 0;    // ... and handed out
    int timed = 0;             // nmn_index_set_timing level of the last search: 0 off, 1 every event, 2 the sweep's two only
    bool scan_ev_in_hist = false;  // the last search's sweep events are hist[] entries (the candidate pipeline), not ev[1..2] (large-k path)
    uint64_t seen_upload_seq = 0;  // last asynchronous upload this workspace's stream has been ordered behind
    bool allocated = false;  // every buffer of ws_alloc exists (set last; a partial allocation is rolled back)
    uint32_t last_nq = 0;
    uint64_t last_rows_scanned = 0;
    uint32_t last_elem_bytes = 4;  // bytes per corpus element the last sweep read (2 on the bf16 mirror, 1 on the 8-bit one)
    uint32_t last_sweep_kind = 0;      // NMN_SWEEP_* of the last search's (first pass's) sweep — reported, never re-derived by callers
    uint32_t last_sweep_launches = 0;  // launches of that sweep (sampling pass + bound kernels + sweep launches)
    bool last_masked = false;
};

// one host-buffer search call waiting for, or riding in, a batch (lives on its caller's stack)
struct HostReq {
    const float* queries;
    uint32_t nq, k;
    int metric;
    const uint64_t* mask;
    bool mask_on_device;
    uint64_t mask_rows;  // rows the mask selects when the caller knows (UINT64_MAX: unknown), for the mixing decision
    // a WHERE predicate instead of a ready bitmap (nmn_index_search_pred): the leader evaluates the predicates of its
    // whole batch in one launch on the batch's stream, right before the sweep
    const nmn_columns* pred_cols = nullptr;
    std::vector<uint8_t> pred_ops;     // compiled program (device op records)
    const uint64_t* pred_consts = nullptr;
    uint64_t pred_n_consts = 0;
    uint64_t* selected_out = nullptr;  // rows the predicate selected
    uint64_t* out_rows;
    float* out_scores;
    uint32_t* out_counts;
    nmn_search_stats* stats;
    nmn_status st = NMN_OK;
    std::string err;  // text of a failure, for the caller's thread-local nmn_last_error
    int slot = -1;    // host slot handed to this request when it is told to lead a batch
    // its caller sleeps on `cv` until the request is done (rode in somebody's batch) or told to lead the next one
    std::mutex m;
    std::condition_variable cv;
    bool done = false, lead = false;  // guarded by m
    bool local = false;  // rides in the batch of the request that brought it along (index_search_hostio_many): nobody waits on it
    bool own_batch = false;  // brings its own riders along: must LEAD (riding in somebody else's batch would leave them unserved)
};

struct nmn_index {
    uint32_t dim = 0, ld = 0;
    uint64_t cap = 0, cap_pad = 0, rows = 0, row_base = 0;
    int device = 0;
    uint32_t cand_cap = kDefaultCandCap;
    // short launch chain (host-buffer searches, nmn_api.hip): a shard whose searches keep overflowing their candidate lists (k = 1000
    // under an 8-bit margin: ~6000 rows within it) would pay a short pass AND the whole chain every time — after a flagged call the
    // next 256 searches enqueue the whole chain at once
    uint64_t short_calls = 0, short_off_until = 0;
    uint32_t ws_nq_limit = 0xFFFFFFFFu;  // queries per pipeline pass the device's free memory allowed (ws_alloc lowers it on OOM)
    uint64_t squeeze_calls = 0;          // searches since a mirror was declined / a pass was shrunk for lack of HBM: every 4096th clears
                                         // the verdicts (q8_failed, half_failed, ws_nq_limit) so that a TRANSIENT squeeze does not last
    bool no_single_launch = false;  // NMN_INDEX_NO_SINGLE_LAUNCH
    float* corpus = nullptr;
    float* half = nullptr;       // bf16 mirror of `corpus` every approximate sweep reads (half the bytes); lazy
    uint64_t half_rows = 0;      // rows [0, half_rows) of `half` are current
    bool half_failed = false;    // allocation failed once: stay on the f32 sweep
    bool mirror_off = false;     // nmn_index_set_mirror(idx, 0): every sweep reads the f32 corpus (SURVEY §8(d)'s bytes)
    bool i8_off = false;         // nmn_index_set_mirror(idx, 2): the bf16 mirror only, never the 8-bit one
    // The 8-bit mirror (nmn_scan_i8.hip): int8 codes + a scale per row, what sweeps of 1-2 queries read where the row
    // length allows it (whole 256-element groups).  Built on first use, extended / patched like `half`.  Its own on/off
    // switch (q8_*; same rule as the bf16 mirror's): a shard whose measured 8-bit margin keeps overflowing the candidate
    // lists goes back to the bf16 mirror for the next 8192 searches.
    int8_t* q8 = nullptr;
    float* q8_scale = nullptr;
    float* q8_vv = nullptr;             // |s_r c_r|^2 per row (the Euclidean estimator of the 8-bit sweep)
    float* q8_cos = nullptr;            // s_r / |v_r| per row (cosine factor of the batched 8-bit sweep)
    float* q8_l2_hint = nullptr;        // device [1]: threshold distance of the last Euclidean selection on the 8-bit mirror (qprep's estimator choice)
    uint32_t* q8_err_bits = nullptr;    // device [2]: max_r |e_r|, max_r |e_r| / |v_r|
    uint32_t* q8_stats = nullptr;       // device [2]: queries selected on the 8-bit mirror / of those, retried in f32
    uint64_t q8_rows = 0;
    bool q8_failed = false;
    uint32_t q8_seen[2] = {0, 0};
    uint64_t q8_calls = 0, q8_off_until = 0;
    uint64_t one_plane_off_until = 0;  // batches multiply both query planes again until q8_calls reaches this (their one-plane margin kept overflowing)
    bool one_plane_recent = false;     // a batch since the last look at q8_stats used one plane
    // Mirror on/off switch: data whose rounding margin keeps overflowing the candidate capacity (a row of enormous norm
    // under a Euclidean metric, ...) pays a bf16 pass AND an f32 retry per query.  select_kernel counts both in
    // half_stats; every 256th search the host reads them and, if more than half of the recent queries were retried, leaves
    // the mirror alone for the next 8192 searches.
    uint32_t* half_stats = nullptr;     // device [2]
    uint32_t half_seen[2] = {0, 0};     // counters at the last look
    uint64_t half_calls = 0, half_off_until = 0;
    uint32_t* half_err_bits = nullptr;  // device [2]: max_r |e_r| and max_r |e_r|/|v_r| of the mirror's rounding (f32 bits)
    float* half_scratch = nullptr;      // |e_r|^2 of the rows being converted; kept (hipMalloc / hipFree per store or per
    size_t half_scratch_cap = 0;        // search after a store would synchronise the whole device every time)
    float* norms = nullptr;
    float* inv_norms = nullptr;         // 1 / |v| (0 for a zero row): what the batched cosine sweep multiplies by (one rcp per ROW at
                                        // ingest instead of one per (row, query) in every sweep's epilogue)
    uint32_t* max_norm_bits = nullptr;
    // Sweeps of a large shard never run side by side: each is HBM-bound on its own, so two at once each take twice as long and
    // every query waits for both.  A search on another stream waits (on the device) for the previous search's SWEEP — not for its
    // selection / rescore tail, which runs under the next sweep (nmn_api.hip: sweep chain).
    hipEvent_t sweep_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t sweep_seq = 0;
    hipStream_t sweep_stream = nullptr;
    hipEvent_t upload_ev = nullptr;     // recorded behind the latest nmn_index_upload_device (nmn_api.hip: upload_fence_*)
    uint64_t upload_seq = 0;
    hipStream_t host_stream = nullptr;  // = host_slots[0]: uploads, exact helpers, and the first search slot
    std::mutex mu;       // guards every field below and all enqueueing; NOT held while a host-buffer search waits
    std::unordered_map<hipStream_t, Workspace*> ws;
    int timing = 0;  // nmn_index_set_timing: 0 off, 1 on, 2 sweep events only
    // Host-buffer searches from several threads overlap on the GPU: each takes one of kHostSlots (stream + workspace),
    // enqueues under `mu`, releases `mu` and waits for its own stream.  Anything that changes the shard (upload,
    // set_row, ...) first waits under `mu` until no slot is busy.
    static constexpr int kHostSlots = 4;
    hipStream_t host_slots[kHostSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool slot_busy[kHostSlots] = {false, false, false, false};
    int slots_busy = 0;
    std::condition_variable cv;
    // Request coalescing: host-buffer searches that arrive while the shard is busy wait in `host_queue`; when a slot
    // frees, the oldest waiter is woken to lead: it runs ITS request together with every queued one of the same
    // (metric, mask) as one query batch — one corpus sweep for up to 64-128 queries instead of one sweep
    // each — and hands the results out (each rider is woken on its own condition variable: no thundering herd).
    // Results do not depend on the batch (exact rescore), so callers cannot tell, except by the clock.
    static constexpr uint32_t kCoalesceQueries = 64;
    std::vector<HostReq*> host_queue;   // arrival order
    // A leader that finds fewer waiters than the previous batch carried gives the stragglers (callers of that batch still
    // waking up and coming back) a moment to arrive, instead of sweeping the shard for itself alone.
    uint32_t last_batch_requests = 0;
    int gathering = 0;                  // leaders inside their gather window (arrivals then signal gather_cv)
    std::condition_variable gather_cv;
    int writers_waiting = 0;            // uploads etc. waiting for the slots to drain: no new batch starts meanwhile
    uint64_t coalesced_batches = 0, coalesced_requests = 0;  // batches of >= 2 requests, and the requests in them
};

struct nmn_columns;
namespace nmn {

// predicates of a query batch, evaluated on the batch's stream (nmn_columns.hip)
nmn_status columns_compile(const nmn_columns* c, const nmn_pred_op* prog, uint32_t n_ops, uint64_t n_consts,
                           uint64_t n_rows, std::vector<uint8_t>* ops_bytes);
size_t pred_desc_bytes();
size_t pred_op_bytes();
void pred_desc_write(uint8_t* dst, uint32_t ops_off, uint32_t n_ops, uint32_t consts_off, uint64_t* mask,
                     unsigned long long* counts, unsigned long long* host_total = nullptr,  // host_total: nullable, pinned host memory (device view)
                     uint32_t* ticket = nullptr);  // ticket: nullable, zero between launches (the last block sums the partials: launch_pred_batch(.., true))
uint32_t pred_batch_blocks(uint64_t n_rows, uint32_t n_prog);
hipError_t launch_pred_batch(const nmn_columns* c, const uint8_t* dev_block, uint32_t n_prog, uint64_t n_rows,
                             hipStream_t s, bool counts_by_ticket = false);
uint64_t columns_words(const nmn_columns* c);
int columns_device(const nmn_columns* c);

// nmn_index_search / nmn_index_search_dmask with the internal metrics allowed (host queries and outputs)
nmn_status index_search_hostio(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k, int metric,
                               const uint64_t* mask, bool mask_on_device, uint64_t* out_rows, float* out_scores,
                               uint32_t* out_counts, nmn_search_stats* stats, uint64_t mask_rows = UINT64_MAX);

// Several single-query searches of ONE caller as one batch (each with its own device bitmap): what the coalescer does for
// concurrent callers, for a caller that has all its queries at hand (the list scans of an IVF chunk).  n <= the value
// index_hostio_many_capacity returns (0: this shard's batches cannot carry a bitmap per query — search them one by one).
struct HostSearchSpec {
    const float* query;      // host, dim floats
    const uint64_t* mask;    // device bitmap (nullable)
    uint64_t mask_rows;      // rows it selects (UINT64_MAX: unknown)
    uint64_t* out_rows;      // host [k]
    float* out_scores;       // host [k]
    uint32_t* out_count;     // host
};
uint32_t index_hostio_many_capacity(const nmn_index* idx, int metric, uint32_t k);
nmn_status index_search_hostio_many(nmn_index* idx, const HostSearchSpec* specs, uint32_t n, uint32_t k, int metric,
                                    nmn_search_stats* stats);

// nmn_index_search_device with the internal metrics allowed (everything in device memory, asynchronous)
// short_chain: for callers that wait for the answer on the host — only the five launches every search needs; a query whose
// candidate list overflowed comes back as out_counts[q] == 0xFFFFFFFF and must be searched again WITHOUT short_chain
nmn_status index_search_device(nmn_index* idx, const float* queries_dev, uint32_t nq, uint32_t k, int metric,
                               const uint64_t* mask_dev, uint64_t* out_rows_dev, float* out_scores_dev,
                               uint32_t* out_counts_dev, hipStream_t stream, bool short_chain = false);
// a caller of index_search_device(short_chain = true) found a flagged query: the shard leaves the short chain alone for a while
void index_short_chain_flagged(nmn_index* idx);

}  // namespace nmn
