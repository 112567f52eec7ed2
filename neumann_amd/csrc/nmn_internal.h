// nmn_internal.h — shared declarations of libneumann_gpu.so (gfx950 only).
//
// Pipeline of one SIMILAR TOP-K call:
//   qprep    pad queries, |q| in reference order, per-query error margins          (nmn_exact.hip)
//   scan     stream the row-major f32 corpus once: approximate score per row +
//            per-64-row tile maximum                                                (nmn_scan.hip)
//   select   per query: 2-pass radix pick of a lower bound on the k-th best score,
//            collect every row within the rounding margin of it                    (nmn_select.hip)
//   [exact fallback: only for queries whose candidate list overflowed]             (nmn_exact.hip)
//   rescore  candidates re-scored bit-exactly in the reference's operation order   (nmn_exact.hip)
//   final    sort candidates by (exact score desc, row asc), emit top-k            (nmn_select.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neumann_gpu.h"

namespace nmn {

// internal metrics (beyond nmn_metric): both rank "nearest first" as a descending score
constexpr int kMetricNegL2 = 16;    // score = -sqrt(sum (q_i - v_i)^2), sequential sum (IVF list scan, ivf.rs:365-369)
constexpr int kMetricNegL2Sq = 17;  // score = -(sum (q_i - v_i)^2); exact kernels only (centroid ranking, ivf.rs:331-337)

constexpr uint32_t kTileRows = 64;        // rows per scan tile (one wave, 16 steps of 4 rows)
constexpr uint32_t kDefaultCandCap = 4096;
constexpr uint32_t kMaxScanWaves = 4096;   // scan waves per query sweep (= entries of the wave-max level)
constexpr uint32_t kKeyMasked = 0u;       // row does not take part (mask / beyond n_rows)
constexpr uint32_t kKeyNaN = 1u;          // NaN score: ranks below -inf
constexpr uint32_t kScoreSentinelBits = 0xFFFFFFFFu;  // scores[] entry of a non-participating row

// Per-query constants produced by qprep.
struct QInfo {
    float qmag;        // simd::magnitude(query), reference order
    float margin_abs;  // candidates: approx >= tau - margin_abs - |tau|*margin_rel
    float margin_rel;
    float pad;         // Euclidean score over the bf16 mirror: > 0 two-sided absolute error of the distance (VALU sweep);
                       // < 0 minus the two-sided absolute error of the SQUARED distance (matrix-core / 8-bit sweep); else 0
    float qscale;      // 8-bit sweep: s_q of the query's split q = s_q (h + l / 256) + e_q (nmn_scan_i8.hip); else 0
    float qq8;         // 8-bit sweep: |q~|^2 of the split query q~ = s_q (h + l / 256) (the Euclidean estimator |q~ - v~|^2);
                       // < 0: this query uses the OTHER estimator, |q|^2 + |v|^2 - 2 q~.v~ with the exact magnitudes
    float pad_sq;      // > 0 (8-bit Euclidean sweep): two-sided absolute error of the SQUARED distance ON TOP OF `pad` (distance space)
    float neg_d;       // 1: the score pad_sq applies to is -d (IVF list scan), else 1 / (1 + d)
};

// Per-query selection state shared by select / fallback / rescore / final.
struct QState {
    uint32_t cand_count;  // candidates written to cand_rows
    uint32_t overflow;    // 1 = candidate list overflowed -> exact fallback takes over; 2 = overflowed, but every row
                          // within the margin went to the crowd list (CrowdParams) and is re-scored from there
    uint32_t n_valid;     // participating keys seen by select (rows or tiles)
    uint32_t thr_key;     // collection threshold key (diagnostics)
};

// ---- score <-> order-preserving u32 key -----------------------------------------------------
// key order == score order for non-NaN scores; -0.0 and +0.0 map to one key (they compare equal,
// so ties between them must fall through to the row-id tie-break).
__host__ __device__ inline uint32_t f2u(float f) {
    union { float f; uint32_t u; } v; v.f = f; return v.u;
}
__host__ __device__ inline float u2f(uint32_t u) {
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}
__host__ __device__ inline uint32_t score_to_key(float s) {
    if (s != s) return kKeyNaN;
    uint32_t b = f2u(s);
    if ((b << 1) == 0u) b = 0u;  // -0.0 -> +0.0
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key_to_score(uint32_t k) {
    if (k <= kKeyNaN) return u2f(0x7FC00000u);
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return u2f(b);
}
__host__ __device__ inline uint32_t bits_to_key(uint32_t score_bits) {  // scores[] entry -> key
    return score_bits == kScoreSentinelBits ? kKeyMasked : score_to_key(u2f(score_bits));
}
constexpr uint32_t kKeyNegInf = 0x007FFFFFu;  // score_to_key(-inf)

// collection threshold key from the radix lower bound T and the query's error margins
static __device__ __forceinline__ uint32_t margin_key(uint32_t T, const QInfo& qi) {
    uint32_t Tc = kKeyNaN;
    if (T > kKeyNegInf) {
        float tau = key_to_score(T);
        float m_abs = qi.margin_abs;
        if (qi.pad_sq > 0.0f) {
            // 8-bit Euclidean sweep: the distance may be off by qi.pad AND, on top, its square by qi.pad_sq (both two-sided):
            // the threshold distance d grows to sqrt((d + pad)^2 + pad_sq); the score is 1 / (1 + d), or -d (IVF list scan)
            if (qi.neg_d != 0.0f) {
                const float d = fmaxf(-tau, 0.0f) + qi.pad;
                tau = -sqrtf(d * d + qi.pad_sq);
            } else if (tau > 0.0f) {
                const float d = fmaxf(1.0f / tau - 1.0f, 0.0f) + qi.pad;
                tau = 1.0f / (1.0f + sqrtf(d * d + qi.pad_sq));
            }
            const float thr = tau - fabsf(tau) * qi.margin_rel;
            uint32_t Tq = kKeyNaN;
            if (thr == thr) {
                Tq = score_to_key(thr);
                if (Tq > T) Tq = T;
                if (Tq < kKeyNaN) Tq = kKeyNaN;
            }
            return Tq;
        }
        // Euclidean score 1/(1+d) swept over the bf16 mirror: the distance may be off by up to qi.pad (two-sided), i.e.
        // the threshold distance 1/tau - 1 grows by qi.pad
        if (qi.pad > 0.0f && tau > 0.0f) tau = tau / (1.0f + qi.pad * tau);
        // ... swept by the matrix cores (|q|^2 + |v|^2 - 2 q.v): the SQUARED distance may be off by up to -qi.pad
        // (two-sided), i.e. the threshold distance d = 1/tau - 1 grows to sqrt(d^2 - qi.pad)
        if (qi.pad < 0.0f && qi.margin_abs < 0.0f) {  // ... with the score -d of the IVF list scan (flag: margin_abs < 0)
            const float d = fmaxf(-tau, 0.0f);
            tau = -sqrtf(d * d - qi.pad);
            m_abs = 0.0f;
        } else if (qi.pad < 0.0f && tau > 0.0f) {
            const float d = fmaxf(1.0f / tau - 1.0f, 0.0f);
            tau = 1.0f / (1.0f + sqrtf(d * d - qi.pad));
        }
        const float thr = tau - m_abs - fabsf(tau) * qi.margin_rel;
        if (thr == thr) {
            Tc = score_to_key(thr);
            if (Tc > T) Tc = T;
            if (Tc < kKeyNaN) Tc = kKeyNaN;
        }
    }
    return Tc;
}


// ---- layout of the approximate-score matrix -------------------------------------------------
// scores[tile][query-in-pass][64 rows]: tile-major, so a batched sweep writes ONE contiguous block per
// tile (64 queries x 256 B = 16 KiB) instead of 64 scattered 256-B pieces (that cost 1.5 ms of a 6.3 ms
// nq=64 sweep); with one query per pass it degenerates to plain row order.
__host__ __device__ inline uint64_t score_at(uint64_t row, uint32_t q, uint32_t nql) {
    return ((row >> 6) * nql + q) * 64ull + (row & 63ull);
}

// ---- kernel launchers (each defined in the .hip file named above) ---------------------------
struct ScanParams {
    const float* corpus;     // [rows][ld]
    const float* corpus_half;   // nullable: bf16 mirror (row stride ld/2 floats) the VALU sweep reads instead of `corpus`
    const int8_t* corpus_i8;    // 8-bit sweep (nmn_scan_i8.hip): int8 codes, row stride ld bytes
    const float* i8_scale;      // ... [rows] per-row scale s_r
    const float* i8_vv;         // ... [rows] |s_r c_r|^2: the squared magnitude of the row AS STORED (Euclidean estimator)
    const float* i8_cos;        // ... [rows] s_r / |v_r| (0 for a zero row): the cosine factor of the matrix-core 8-bit sweep
    const uint32_t* qi8;        // ... [nq][2][ld / 4]: the h plane and the l plane of every query (int8, zero padded)
    const QState* retry_state;  // nullable: sweep only the queries whose candidate list overflowed (f32 retry of a bf16 pass)
    const float* norms;      // [rows]
    const float* inv_norms;  // [rows] 1 / |v|, 0 for a zero row (matrix-core cosine sweep)
    const float* qpad;       // [nq][ld] zero padded
    const QInfo* qinfo;      // [nq]
    const uint64_t* mask;    // nullable, ceil(rows/64) words
    const uint64_t* const* qmasks;  // nullable [nq] (MFMA sweep only): one bitmap pointer PER QUERY (null entry = every
                                    // row takes part) — a batch of differently filtered searches in one sweep
    uint32_t* scores;        // score_at(row, q, nql): f32 bits (sentinel for non-participating rows)
    uint32_t* tmax;          // [nq][tmax_stride] tile maximum key (0 = empty tile)
    uint32_t* wmax;          // [nq][wmax_stride] maximum key over the tiles of each scan wave
    uint64_t tmax_stride;
    uint64_t wmax_stride;
    uint64_t n_rows;
    uint32_t nql;            // queries interleaved per tile in `scores` (= queries of this pass)
    // MFMA sweep only: sampling and score-write suppression (nmn_scan_mfma.hip)
    uint32_t tile_step;      // 1: every tile; S: sample pass over tiles 0,S,2S,.. (tile maxima only -> tmax[q][i])
    uint32_t fold_ny;        // > 1: 1-D grid folded over this many query blocks (set by the launcher)
    const uint32_t* skip_key;  // nullable [nq]: scores of a tile are written only if its maximum key >= skip_key[q]
    uint32_t* tmax_main;     // sampling pass (tile_step = S > 1), nullable: also FINISH the sampled tiles for the main sweep — their
    uint64_t tmax_main_stride;  // maxima into tmax_main[q][tile] and their scores written — so that the main sweep need not read them again
    uint32_t skip_sampled;   // main sweep: S > 0 = tiles that are multiples of S were finished by the sampling pass: not streamed
    uint32_t ld;             // floats per row, multiple of 8
    uint32_t n_tiles;
    uint32_t nq;
    uint32_t tiles_per_wave;
    uint32_t strided;        // VALU sweeps: 0 = wave w owns the tiles [w * tiles_per_wave, ...) (a contiguous range); 1 = the tiles
                             // w, w + W, w + 2W, ... (W = scan waves): what masked sweeps use — a selection that is a few RUNS of rows
                             // (an IVF list in the list-major copy, a WHERE over a time range) would otherwise land on a few waves
    uint32_t walk;           // masked 8-bit VALU sweep: 1 = the survivor walk (participating rows of up to 64 tiles listed and read four
                             // per step across tile borders, nmn_scan_i8.hip); 0 = tile by tile
    uint32_t bx_base, bx_count;  // MFMA sweep: this launch covers workgroups [bx_base, bx_base + bx_count) (0 = to the end)
    // MFMA sweep, ONE launch per batch (round 6): the bound that gates the score stores is derived INSIDE the sweep.  Every workgroup
    // publishes its running maximum per query (run_slots[q][workgroup], a plain store when it rises: the value the sweep leaves in
    // wmax at its end, made visible early); the run_S-th largest of a query's <= 1024 running maxima (run_S = k) is reached by k
    // different workgroups, hence k different tiles: a valid lower bound on the k-th best approximate score at ANY moment, rising as
    // the sweep proceeds.  Every wave re-reads the maxima of one query every 16 tiles (LDS-DMA: no VGPR-destination load in the
    // loop), selects the k-th largest and publishes margin_key(it) into run_bound[q] (atomicMax); the workgroups pick the current
    // run_bound up with the row magnitudes of each tile.  A tile writes its scores when its maximum reaches the bound it sees —
    // never above the final value of run_bound, which is what select_kernel is given as skip_key.  qprep_kernel resets both arrays.
    uint32_t* run_slots;     // [nq][1024] the workgroups' running maxima (keys; 0 = nothing yet)
    uint32_t* run_bound;     // [nq rounded up to 128] keys (kKeyNaN = no bound yet: every tile writes)
    uint32_t run_S;          // 0 = off; else the rank of the bound among the running maxima (= k)
    uint32_t i8_one_plane;   // 8-bit matrix-core sweep: 1 = the queries' h plane only (qprep approx_pass bit 16 measured the rounding accordingly)
    uint32_t run_dbg;        // measurement only (NMN_RUN_BOUND_DEBUG): 1 = decide the stores by skip_key, 2 = no slot atomics, 4 = no refresh, 8 = no bound DMA
    int metric;
};
hipError_t launch_scan(const ScanParams& p, hipStream_t s);
// ONE unmasked query over the f32 rows, f32 arithmetic, the rows streamed through the LDS-DMA ring (nmn_scan_ring.hip);
// tiles_per_wave = tiles per WORKGROUP there (wmax is indexed by workgroup, as on the matrix-core path)
bool scan_ring_supported(uint32_t ld, uint32_t dim, int metric);
hipError_t launch_scan_ring(const ScanParams& p, hipStream_t s);
// converts rows [row0, row0+n) into the bf16 mirror and folds their rounding-error norms into err_bits[0..1]
// (row_err2_scratch: n floats of device scratch)
hipError_t launch_half_rows(const float* corpus, float* half, uint32_t ld, uint64_t row0, uint64_t n, const float* norms,
                            float* row_err2_scratch, uint32_t* err_bits, hipStream_t s);
bool scan_half_supported(uint32_t ld, int metric);
// the 8-bit mirror (nmn_scan_i8.hip): 1-2 queries per sweep over int8 codes with a per-row scale
bool scan_i8_supported(uint32_t ld, uint32_t dim, int metric);
hipError_t launch_scan_i8(const ScanParams& p, hipStream_t s);
// quantizes rows [row0, row0+n) into q8 / scale and folds their error norms into err_bits[0..1] (row_err2_scratch: n floats)
hipError_t launch_q8_rows(const float* corpus, int8_t* q8, float* scale, float* vv, float* cosf, uint32_t ld, uint64_t row0, uint64_t n, const float* norms,
                          float* row_err2_scratch, uint32_t* err_bits, hipStream_t s);
// one pass over freshly written rows (nmn_ingest.hip): magnitudes in reference order + (half != nullptr) their bf16 mirror
// rows and the mirror's error norms folded into err_bits[0..1]
bool ingest_supported(uint32_t ld, uint32_t dim);
// exact scores of every row for flagged queries, a lane per row (nmn_ingest.hip): rows of whole 32-float stages
// flagged: 1 = queries with qstate.overflow == 1, 2 = overflow != 0 (null qstate: every query)
bool exact_rows_supported(uint32_t ld, uint32_t dim, int metric);
hipError_t launch_exact_rows(const float* corpus, const float* norms, uint32_t ld, uint64_t n_rows, const float* qpad, const QInfo* qinfo,
                             const QState* qstate, int flagged, const uint64_t* mask, const uint64_t* const* qmasks, uint32_t* scores,
                             uint32_t nql, uint32_t nq, int metric, hipStream_t s);
hipError_t launch_ingest(const float* corpus, uint32_t ld, uint64_t row0, uint64_t n, float* norms, float* inv_norms,
                         uint32_t* max_norm_bits, float* half, uint32_t* err_bits, hipStream_t s);
// ONE read of freshly written rows for a shard whose mirror is the 8-bit one (nmn_ingest.hip): magnitudes in reference order, int8
// codes, scales, |s c|^2, s / |v|, and the quantization-error maxima folded into err_bits[0..1] — launch_ingest + launch_q8_rows in one
bool ingest_q8_supported(uint32_t ld, uint32_t dim);
hipError_t launch_ingest_q8(const float* corpus, uint32_t ld, uint64_t row0, uint64_t n, float* norms, float* inv_norms, uint32_t* max_norm_bits,
                            int8_t* q8, float* scale, float* vv, float* cosf, uint32_t* err_bits, hipStream_t s);
hipError_t launch_read_probe(const float* corpus, uint64_t n_rows, uint32_t ld, float* sink, hipStream_t s);
hipError_t launch_ring_probe(const float* corpus, uint64_t n_rows, uint32_t ld, uint32_t tiles_per_wg, hipStream_t s);  // nmn_scan_ring.hip
// batched-query sweep on the matrix cores (nmn_scan_mfma.hip); tiles_per_wave = tiles per WORKGROUP there
bool scan_mfma_supported(uint32_t ld, uint32_t dim, int metric);
hipError_t launch_scan_mfma(const ScanParams& p, hipStream_t s);
// ... over the row-major f32 corpus itself (no mirror pointer set: launch_scan_mfma dispatches here), nmn_scan_mfma_f32.hip
hipError_t launch_scan_mfma_f32(const ScanParams& p, hipStream_t s);
// ... over the 8-bit mirror (p.corpus_i8 set: launch_scan_mfma dispatches on it): unmasked batches, rows of 256 .. 1536 elements
bool scan_mfma_i8_supported(uint32_t ld, uint32_t dim, int metric);
// ... with ONE query plane (p.i8_one_plane; cosine / dot product): nmn_scan_mfma_i8x.hip
bool scan_mfma_i8_one_plane_supported(int metric);
hipError_t launch_scan_mfma_i8_one(const ScanParams& p, hipStream_t s);

struct SelectParams {
    const uint32_t* scores;  // score_at(row, q, nql)
    const uint32_t* tmax;    // [nq][tmax_stride]
    const uint32_t* wmax;    // [nq][wmax_stride]
    uint64_t tmax_stride;
    uint64_t wmax_stride;
    uint32_t n_waves;        // scan waves that own tiles (<= kMaxScanWaves)
    uint32_t tiles_per_wave;
    uint32_t strided;        // as ScanParams::strided: tile j of wave w is j * n_waves + w instead of w * tiles_per_wave + j
    const QInfo* qinfo;
    QState* qstate;
    uint32_t* cand_rows;     // [nq][cand_cap]
    uint64_t n_rows;
    uint32_t nql;
    uint32_t n_tiles;
    uint32_t nq;
    uint32_t k;
    uint32_t cand_cap;
    const uint32_t* skip_key;  // nullable [nq]: scores of tiles whose maximum is below it were never written
    const uint32_t* k_extra;   // nullable: added to k for the threshold rank (rows forced to +inf, f64 similarity)
    int retry;                 // 1: second selection after the f32 retry sweep — only queries flagged `overflow` take part
    uint32_t* half_stats;      // nullable [2]: queries selected on the bf16 mirror / of those, queries that needed the retry
    int retry_follows;         // 1: an f32 retry sweep follows this selection (it may flag a query as not worth retrying)
    unsigned long long* fb_sync_reset;  // nullable [2]: counters of the fallback_select launch that follows, zeroed here
    uint32_t* crowd_count_reset;  // nullable [nq]: the crowd kernels' row totals, zeroed here (they follow on this stream)
    int crowd_follows;         // 1: the crowd kernels run behind this selection (it may hand a query with many tiles over to them early)
    float* l2_hint;            // nullable: query 0's threshold distance is left here (feeds qprep's choice of the 8-bit Euclidean estimator)
    int count_overflows;       // 1: half_stats[1] counts the queries whose candidate list overflowed in THIS selection (batched 8-bit
                               // sweeps have no f32 retry whose selections could be counted)
    int flat;                  // set by launch_select: shards of <= 16 384 tiles select from all their tile maxima at once (NMN_NO_FLAT_SELECT=1: off)
    // Lone callers on LARGE shards (round 6): `split` workgroups per query, each holding a part of the query's tile maxima in LDS
    // (<= 16 384 tiles per part); the parts meet ONCE — per-thread group maxima folded by atomicMax into split_sg[q][1024], an
    // arrival counter — pick the same bound from the 1024 super-group maxima, then each compacts and gathers ITS tiles and appends
    // its candidates to the query's list; the last part to finish writes the query's state.  0 = one workgroup per query.
    uint32_t split;
    uint32_t* split_sg;        // [nq][1024] keys, zero between launches
    uint32_t* split_ctr;       // [nq][4]: [0] parts arrived at the bound, [2..3] one 64-bit word (parts finished << 32 | candidates reserved); zero between launches
};
hipError_t launch_select(const SelectParams& p, hipStream_t s);
hipError_t launch_count_untrusted(const float* norms, uint64_t n_rows, uint32_t* out, hipStream_t s);
// per query: skip_key[q] = margin_key(k-th largest of the sampled tile maxima) (kKeyNaN if fewer than k are valid)
// (combine_max: keep the larger of the bound already in skip_key and the new one)
hipError_t launch_sample_bound(const uint32_t* tmax_sample, uint64_t stride, uint32_t n_sample, const QInfo* qinfo,
                               uint32_t nq, uint32_t k, uint32_t* skip_key, hipStream_t s, int combine_max = 0);

// Crowd path: a query with more than cand_cap rows within the margin of its k-th score (masses of duplicates,
// near-duplicates) does not fall back to the exact scan of the whole shard at once: the rows with approximate key >=
// the selection's threshold — the only ones that can be in the answer — are written to a slice of a shared pool,
// re-scored exactly from there, and the top-k is selected among them.  Pool full (or slice too large) -> the query
// keeps overflow == 1 and takes the f32 retry / exact fallback as before.
struct CrowdParams {
    QState* qstate;
    const uint32_t* tmax;      // [nq][tmax_stride]
    uint64_t tmax_stride;
    const uint32_t* scores;    // approximate scores, score_at(row, q, nql)
    uint32_t nql, nq, n_tiles;
    uint64_t n_rows;
    uint32_t* count;           // [nq] rows found (zeroed between searches by the alloc kernel)
    uint32_t* offset;          // [nq] slice start in the pool
    uint32_t* fill;            // [nq] append cursor (unused since the per-workgroup slice parts: kept zeroed)
    uint32_t* wg_count;        // [nq][kCrowdMaxGrid] rows found per workgroup of the count launch = where each workgroup's part of the slice starts
    uint32_t* pool_rows;       // [pool_cap]
    float* pool_scores;        // [pool_cap] exact scores (crowd_rescore)
    uint32_t pool_cap;
};
constexpr uint32_t kCrowdMaxGrid = 512;  // workgroups per query of the count / fill launches
hipError_t launch_crowd_collect(const CrowdParams& p, hipStream_t s);  // count -> fill (every workgroup allocates the slices for itself)

// Device-wide exact-fallback selection (nmn_select.hip: fallback_select_kernel), shards of >= 2^18 rows
struct FallbackParams {
    QState* qstate;             // overflow == 1: exact scores of every row are in `scores`; set to 3 when the list is ready
    const uint32_t* scores;     // score_at(row, q, nql), exact
    uint32_t nql, nq, k;
    uint64_t n_rows;
    uint32_t* ghist;            // [6][2048] + 2 counters, scratch
    unsigned long long* list;   // [nq][NMN_MAX_TOP_K] composites (score key << 32 | ~row) of the top-k
    uint32_t* list_count;       // [nq]
    unsigned long long* sync;   // [2] grid-barrier arrival counter + abort flag, both zeroed before every launch
    unsigned long long timeout_ticks;  // patience of a barrier wait in 100 MHz wall-clock ticks (0: default, ~100 ms)
    uint32_t list_cap;          // entries per query in `list` (0: NMN_MAX_TOP_K)
    int all;                    // 1: every query is selected and qstate is not touched (the large-k path: nq = 1, k <= list_cap)
};
hipError_t launch_fallback_select(const FallbackParams& p, hipStream_t s);

struct FinalParams {
    const unsigned long long* fb_list;  // nullable: lists of fallback_select_kernel (queries with overflow == 3)
    const uint32_t* fb_count;
    const uint32_t* crowd_offset;  // nullable: slices of the crowd pool (queries with overflow == 2)
    const uint32_t* crowd_rows;
    const float* crowd_scores;
    const uint32_t* cand_rows;   // [nq][cand_cap]
    const float* cand_scores;    // [nq][cand_cap] exact
    QState* qstate;
    const uint32_t* scores;      // score_at(row, q, nql): EXACT scores of every row for overflowed queries
    uint32_t nql;
    uint64_t n_rows;
    uint64_t row_base;
    uint32_t nq, k, cand_cap;
    int short_chain;             // 1: a query whose candidate list overflowed is reported (out_counts[q] = 0xFFFFFFFF), not answered
    uint64_t* out_rows;          // [nq][k]
    float* out_scores;
    uint32_t* out_counts;
    // (nullable) a lone host caller polls a word of pinned host memory instead of synchronising the stream (as tiny_search_kernel's
    // caller does): the LAST workgroup of the launch to finish (done_ctr: zero between launches) stores `done_seq` there,
    // system-scope release, after every result store of the launch
    uint32_t* done_word;
    uint32_t* done_ctr;
    uint32_t done_seq;
};
hipError_t launch_final(const FinalParams& p, hipStream_t s);
struct RescoreParams;
// rescore + final as ONE launch for the short chain (nmn_exact.hip): the last workgroup to finish a query's candidates sorts and emits;
// a query flagged `overflow` is reported as out_counts[q] = 0xFFFFFFFF.  `ticket` [nq] must be zero (the kernel leaves it zero).
hipError_t launch_rescore_final(const RescoreParams& p, const FinalParams& f, uint32_t* ticket, hipStream_t s);

// list l of each field starts `list_stride_bytes` * l bytes after list 0 (0 = contiguous [list][nq][k])
hipError_t launch_merge(const uint64_t* rows, const float* scores, const uint32_t* counts, uint64_t list_stride_bytes,
                        uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* out_rows, float* out_scores,
                        uint32_t* out_counts, hipStream_t s);

// exact (reference-order) kernels
hipError_t launch_norms(const float* corpus, uint32_t ld, uint32_t dim, uint64_t row0, uint64_t n, float* norms,
                        float* inv_norms, uint32_t* max_norm_bits, hipStream_t s);
hipError_t launch_qprep(const float* queries, uint32_t nq, uint32_t dim, uint32_t ld, int metric,
                        const uint32_t* max_norm_bits, float* qpad, QInfo* qinfo, QState* qstate, int approx_pass,
                        hipStream_t s, const uint32_t* half_err_bits = nullptr, uint32_t* qi8 = nullptr,
                        const float* l2_hint = nullptr, QInfo* qinfo_plain = nullptr,  // qinfo_plain: also the approx_pass == 0 record (f32 retry)
                        uint32_t* run_slots = nullptr, uint32_t run_S = 0, uint32_t* run_bound = nullptr);  // (ScanParams::run_*: reset here)
// approx_pass bits: 1 = the sweep's copy of the query is rounded (bf16 on the MFMA sweep; with bit 4 the int8 split
// q = s_q (h + l / 256), written to qi8[q][2][ld] and QInfo.qscale), 2 = the sweep reads a mirror of the corpus
// (half_err_bits = that mirror's measured rounding errors: [0] max |e_r|, [1] max |e_r| / |v_r|); 0 = plain f32 sweep;
// 16 (with 4): ONE query plane — l = 0, the residual q - s_q h is the measured rounding (the matrix-core 8-bit sweep, ONE = true);
// 8 (with 4, Euclidean, the 1-2 query sweep): the estimator may be chosen per query from *l2_hint, the threshold distance of
// the shard's previous Euclidean selection (nmn_scan_i8.hip: "which Euclidean estimator")
struct RescoreParams {
    const float* corpus;
    const float* norms;
    const float* qpad;
    const QInfo* qinfo;
    const QState* qstate;
    const uint32_t* cand_rows;
    float* cand_scores;
    // crowd duty (queries with qstate.overflow == 2): exact score of the rows of the query's slice of the crowd pool
    const uint32_t* crowd_offset;
    const uint32_t* crowd_rows;
    float* crowd_scores;
    // exact-fallback duty (queries with qstate.overflow == 1): exact score of EVERY row -> scores
    const uint64_t* mask;
    const uint64_t* const* qmasks;  // nullable [nq]: per-query bitmaps (see ScanParams)
    uint32_t* scores;
    uint64_t n_rows;
    uint32_t nql;
    uint32_t ld, dim, nq, cand_cap;
    int metric;
    int skip_fallback;  // 1: the exact-fallback duty is served by another launch (launch_exact_l2_rows)
};
hipError_t launch_rescore(const RescoreParams& p, hipStream_t s);
// exact score of explicit (query,row) pairs: out[q][i] = score(q, rows[i])
hipError_t launch_score_rows(const float* corpus, const float* norms, const float* qpad, const QInfo* qinfo,
                             const uint64_t* rows, uint32_t n_rows, uint32_t nq, uint32_t ld, uint32_t dim,
                             int metric, float* out, hipStream_t s);
struct ExactScanParams {
    const float* corpus;
    const float* norms;
    const float* qpad;
    const QInfo* qinfo;
    const QState* qstate;    // nullable; when set only queries with overflow==1 are processed
    const uint64_t* mask;
    uint32_t* scores;
    uint64_t n_rows;
    uint32_t nql;
    uint32_t ld, dim, nq;
    int metric;
};
hipError_t launch_exact_scan(const ExactScanParams& p, hipStream_t s);
hipError_t launch_count_cmp(const uint32_t* scores, uint64_t n_rows, float score, unsigned long long* out2,
                            hipStream_t s);

// the whole search of a small shard in one launch (nmn_exact.hip: tiny_search_kernel); query in the kernel arguments, results
// written by the kernel into (mapped, pinned) host memory
constexpr int kTinyMaxDim = 768;
bool tiny_supported(uint64_t n_rows, uint32_t ld, uint32_t dim, uint32_t k);
void tiny_geometry(uint64_t n_rows, uint32_t k, uint32_t* grid, uint32_t* rows_per_wg, uint32_t* kcap);
hipError_t launch_tiny_search(const float* corpus, const float* norms, const uint64_t* mask_dev, uint64_t n_rows, uint64_t row_base,
                              uint32_t ld, uint32_t dim, uint32_t k, int metric, const float* query_host, unsigned long long* pool,
                              uint32_t* ticket, uint64_t* out_rows, float* out_scores, uint32_t* out_count, uint32_t* out_seq, uint32_t seq,
                              hipStream_t s);

// large-k path (nmn_sortk.hip): keys[] holds largek_sort_len(n_rows) u64 (next power of two >= max(n_rows, 4096));
// score_bits[] are exact scores in plain row order (kScoreSentinelBits = row does not take part)
uint64_t largek_sort_len(uint64_t n_rows);
hipError_t launch_largek(const uint32_t* score_bits, uint64_t n_rows, uint64_t* keys, uint32_t k, uint64_t row_base,
                         uint64_t* out_rows, float* out_scores, uint32_t* out_count, hipStream_t s,
                         uint32_t* sel_hist = nullptr, unsigned long long* sel_sync = nullptr);

// k-means training of an IVF index (nmn_kmeans.hip)
hipError_t launch_kmeans_update(const float* corpus, uint32_t ld, uint32_t dim, const uint32_t* members,
                                const uint64_t* offsets, uint32_t k, float* new_centroids, hipStream_t s);
hipError_t launch_kmeans_min_update(float* dist, const uint32_t* neg_bits, uint64_t n, hipStream_t s);

// synthetic data
hipError_t launch_synth_fill(float* corpus, uint32_t ld, uint32_t dim, uint64_t seed, uint64_t global_row0,
                             uint64_t local_row0, uint64_t n, hipStream_t s);

// error reporting shared by the translation units that export C ABI entry points (nmn_api.hip owns
// the thread-local message behind nmn_last_error)
nmn_status set_error(nmn_status code, const char* what);
nmn_status set_error_hip(hipError_t e, const char* what);

}  // namespace nmn
