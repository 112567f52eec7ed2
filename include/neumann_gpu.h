/*
 * neumann_gpu.h — C ABI of libneumann_gpu.so, the MI355X (gfx950) flat-scan index behind
 * Neumann's `vector_engine` SIMILAR TOP-K hot path.
 *
 * The reference (Shadylukin/Neumann, Rust, `unsafe_code = "forbid"`) has no FFI of its own; the
 * boundary it offers is the public API of crate `vector_engine`.  Every entry point below therefore
 * cites the reference code it REPLACES (paths relative to the reference root) and is exactly what
 * the `ffi` module of a GPU-enabled `vector_engine` crate binds (see INTEGRATION.md for the
 * `extern "C"` block and the safe `GpuFlatIndex` wrapper).
 *
 * Conventions: plain C, caller-owned buffers, opaque handles, `nmn_status` return (0 = ok,
 * negative = error; values mirror `VectorError`, vector_engine/src/lib.rs:101-149), no exceptions
 * or torch types across the boundary.  Streams are passed as `void*` (a `hipStream_t`; NULL = the
 * legacy default stream).  All `*_device` entry points are asynchronous on that stream.
 *
 * There is NO CPU fallback anywhere in this library: without a usable gfx950 device every compute
 * entry point returns NMN_ERR_NO_DEVICE.
 */
#ifndef NEUMANN_GPU_H
#define NEUMANN_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t nmn_status;

/* Status codes.  The first block mirrors VectorError (vector_engine/src/lib.rs:101-149). */
#define NMN_OK 0
#define NMN_ERR_NOT_FOUND (-1)            /* VectorError::NotFound            */
#define NMN_ERR_DIMENSION_MISMATCH (-2)   /* VectorError::DimensionMismatch   */
#define NMN_ERR_EMPTY_VECTOR (-3)         /* VectorError::EmptyVector         */
#define NMN_ERR_INVALID_TOP_K (-4)        /* VectorError::InvalidTopK         */
#define NMN_ERR_STORAGE (-5)              /* VectorError::StorageError (HIP runtime failure) */
#define NMN_ERR_CONFIGURATION (-6)        /* VectorError::ConfigurationError  */
#define NMN_ERR_COLLECTION_EXISTS (-7)    /* VectorError::CollectionExists    */
#define NMN_ERR_COLLECTION_NOT_FOUND (-8) /* VectorError::CollectionNotFound  */
#define NMN_ERR_SEARCH_TIMEOUT (-9)       /* VectorError::SearchTimeout       */
/* Shim-only codes (no VectorError counterpart). */
#define NMN_ERR_INVALID_ARGUMENT (-20)
#define NMN_ERR_NO_DEVICE (-21)
#define NMN_ERR_OUT_OF_MEMORY (-22)
#define NMN_ERR_TOP_K_TOO_LARGE (-23)     /* k > NMN_MAX_TOP_K */
#define NMN_ERR_CAPACITY (-24)            /* upload beyond capacity_rows */
#define NMN_ERR_BUFFER_TOO_SMALL (-25)

/* Largest k one search call accepts (single-workgroup final sort in LDS). */
#define NMN_MAX_TOP_K 4096u
/* Largest number of queries one search call accepts. */
#define NMN_MAX_QUERIES 1024u

/* DistanceMetric, same order as vector_engine/src/lib.rs:268-289. */
typedef enum nmn_metric {
    NMN_METRIC_COSINE = 0,
    NMN_METRIC_EUCLIDEAN = 1,
    NMN_METRIC_DOT_PRODUCT = 2
} nmn_metric;

typedef struct nmn_index nmn_index; /* opaque: one row-range shard resident on one GPU */

typedef struct nmn_index_desc {
    uint32_t dim;            /* vector dimension d (>0) */
    uint32_t flags;          /* NMN_INDEX_* bits, 0 = defaults */
    uint64_t capacity_rows;  /* rows this shard can hold (HBM is allocated up front) */
    uint64_t row_base;       /* global id of local row 0 (shard offset; §8e row-range sharding) */
    int32_t device;          /* HIP device ordinal, -1 = current device */
    uint32_t cand_cap;       /* per-query candidate capacity before the exact-fallback path; 0 = default 4096 */
} nmn_index_desc;

/* Timing / accounting of the most recent search on a workspace (nullable everywhere). */
typedef struct nmn_search_stats {
    uint64_t rows_scanned;        /* rows whose vectors were read (mask-excluded rows are not) */
    uint64_t bytes_scanned;       /* algorithmic bytes of the scan: rows_scanned * dim * 4 */
    uint32_t candidates_rescored; /* max over queries of rows re-scored in reference order */
    uint32_t fallback_queries;    /* queries that took the exact-fallback path */
    float scan_ms;                /* hipEvent span of the scan kernel(s); -1 if not timed */
    float total_ms;               /* hipEvent span scan+select+rescore+sort; -1 if not timed */
} nmn_search_stats;

/* ---- device / lifecycle ------------------------------------------------------------------- */

/* Number of usable HIP devices (0 and NMN_OK when none). */
nmn_status nmn_device_count(int32_t* n);

/* Human-readable text of a status code.  Mirrors `impl Display for VectorError`
 * (vector_engine/src/lib.rs:151-183) for the mirrored codes. */
const char* nmn_status_str(nmn_status s);

/* Text of the last HIP/runtime failure on the calling thread ("" if none). */
const char* nmn_last_error(void);

/* Library version "major.minor.patch". */
const char* nmn_version(void);

/* Allocate one shard: corpus[capacity_rows][ld] f32 row-major (ld = dim rounded up to 4, zero
 * padded), norms[capacity_rows] f32.  Replaces the per-row `TensorStore` reads of the hot loop
 * (vector_engine/src/lib.rs:2121-2138; tensor_store/src/lib.rs:948-963) by one resident matrix. */
nmn_status nmn_index_create(const nmn_index_desc* desc, nmn_index** out);
nmn_status nmn_index_destroy(nmn_index* idx);

/* Copy n rows (row-major n x dim f32, HOST memory) into local rows [row0, row0+n) and compute
 * their magnitudes in reference order (`simd::magnitude`, tensor_store/src/hnsw.rs:198-229), so
 * the scan never recomputes |v| (the reference does, per row per query: lib.rs:2257-2266).
 * Rows above the current row count extend it; gaps are not allowed (row0 <= rows). */
nmn_status nmn_index_upload(nmn_index* idx, const float* rows_host, uint64_t row0, uint64_t n);
/* Same, rows already in DEVICE memory (row-major n x dim, tightly packed). Asynchronous. */
nmn_status nmn_index_upload_device(nmn_index* idx, const float* rows_dev, uint64_t row0, uint64_t n,
                                   void* stream);
/* Truncate/extend the logical row count without touching data (rows <= capacity). */
nmn_status nmn_index_set_rows(nmn_index* idx, uint64_t rows);

uint64_t nmn_index_rows(const nmn_index* idx);
uint32_t nmn_index_dim(const nmn_index* idx);
uint64_t nmn_index_row_base(const nmn_index* idx);
/* Device pointers of the resident data (for zero-copy producers and for tests). */
const float* nmn_index_corpus_device(const nmn_index* idx, uint32_t* ld_out);
const float* nmn_index_norms_device(const nmn_index* idx);

/* ---- the hot path ------------------------------------------------------------------------- */

/* SIMILAR TOP-K over the shard.  Replaces `search_similar` / `search_similar_with_metric` /
 * `search_in_collection` scoring + full sort + truncate (vector_engine/src/lib.rs:1950-2101,
 * 1585-1689) and, with `mask`, the survivor scan of `search_with_pre_filter` (lib.rs:3514-3557).
 *
 *   queries  nq x dim f32, HOST.          k  1..NMN_MAX_TOP_K
 *   mask     nullable HOST bitmap, ceil(rows/64) u64 words, bit i of word i/64 (LSB first) = row i
 *            takes part (layout of relational_engine's selection bitmaps, simd.rs:6-311).
 *   out_rows   nq x k  global row ids (row_base + local row), best first; unused slots = UINT64_MAX
 *   out_scores nq x k  scores exactly as the reference computes them (compute_score,
 *            lib.rs:2231-2266; lane order hnsw.rs:168-229); unused slots = -inf
 *   out_counts nq      min(k, rows taking part)
 * Ranking: score descending, equal scores by ascending row id (the reference's own tie order is
 * unspecified: slab_router.rs:287-305).  Zero-magnitude handling is the caller's (lib.rs:1970-1974):
 * at this level a zero query under COSINE scores every row 0.0, as cosine_similarity does.
 * Synchronous: returns after the results are in the host buffers. */
nmn_status nmn_index_search(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k,
                            nmn_metric metric, const uint64_t* mask, uint64_t* out_rows,
                            float* out_scores, uint32_t* out_counts, nmn_search_stats* stats);

/* Same with every buffer in DEVICE memory; enqueues on `stream` and returns immediately.
 * Calls on one stream may be pipelined back to back (they share that stream's workspace). */
nmn_status nmn_index_search_device(nmn_index* idx, const float* queries_dev, uint32_t nq, uint32_t k,
                                   nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows_dev,
                                   float* out_scores_dev, uint32_t* out_counts_dev, void* stream);

/* Stats of the last search enqueued on `stream` (synchronises that stream). */
nmn_status nmn_index_last_stats(nmn_index* idx, void* stream, nmn_search_stats* stats);
/* Turn hipEvent timing of the scan kernel on/off for `*_device` searches (default off). */
nmn_status nmn_index_set_timing(nmn_index* idx, int32_t enabled);

/* Reference-order scores of arbitrary rows (the exact-rescore kernel exposed on its own):
 * out[q*n_rows + i] = compute_score(query q, row rows[i]).  HOST buffers.  Used by the parity
 * tests and by `compute_similarity`-style callers (lib.rs:2268-2290). */
nmn_status nmn_index_score_rows(nmn_index* idx, const float* queries, uint32_t nq, nmn_metric metric,
                                const uint64_t* local_rows, uint32_t n_rows, float* out_scores);

/* Count local rows whose reference-order score is > / == the given score (full exact pass; a
 * size-independent certificate for top-k results at scales the CPU oracle cannot reach). */
nmn_status nmn_index_count_exact(nmn_index* idx, const float* query, nmn_metric metric,
                                 const uint64_t* mask, float score, uint64_t* n_greater,
                                 uint64_t* n_equal);

/* ---- shard merge (multi-GPU) -------------------------------------------------------------- */

/* Merge `n_lists` per-shard top-k lists (each nq x k, padded as nmn_index_search pads) into one:
 * concatenate, order by (score desc, row asc), keep k — `ResultMerger::merge_top_k`
 * (query_router/src/distributed.rs:413-433).  Layout of the inputs: [list][query][k], i.e. what an
 * all-gather of per-rank outputs produces.  HOST version (router-side merge): */
nmn_status nmn_merge_topk_host(const uint64_t* rows, const float* scores, const uint32_t* counts,
                               uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* out_rows,
                               float* out_scores, uint32_t* out_counts);
/* DEVICE version, asynchronous on `stream` (merges the RCCL all-gather output in place on GPU). */
nmn_status nmn_merge_topk_device(const uint64_t* rows_dev, const float* scores_dev,
                                 const uint32_t* counts_dev, uint32_t n_lists, uint32_t nq, uint32_t k,
                                 uint64_t* out_rows_dev, float* out_scores_dev, uint32_t* out_counts_dev,
                                 void* stream);

/* Same merge over an all-gather of PACKED per-rank blocks: rank l's rows / scores / counts start at
 * (char*)rows_dev + l*list_stride_bytes etc., so one collective can move all three fields.  With
 * list_stride_bytes == 0 the layout is the contiguous one of nmn_merge_topk_device. */
nmn_status nmn_merge_topk_device_strided(const uint64_t* rows_dev, const float* scores_dev,
                                         const uint32_t* counts_dev, uint64_t list_stride_bytes, uint32_t n_lists,
                                         uint32_t nq, uint32_t k, uint64_t* out_rows_dev, float* out_scores_dev,
                                         uint32_t* out_counts_dev, void* stream);

/* ---- synthetic data (bench / tests) ------------------------------------------------------- */

/* value(seed,row,col): a counter-based generator that is bit-identical on host and device
 * (integer hash -> sum of four 16-bit uniforms -> one exact f32 scale; approx. N(0,1)).
 * `row` is the GLOBAL row id, so shards of one corpus agree with the unsharded corpus. */
float nmn_synth_value(uint64_t seed, uint64_t row, uint32_t col);
/* Fill host memory: out[i*dim + c] = value(seed, row0+i, c). */
nmn_status nmn_synth_fill_host(float* out, uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim);
/* Fill local rows [row0,row0+n) of the shard on the GPU with value(seed, row_base+row, col) and
 * compute their norms; no host traffic (the 10M x 768 corpus is 30.7 GB). */
nmn_status nmn_index_fill_synthetic(nmn_index* idx, uint64_t seed, uint64_t row0, uint64_t n);
/* Overwrite one local row from host memory (used to plant near-duplicates); recomputes its norm. */
nmn_status nmn_index_set_row(nmn_index* idx, uint64_t row, const float* vec_host);

#ifdef __cplusplus
}
#endif
#endif /* NEUMANN_GPU_H */
