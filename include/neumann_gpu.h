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
#define NMN_ERR_IO (-10)                  /* VectorError::IoError             (index files: open / read / write) */
#define NMN_ERR_SERIALIZATION (-11)       /* VectorError::SerializationError  (index files: not decodable / corrupt) */
/* Shim-only codes (no VectorError counterpart). */
#define NMN_ERR_INVALID_ARGUMENT (-20)
#define NMN_ERR_NO_DEVICE (-21)
#define NMN_ERR_OUT_OF_MEMORY (-22)
#define NMN_ERR_TOP_K_TOO_LARGE (-23)     /* reserved (k > NMN_MAX_TOP_K is served by the large-k path) */
#define NMN_ERR_CAPACITY (-24)            /* upload beyond capacity_rows */
#define NMN_ERR_BUFFER_TOO_SMALL (-25)

/* Largest k the candidate pipeline serves (single-workgroup final sort in LDS).  Larger k is legal — the
 * reference sorts everything and truncates (lib.rs:2026-2034) — and takes the large-k path: exact score
 * of every row, full device sort, first k (neumann_amd/csrc/nmn_sortk.hip). */
#define NMN_MAX_TOP_K 4096u
/* Largest number of queries one search call accepts. */
#define NMN_MAX_QUERIES 1024u

/* DistanceMetric, same order as vector_engine/src/lib.rs:268-289. */
typedef enum nmn_metric {
    NMN_METRIC_COSINE = 0,
    NMN_METRIC_EUCLIDEAN = 1,
    NMN_METRIC_DOT_PRODUCT = 2,
    /* Not a vector_engine metric: tensor_blob's artifact similarity (tensor_blob/src/lib.rs:591-625), i.e.
     * SparseVector::from_dense(a).cosine_similarity(&SparseVector::from_dense(b)) — f64 dot and magnitudes over
     * the non-zero positions, NaN/Inf -> 0, clamped to [-1, 1], rounded once to f32 (sparse_vector.rs:583-599). */
    NMN_METRIC_SPARSE_COSINE_F64 = 3
} nmn_metric;

typedef struct nmn_index nmn_index; /* opaque: one row-range shard resident on one GPU */

/* nmn_index_desc.flags.  By default a row length is padded up to the next multiple of 128 elements only when that costs at
 * most 1/8 more bytes per sweep (1000 -> 1024); with NMN_INDEX_WIDE_ROWS also when it costs up to 1/2 more (300 -> 384,
 * 200 -> 256, 100 -> 128): batches of queries and concurrent callers then take the matrix-core sweep (up to 128 queries per
 * corpus read instead of 4) at the price of that much more HBM and single-query sweep time.  Results are the same. */
#define NMN_INDEX_WIDE_ROWS 1u
/* Shards of <= 65 536 rows (<= 64 MiB, dim <= 768) answer a single-query nmn_index_search (k <= 1024) with ONE kernel launch:
 * exact reference-order scores of every row, selection and sort in that launch, the query in the kernel arguments and the
 * result written straight into pinned host memory (neumann_amd/csrc/nmn_exact.hip: tiny_search_kernel).  With this flag
 * such searches take the general pipeline (sweep, select, rescore, final) instead: same answers; for tests and A/B runs. */
#define NMN_INDEX_NO_SINGLE_LAUNCH 2u

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
    uint64_t bytes_scanned;       /* algorithmic bytes of the scan: rows_scanned * dim * (4: f32 corpus, 2: bf16 mirror, 1: 8-bit mirror) */
    uint32_t candidates_rescored; /* max over queries of rows re-scored in reference order */
    uint32_t fallback_queries;    /* queries that took the exact-fallback path */
    float scan_ms;                /* hipEvent span of the scan kernel(s); -1 if not timed */
    float total_ms;               /* hipEvent span scan+select+rescore+sort; -1 if not timed */
    uint32_t sweep_kind;          /* NMN_SWEEP_*: which kernel streamed the rows for this search (the library's own dispatch decision) */
    uint32_t sweep_launches;      /* kernel launches of that sweep per query pass (sampling pass and bound kernels included) */
} nmn_search_stats;

/* nmn_search_stats.sweep_kind: the kernel that read the corpus (or its mirror) for the last search. */
#define NMN_SWEEP_NONE 0u       /* no rows (empty shard) */
#define NMN_SWEEP_RING_F32 1u   /* nmn::scan_ring_kernel: one unmasked query, f32 rows through the LDS-DMA ring (SURVEY §8(d) headline) */
#define NMN_SWEEP_VALU_F32 2u   /* nmn::scan_kernel over the f32 rows (bitmaps, 2-4 queries, small shards) */
#define NMN_SWEEP_VALU_BF16 3u  /* nmn::scan_kernel over the bf16 mirror */
#define NMN_SWEEP_VALU_I8 4u    /* nmn::scan_i8_kernel over the 8-bit mirror */
#define NMN_SWEEP_MFMA_F32 5u   /* nmn::scan_mfma_kernel over the f32 rows (rounded to bf16 in registers) */
#define NMN_SWEEP_MFMA_BF16 6u  /* nmn::scan_mfma_kernel over the bf16 mirror */
#define NMN_SWEEP_MFMA_I8 7u    /* nmn::scan_mfma_kernel over the 8-bit mirror */
#define NMN_SWEEP_EXACT 8u      /* exact reference-order scores of every row, no approximate sweep (tiny_search_kernel, large-k path) */
/* Name of a NMN_SWEEP_* value ("ring_f32", "valu_f32", "valu_bf16", "valu_i8", "mfma_f32", "mfma_bf16", "mfma_i8", "exact", "none"). */
const char* nmn_sweep_kind_str(uint32_t sweep_kind);

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

/* Allocate one shard: corpus[capacity_rows][ld] f32 row-major (ld = dim rounded up to 8, or to the next multiple of 128 if that is at most 1/8 more; zero
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
/* Elements per stored row including the zero padding (>= dim; a multiple of 128 when batches take the matrix-core sweep). */
uint32_t nmn_index_row_stride(const nmn_index* idx);
uint64_t nmn_index_row_base(const nmn_index* idx);
/* Device pointers of the resident data (for zero-copy producers and for tests). */
const float* nmn_index_corpus_device(const nmn_index* idx, uint32_t* ld_out);
const float* nmn_index_norms_device(const nmn_index* idx);

/* ---- the hot path ------------------------------------------------------------------------- */

/* SIMILAR TOP-K over the shard.  Replaces `search_similar` / `search_similar_with_metric` /
 * `search_in_collection` scoring + full sort + truncate (vector_engine/src/lib.rs:1950-2101,
 * 1585-1689) and, with `mask`, the survivor scan of `search_with_pre_filter` (lib.rs:3514-3557).
 *
 *   queries  nq x dim f32, HOST.          k  >= 1 (k > NMN_MAX_TOP_K: large-k path, exact scores + select/sort per query)
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
/* With timing on (nmn_index_set_timing): the sweep durations (HIP events on the launch stream, around the dominant kernel) of
 * the searches enqueued on `stream` since the last call — at most the 64 most recent, oldest first; synchronises the stream.
 * What bench.py averages over its timed loop for `roofline.achieved`. */
nmn_status nmn_index_scan_history(nmn_index* idx, void* stream, float* scan_ms, uint32_t cap, uint32_t* n_out);
/* Which matrix the approximate sweep streams.  enabled == 1 (default): the smallest mirror that serves the call —
 *   the shard's 8-BIT mirror (int8 codes with a scale per row, 1 byte per corpus element) for sweeps of 1-2 queries over rows
 *     whose stride is a multiple of 128 elements up to 4096 (128, 256, 384, ... — nmn_index_create pads row lengths just short
 *     of one up to it), and for query batches (matrix cores) where the stride is a multiple of 256 up to 1536, or 2048, or 3072
 *     (3072 under NMN_METRIC_EUCLIDEAN stays on the bf16 mirror);
 *   else its bf16 mirror (2 bytes per element; every row stride, VALU sweeps and — strides of 128 .. 768, 1024, 1280, 1536,
 *     2048, 3072, 4096 — the matrix cores);
 *   else the f32 rows.
 * A shard keeps ONE mirror by default: the one built while its rows arrive (the 8-bit one where the stride allows it: 5 bytes
 * per element resident with the f32 rows; nmn_index_hbm_bytes reports it), and builds the other only when a call needs it.  A
 * mirror that does not fit the device's free memory (with room to spare for workspaces) is not built: the call is served from
 * the next one down — never an error.  Mirrors carry their MEASURED rounding error into the candidate margin, and a shard whose
 * 8-bit margin keeps overflowing the candidate lists returns to the bf16 mirror by itself.
 * enabled == 2: the bf16 mirror only.  enabled == 0: the row-major f32 corpus itself — the sweep SURVEY.md §8(d) prices at
 * rows * dim * 4 bytes per query (bench.py's headline: `value`, `roofline`); query batches then take the matrix-core sweep over
 * the f32 rows themselves (rounded to bf16 in registers: the rows' bytes once per 64-128 queries, nmn_scan_mfma_f32.hip).  Any other
 * value: NMN_ERR_INVALID_ARGUMENT.  Results are identical in every mode (every candidate is re-scored from the f32 corpus in
 * the reference's order). */
nmn_status nmn_index_set_mirror(nmn_index* idx, int32_t enabled);
/* Device memory the shard holds for its rows: corpus_bytes = the f32 rows (capacity x stride x 4), mirror_bytes = the 8-bit and
 * bf16 mirrors that exist right now, with their per-row factors; per_row_bytes = magnitudes and their reciprocals.  Workspaces
 * (per stream, sized by the largest search seen) are not counted.  Any pointer may be null.  No side effects. */
nmn_status nmn_index_hbm_bytes(nmn_index* idx, uint64_t* corpus_bytes, uint64_t* mirror_bytes, uint64_t* per_row_bytes);
/* A mirror that was declined (or a query pass that was shrunk) because the device was short of memory stays declined until the
 * verdict expires (every 4096 searches) — or until this call: the next search that wants the mirror / the larger pass asks the device
 * again.  For a host that has just freed memory on the device (dropped a sibling shard, a collection). */
nmn_status nmn_index_retry_declined(nmn_index* idx);
/* hipEvent timing of searches (default 0 = off).  1: events at the start and end of the pipeline and around its sweep (scan_ms,
 * total_ms of nmn_index_last_stats).  2: the two events around the sweep only (scan_ms, nmn_index_scan_history) — what a timed
 * loop can afford on a small shard: every event is a packet of its own in the queue (1M x 768: four more cost ~9 % of a step). */
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

/* Measurement aid: a pure read sweep over the shard, best of `reps` runs, in GB/s: the data movement of the sweep a single f32 query
 * takes on this shard with the arithmetic and every store removed — the LDS-DMA ring of nmn::scan_ring_kernel where that kernel
 * serves (>= 4096 tiles, stride a multiple of 128 up to 4096), nmn::scan_kernel's register loads elsewhere.  What a read-only
 * kernel can reach on this device; bench.py reports the sweep against it (`ring_only_read_ceiling`). */
nmn_status nmn_index_read_probe(nmn_index* idx, uint32_t reps, double* gbps_out);

/* Measurement aid: `threads` host threads call nmn_index_search(nq = 1, k, metric) in a loop for `seconds`, thread t
 * with query t of `queries` (HOST, threads x dim).  Reports calls per second, how many sweeps carried >= 2 calls and
 * how many calls rode in them, and how many of the checked answers differed bit-wise from the same query searched
 * alone before the threads started (must be 0).  bench.py reports it as `concurrent_callers`. */
nmn_status nmn_index_callers_probe(nmn_index* idx, const float* queries, uint32_t threads, uint32_t k, nmn_metric metric,
                                   double seconds, double* calls_per_s, uint64_t* merged_batches,
                                   uint64_t* merged_calls, uint64_t* mismatches);

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

/* ---- multi-GPU in ONE process: row-range shards over the GPUs of a node --------------------- */

/* The reference host is one process sharing an Arc<VectorEngine> (query_router/src/lib.rs:710); what it offers for
 * spreading a SIMILAR over several holders of the data is the distributed planner's scatter-gather — every shard runs
 * the query, ResultMerger::merge_top_k concatenates, sorts by score descending and truncates
 * (query_router/src/distributed.rs:173-180, 413-433).  nmn_sharded is that with GPUs as shards, behind one handle a
 * single host process drives (SURVEY.md §8b `nmn_index_desc{.., n_gpus, gpu_ids[]}`, §8e): shard g holds the global
 * rows [g*ceil(N/G), (g+1)*ceil(N/G)) on devices[g]; a search replicates the queries, runs the single-shard pipeline
 * on every device's own stream, gathers the per-shard top-k blocks with ONE collective — an RCCL all-gather over xGMI
 * (ncclCommInitAll communicators, grouped calls) when every shard has a device of its own, peer copies when several
 * LOGICAL shards share a device — and merges them on the device of shard 0 (nmn_merge_topk_device_strided's kernel).
 * The answer is the unsharded one: rows, scores, tie order (global top-k is a subset of the union of the local ones).
 * Calls on one handle are serialised; use one handle per concurrent client group, or the per-shard handles
 * (nmn_sharded_shard) with nmn_merge_topk_* directly. */
typedef struct nmn_sharded nmn_sharded;

#define NMN_MAX_SHARDS 64u
#define NMN_GATHER_AUTO 0u /* RCCL when the shards' devices are distinct (and there are >= 2), else peer copies */
#define NMN_GATHER_RCCL 1u /* ncclAllGather, one communicator rank per shard; needs distinct devices (1 shard is legal) */
#define NMN_GATHER_PEER 2u /* hipMemcpyPeerAsync of every block into the merging device's gather buffer */
/* How the global rows map to the shards.  RANGES (default): shard g holds [g*ceil(N/G), (g+1)*ceil(N/G)) of
 * the CAPACITY — the layout of a corpus loaded once (bench.py, config 4).  CYCLIC: 64-row blocks (one scan tile, one bitmap
 * word) are dealt round-robin, block b -> shard b % G, so the rows HELD are spread evenly however few of the capacity's rows
 * exist and wherever appends land — what a store that grows and shrinks needs (the engine's mirrors: with RANGES the n live
 * rows of a mirror with 50 % spare capacity filled shard 0, then shard 1, ... and left the last GPUs idle).  Same API, same
 * answers: row ids in and out are the global ones. */
#define NMN_SHARDED_LAYOUT_RANGES 0u
#define NMN_SHARDED_LAYOUT_CYCLIC 1u

typedef struct nmn_sharded_desc {
    uint32_t dim;            /* vector dimension d (>0) */
    uint32_t flags;          /* NMN_INDEX_* bits, applied to every shard */
    uint64_t capacity_rows;  /* rows of the WHOLE corpus; shard g gets the rows [g*ceil(N/G), ...) */
    uint64_t row_base;       /* global id of row 0 of the whole corpus */
    uint32_t n_shards;       /* G, 1..NMN_MAX_SHARDS */
    uint32_t gather;         /* NMN_GATHER_* */
    const int32_t* devices;  /* [n_shards] HIP device ordinal of each shard (repeats = logical shards on one GPU);
                                NULL = round-robin over the node's devices */
    uint32_t cand_cap;       /* as nmn_index_desc.cand_cap */
    uint32_t layout;         /* NMN_SHARDED_LAYOUT_*; 0 = contiguous row ranges */
} nmn_sharded_desc;

nmn_status nmn_sharded_create(const nmn_sharded_desc* desc, nmn_sharded** out);
nmn_status nmn_sharded_destroy(nmn_sharded* s);
/* nmn_index_upload over the GLOBAL row numbering: rows [row0, row0+n) (HOST, row-major n x dim) go to the shards whose
 * ranges they fall into.  Rows must arrive in global order (row0 <= nmn_sharded_rows: a gap is refused before any shard is
 * touched).  With shards on several devices the parts are copied side by side, one host thread per shard (the same threads
 * enqueue every shard's search pipeline at once; environment NMN_SHARDED_CREW=1|0 forces them on / off). */
nmn_status nmn_sharded_upload(nmn_sharded* s, const float* rows_host, uint64_t row0, uint64_t n);
/* nmn_index_fill_synthetic over the global numbering: the shards together hold exactly the unsharded corpus. */
nmn_status nmn_sharded_fill_synthetic(nmn_sharded* s, uint64_t seed, uint64_t row0, uint64_t n);
/* nmn_index_search over all shards: same arguments (HOST buffers; `mask` is ONE bitmap over the global rows, bit i =
 * row row_base + i, sliced per shard here), same outputs, same ranking.  stats: rows / bytes summed over the shards,
 * times = the slowest shard's (they run side by side).  Synchronous. */
nmn_status nmn_sharded_search(nmn_sharded* s, const float* queries, uint32_t nq, uint32_t k, nmn_metric metric,
                              const uint64_t* mask, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                              nmn_search_stats* stats);
uint32_t nmn_sharded_shards(const nmn_sharded* s);
uint64_t nmn_sharded_rows(const nmn_sharded* s);               /* rows held, all shards */
nmn_index* nmn_sharded_shard(nmn_sharded* s, uint32_t g);      /* the shard's own handle (owned by s) */
int32_t nmn_sharded_device(const nmn_sharded* s, uint32_t g);  /* HIP device of shard g */
uint32_t nmn_sharded_gather_mode(const nmn_sharded* s);        /* NMN_GATHER_RCCL or NMN_GATHER_PEER: what create chose */
uint32_t nmn_sharded_layout(const nmn_sharded* s);             /* NMN_SHARDED_LAYOUT_* in effect (1 shard: RANGES) */
/* global id (row_base included) of local row `local_row` of shard g under the handle's layout; UINT64_MAX for a bad shard */
uint64_t nmn_sharded_global_row(const nmn_sharded* s, uint32_t g, uint64_t local_row);
/* nmn_sharded_create on >= 2 shards ends with a SELF-TEST of the collective a search will use: every shard puts its rank
 * into its result block, the blocks travel through the very gather code path (grouped ncclAllGather on the communicators of
 * ncclCommInitAll, or the peer copies), and the merging device — with RCCL every device — must hold ranks 0..G-1 in order.
 * A failure is NMN_ERR_STORAGE with the reason in nmn_last_error(), at create time instead of inside the first search.
 * nmn_sharded_rccl_ranks: communicator ranks that took part in that all-gather (ncclCommCount of the handle's communicators,
 * cross-checked against the gathered ranks); 0 when the gather is peer copies. */
uint32_t nmn_sharded_rccl_ranks(const nmn_sharded* s);
nmn_status nmn_sharded_set_timing(nmn_sharded* s, int32_t enabled);  /* hipEvent timing of the shards' sweeps and of gather+merge */
nmn_status nmn_sharded_set_mirror(nmn_sharded* s, int32_t enabled);  /* nmn_index_set_mirror on every shard */
/* hipEvent span of the last search's collective + merge on the merging device (ms; -1 when not timed).  It starts when
 * shard 0's pipeline is done, so it includes waiting for slower shards. */
nmn_status nmn_sharded_last_gather_ms(const nmn_sharded* s, float* ms);
/* Concurrent callers: the handle may be searched from any number of threads (the reference shares one Arc<VectorEngine>,
 * query_router/src/lib.rs:710).  A search that arrives while another runs waits; when the running one ends the oldest
 * waiter leads the next sweep and takes every waiting search of the same metric (no bitmap, k <= NMN_MAX_TOP_K) along as ONE query batch
 * — up to 128 queries, k = the largest asked for; each caller receives the first k entries of its own queries' lists,
 * i.e. exactly what it gets alone.  Uploads run alone, in arrival order.  batches / calls: sweeps that carried two or
 * more calls, and the calls in them. */
nmn_status nmn_sharded_coalesce_stats(nmn_sharded* s, uint64_t* batches, uint64_t* calls);

/* ---- columnar metadata + WHERE-predicate programs ----------------------- */

/* Filtered SIMILAR in the reference evaluates the predicate per key on the host: one `store.get`
 * (BTreeMap lookup + TensorData clone) and one `evaluate_filter` walk per stored row
 * (vector_engine/src/lib.rs:3526-3530, 3582-3630).  Here the metadata fields of the mirrored rows
 * live as typed COLUMNS in HBM, aligned with the rows of an nmn_index, and the predicate is a
 * postfix program one kernel evaluates for every row, writing the selection bitmap the masked scan
 * consumes (same layout as relational_engine's bitmaps: bit i of word i/64, LSB first).
 *
 * A cell is (kind u8, payload u64).  Strings are dictionary-encoded per column by the caller; every
 * string predicate reaches the device as a bitset over dictionary ids (NMN_PRED_STRSET), so the
 * string comparison itself runs once per DISTINCT value on the host, never per row. */
typedef struct nmn_columns nmn_columns;

#define NMN_CELL_ABSENT 0u /* the row has no such field (TensorData::get -> None)        */
#define NMN_CELL_NULL 1u   /* ScalarValue::Null                                            */
#define NMN_CELL_BOOL 2u   /* ScalarValue::Bool,   payload 0/1                             */
#define NMN_CELL_INT 3u    /* ScalarValue::Int,    payload = the i64's bits                */
#define NMN_CELL_FLOAT 4u  /* ScalarValue::Float,  payload = the f64's bits                */
#define NMN_CELL_STRING 5u /* ScalarValue::String, payload = dictionary id in this column  */

/* Program opcodes: postfix over a 64-deep boolean stack; the program must leave exactly one value. */
#define NMN_PRED_TRUE 0u   /* push true                       (FilterCondition::True)                   */
#define NMN_PRED_FALSE 1u  /* push false                      (a field no row has)                      */
#define NMN_PRED_AND 2u    /* pop b, pop a, push a && b       (FilterCondition::And)                    */
#define NMN_PRED_OR 3u     /* pop b, pop a, push a || b       (FilterCondition::Or)                     */
#define NMN_PRED_EXISTS 4u /* push kind != ABSENT             (Exists, lib.rs:3602)                     */
#define NMN_PRED_CMP 5u    /* push cmp(cell, value)           (Eq/Ne/Lt/Le/Gt/Ge, lib.rs:3603-3620; the
                              typed comparison of compare_tensor_value_to_filter, lib.rs:3648-3670:
                              Int/Int, Float/Float, Float/Int, Int/Float (int side widened `as f64`),
                              Bool/Bool, Null/Null; anything else, or a NaN operand, is false) */
#define NMN_PRED_IN 6u     /* push any_j cell == value_j      (In, lib.rs:3627-3629); values are the
                              (kind,payload) pairs consts[a .. a+2*b) */
#define NMN_PRED_STRSET 7u /* push kind == STRING && bit payload of the bitset consts[a ..] (b = number
                              of ids covered)              (string Eq/Ne/Lt/Le/Gt/Ge, Contains,
                              StartsWith, string members of In; lib.rs:3621-3626, 3673-3692) */

/* cmp field of NMN_PRED_CMP, FilterCondition order (lib.rs:296-309). */
#define NMN_CMP_EQ 0u
#define NMN_CMP_NE 1u
#define NMN_CMP_LT 2u
#define NMN_CMP_LE 3u
#define NMN_CMP_GT 4u
#define NMN_CMP_GE 5u

typedef struct nmn_pred_op {
    uint32_t op;     /* NMN_PRED_*                                            */
    uint32_t cmp;    /* NMN_CMP_* (CMP only)                                  */
    uint32_t vkind;  /* NMN_CELL_* kind of the filter value (CMP only)        */
    uint32_t column; /* column id (EXISTS, CMP, IN, STRSET)                   */
    uint64_t a;      /* CMP: value payload; IN / STRSET: offset into consts   */
    uint64_t b;      /* IN: number of values; STRSET: number of ids covered   */
} nmn_pred_op;

/* A column set for `capacity_rows` rows on `device` (-1 = current).  All cells start ABSENT and the
 * row-validity bitmap starts all-zero. */
nmn_status nmn_columns_create(int32_t device, uint64_t capacity_rows, nmn_columns** out);
nmn_status nmn_columns_destroy(nmn_columns* cols);
/* Add one column (every cell ABSENT); its id is returned in *column_out. */
nmn_status nmn_columns_add(nmn_columns* cols, uint32_t* column_out);
uint32_t nmn_columns_count(const nmn_columns* cols);
/* Write cells [row0, row0+n) of one column from HOST arrays. */
nmn_status nmn_columns_write(nmn_columns* cols, uint32_t column, uint64_t row0, uint64_t n,
                             const uint8_t* kinds, const uint64_t* payloads);
/* Make every column's cell of `row` ABSENT (a key overwritten in place drops its old metadata:
 * `put` replaces the whole TensorData, tensor_store/src/lib.rs:927-938, vector_engine/src/lib.rs:3291-3307). */
nmn_status nmn_columns_clear_row(nmn_columns* cols, uint64_t row);
/* Write words [word0, word0+n) of the row-validity bitmap (rows that exist and are not deleted); every
 * evaluation is ANDed with it. */
nmn_status nmn_columns_write_valid(nmn_columns* cols, uint64_t word0, uint64_t n_words, const uint64_t* words);
/* Evaluate the program over rows [0, n_rows): the resulting bitmap stays in device memory
 * (nmn_columns_mask_device), *count_out = number of selected rows.  `prog` and `consts` are HOST
 * arrays.  Synchronous.  Replaces the `list_keys().filter(evaluate_filter_for_key)` stage of
 * search_with_pre_filter (lib.rs:3526-3530) and of search_filtered_in_collection (lib.rs:1776-1784). */
nmn_status nmn_columns_eval(nmn_columns* cols, const nmn_pred_op* prog, uint32_t n_ops, const uint64_t* consts,
                            uint64_t n_consts, uint64_t n_rows, uint64_t* count_out);
/* The same evaluation for concurrent callers (filtered searches of many threads under a shared lock): the result goes
 * to a bitmap of its own, *mask_out (device, ceil(capacity_rows/64) words), which the caller holds — together with the
 * stream and staging the evaluation ran on — as slot *slot_out until nmn_columns_eval_release.  Evaluations of
 * different slots overlap on the device; searches that pass different slot bitmaps to nmn_index_search_dmask share one
 * corpus sweep (request coalescing, one bitmap per query).  Must not run concurrently with the nmn_columns_write*
 * family (the engine's writers hold its exclusive lock). */
nmn_status nmn_columns_eval_acquire(nmn_columns* cols, const nmn_pred_op* prog, uint32_t n_ops, const uint64_t* consts,
                                    uint64_t n_consts, uint64_t n_rows, uint64_t* count_out, uint32_t* slot_out,
                                    const uint64_t** mask_out);
nmn_status nmn_columns_eval_release(nmn_columns* cols, uint32_t slot);
/* Filtered search in ONE call: evaluate `prog` over `cols` (rows [0, nmn_index_rows(idx)), same row numbering as the
 * shard) and search the rows it selects — `search_with_pre_filter` (lib.rs:3514-3557) end to end.  For concurrent
 * callers this is the fast form: the request coalescer hands the predicates of all the calls riding one query batch
 * to a single launch on the batch's stream, right before the ONE sweep that serves them (one bitmap per query), so
 * a filtered search waits for one batch, not for an evaluation and then a batch.  *selected_out (nullable) = rows the
 * predicate selected; results as nmn_index_search_dmask over that bitmap (k <= NMN_MAX_TOP_K).  queries HOST nq x dim. */
nmn_status nmn_index_search_pred(nmn_index* idx, nmn_columns* cols, const nmn_pred_op* prog, uint32_t n_ops,
                                 const uint64_t* consts, uint64_t n_consts, const float* queries, uint32_t nq, uint32_t k,
                                 nmn_metric metric, uint64_t* out_rows, float* out_scores, uint32_t* out_counts,
                                 uint64_t* selected_out, nmn_search_stats* stats);
/* Device bitmap of the last nmn_columns_eval, ceil(capacity_rows/64) words. */
const uint64_t* nmn_columns_mask_device(const nmn_columns* cols);
/* Device row-validity bitmap (a ready-made mask of the live rows). */
const uint64_t* nmn_columns_valid_device(const nmn_columns* cols);
/* Copy words of the last evaluation's bitmap to the host (tests). */
nmn_status nmn_columns_read_mask(nmn_columns* cols, uint64_t* out_words, uint64_t n_words);

/* nmn_index_search with the selection bitmap already in DEVICE memory (e.g. nmn_columns_mask_device);
 * queries and outputs are HOST buffers as in nmn_index_search.  Synchronous. */
nmn_status nmn_index_search_dmask(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                  nmn_metric metric, const uint64_t* mask_dev, uint64_t* out_rows,
                                  float* out_scores, uint32_t* out_counts, nmn_search_stats* stats);
/* The same, with the number of rows the bitmap selects (the predicate kernel's count) as a hint for the request
 * coalescer: concurrent searches with DIFFERENT device bitmaps share one sweep (one bitmap per query, every row read)
 * once their selectivities add up to a whole sweep; below that each is swept alone and skips what it excludes. */
nmn_status nmn_index_search_dmask_hint(nmn_index* idx, const float* queries, uint32_t nq, uint32_t k, nmn_metric metric,
                                       const uint64_t* mask_dev, uint64_t mask_rows, uint64_t* out_rows,
                                       float* out_scores, uint32_t* out_counts, nmn_search_stats* stats);

/* Request coalescing of the host-buffer searches (nmn_index_search / _dmask): callers that arrive while a search is
 * running on the shard wait and leave together as ONE query batch (same metric and mask, <= 64 queries, k <= 4096),
 * i.e. one corpus sweep instead of one each.  The reference serves concurrent `search_similar` calls of an
 * `Arc<VectorEngine>` (query_router/src/lib.rs:710, 5615-5666) one scan per call; results here are the same per
 * call whatever the batch.  Counters: batches that merged >= 2 calls, and the calls merged.  NMN_NO_COALESCE=1
 * turns it off. */
nmn_status nmn_index_coalesce_stats(nmn_index* idx, uint64_t* batches, uint64_t* requests);

/* ---- IVF-Flat probe ------------------------------------------------------ */

/* `tensor_store::ivf::IVFIndex` with `IVFStorage::Flat` (tensor_store/src/ivf.rs:160-406), searched on the
 * GPU.  nmn_ivf_create takes centroids trained elsewhere; nmn_ivf_build trains them (ivf.rs:222-233) on the GPU.
 * Vectors live in id (insertion) order; ids are the row numbers `add` assigns (ivf.rs:287-289). */
typedef struct nmn_ivf nmn_ivf;
/* desc: dim, capacity_rows (vectors that can be added), device; centroids: HOST, n_clusters x dim. */
nmn_status nmn_ivf_create(const nmn_index_desc* desc, const float* centroids, uint32_t n_clusters, nmn_ivf** out);
nmn_status nmn_ivf_destroy(nmn_ivf* ivf);
/* IVFIndex::train(vectors) followed by add(v) for every vector (ivf.rs:222-316), the k-means
 * (tensor_store/src/delta_vector.rs:737-901, KMeans::fit) run on the GPU bit for bit: assignment = the exact
 * centroid sweep, centroid update = one thread per (cluster, dimension) adding the members in vector order,
 * k-means++ distances on the device with the f32 running sums on the host.  rows_host: n x dim; the index gets
 * min(num_clusters, n) lists; desc->capacity_rows >= n. */
typedef struct nmn_kmeans_options {
    uint64_t max_iterations;      /* KMeansConfig::max_iterations (100)        */
    float convergence_threshold;  /* KMeansConfig::convergence_threshold (1e-4) */
    uint64_t seed;                /* KMeansConfig::seed (42)                   */
    int32_t init_method;          /* 0 = KMeansInit::Random, 1 = KMeansPlusPlus */
} nmn_kmeans_options;
nmn_status nmn_ivf_build(const nmn_index_desc* desc, const float* rows_host, uint64_t n, uint32_t num_clusters,
                         const nmn_kmeans_options* opt, nmn_ivf** out);
/* The centroids, row-major n_clusters x dim, into HOST memory. */
nmn_status nmn_ivf_centroids(nmn_ivf* ivf, float* out, uint64_t cap_floats);
/* IVFIndex::add for n vectors (HOST, n x dim): each goes to the list of its nearest centroid
 * (squared Euclidean, first minimum: find_nearest_centroid, ivf.rs:490-497) and gets the next id.
 * clusters_out (nullable, HOST [n]) receives the chosen clusters. */
nmn_status nmn_ivf_add(nmn_ivf* ivf, const float* rows_host, uint64_t n, uint32_t* clusters_out);
uint64_t nmn_ivf_len(const nmn_ivf* ivf);       /* IVFIndex::len, ivf.rs:409-415 */
uint32_t nmn_ivf_clusters(const nmn_ivf* ivf);
nmn_status nmn_ivf_cluster_sizes(nmn_ivf* ivf, uint64_t* out_sizes /* [n_clusters] */); /* ivf.rs:448-454 */
/* IVFIndex::search_with_nprobe (ivf.rs:325-406): the nprobe nearest centroids (squared distance,
 * ascending, ties by index), every vector of their lists scored by Euclidean distance
 * `squared_euclidean(q, v).sqrt()`, ascending; equal distances in probe order of the cluster, then id.
 *   queries HOST nq x dim;  out_ids nq x k (unused = UINT64_MAX);  out_distances nq x k (unused = +inf);
 *   out_counts nq = min(k, vectors in the probed lists).  Synchronous.
 * Every query is answered as if it had been searched alone (the reference's method takes one query); what nq > 1 buys is
 * throughput: up to 64 queries at a time share the centroid sweep and its host round trip, and their list scans run as one
 * batched sweep that reads a bitmap per query whenever that is cheaper than nq launch-bound scans (2M x 768, nprobe 8:
 * 0.21 ms per call at nq = 1, 0.033 ms per query at nq = 32). */
nmn_status nmn_ivf_search(nmn_ivf* ivf, const float* queries, uint32_t nq, uint32_t k, uint32_t nprobe,
                          uint64_t* out_ids, float* out_distances, uint32_t* out_counts, nmn_search_stats* stats);
/* Vectors (ids [0, n)) the LIST-MAJOR copy covers: the reference keeps a Vec of entries per list (ivf.rs:160-175), and so
 * does the device — a second copy of the vectors ordered by list, over which a probe reads contiguous row ranges and no
 * per-row array; laid out after nmn_ivf_build / nmn_ivf_load and again whenever the vectors added since make up an eighth of
 * it (those are scanned through a bitmap over the id-ordered rows meanwhile).  0: no copy (fewer than 4096 vectors, or no
 * HBM for it) — every probe goes through the bitmap.  Results are the same either way. */
uint64_t nmn_ivf_list_major_rows(const nmn_ivf* ivf);
/* The flat index holding the vectors (exhaustive search over the same rows, stats, ...). */
nmn_index* nmn_ivf_vectors(nmn_ivf* ivf);

/* ---- persistence of the device layout ----------------------------------- */

/* The reference persists a collection as PersistentVectorIndex — (key, vector, metadata) entries in JSON or bitcode —
 * and guards every load with VectorEngineConfig::max_index_file_bytes / max_index_entries (vector_engine/src/lib.rs:
 * 509-531, 644-646, 660-661, 3794-3899).  These entry points persist what the GPU path adds to that: the matrix of a
 * shard as it sits in HBM.  File: 64-byte header | rows x dim f32 (row stride removed) | rows f32 magnitudes.  A load
 * is sequential reads + bulk H2D copies; the magnitudes the upload recomputes on the GPU (reference order) must equal
 * the stored ones bit for bit, which is the file's integrity check.  The bf16 mirror is re-derived on first search.
 *   overrides   nullable: device, capacity_rows (>= the file's rows; spare room for appends), flags, cand_cap,
 *               row_base (0 = keep the file's); dim 0 or equal to the file's (else NMN_ERR_DIMENSION_MISMATCH)
 *   max_file_bytes / max_entries   0 = no limit; otherwise the reference's checks, in its order (file size before
 *               reading, entry count after the header) and with its texts: NMN_ERR_CONFIGURATION
 *               "index file size {} exceeds limit {}" / "index entry count {} exceeds limit {}". */
nmn_status nmn_index_save(nmn_index* idx, const char* path);
nmn_status nmn_index_load(const char* path, const nmn_index_desc* overrides, uint64_t max_file_bytes,
                          uint64_t max_entries, nmn_index** out);
/* The same for an IVF index: trained centroids, the list of every vector (`assign[]`) and the vectors in id order, so a
 * restart neither re-runs k-means (tensor_store/src/ivf.rs:222-233) nor re-assigns a single vector. */
nmn_status nmn_ivf_save(nmn_ivf* ivf, const char* path);
nmn_status nmn_ivf_load(const char* path, const nmn_index_desc* overrides, uint64_t max_file_bytes, uint64_t max_entries,
                        nmn_ivf** out);

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
