/*
 * neumann_engine.h — C ABI of the host-side mirror of Neumann's `VectorEngine` (the part of it that is
 * on or next to the SIMILAR TOP-K path), implemented in C++ in neumann_amd/csrc/nmn_engine.cpp on top
 * of include/neumann_gpu.h.
 *
 * Why it exists: the reference's drop-in boundary is the public Rust API of crate `vector_engine`
 *  and this image has no Rust toolchain, so the host side is written in C++ with the
 * reference's method names, argument meaning and error behaviour, and driven from Python tests that
 * read like the reference's own (vector_engine/src/lib.rs:4024+).  A Rust maintainer binds
 * neumann_gpu.h directly; this header is the test/embedding surface of the mirror.
 *
 * Every search entry point validates exactly as the reference does (EmptyVector, InvalidTopK,
 * max_dimension, collection dimension, zero-magnitude query, deadline), then runs on the GPU mirror of
 * the affected collection.  The mirror is a derived cache with the lifecycle of the reference's
 * `hnsw_cache` (lib.rs:98,1156,1311-1328): built lazily on first search, dropped by every
 * store/delete in that collection (lib.rs:1497,1532,1866,1923).
 *
 * Status codes are the nmn_status values of neumann_gpu.h; `nmn_engine_last_error()` returns the
 * reference's `Display` text of the VectorError (e.g. "Dimension mismatch: expected 3, got 2").
 */
#ifndef NEUMANN_ENGINE_H
#define NEUMANN_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#include "neumann_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nmn_engine nmn_engine;
typedef struct nmn_results nmn_results; /* Vec<SearchResult> */
typedef struct nmn_filter nmn_filter;   /* FilterCondition (lib.rs:296-324) */
typedef struct nmn_strlist nmn_strlist; /* Vec<String> */

/* VectorEngineConfig (lib.rs:626-664); 0 / negative = None where the reference has Option. */
typedef struct nmn_engine_config {
    uint64_t default_dimension;  /* 0 = None */
    float sparse_threshold;      /* 0.5 (storage format only; scoring is unaffected) */
    uint64_t parallel_threshold; /* 5000: kept for API parity, the GPU path has no sequential mode */
    int32_t default_metric;      /* nmn_metric */
    uint64_t max_dimension;      /* 0 = None */
    uint64_t max_keys_per_scan;  /* 0 = None; else list_keys / list_keys_paginated / clear / search_entities /
                                    scan_entities_with_embeddings stop after that many keys of the scan (lib.rs:2322, 2341,
                                    2948, 3179, 3225).  Some(0) is a ConfigurationError in the reference (lib.rs:728-733): a
                                    binding rejects it before it fills this struct */
    int64_t search_timeout_ms;   /* <0 = None */
    int32_t device;              /* GPU ordinal, -1 = current (new knob, additive) */
    uint32_t cand_cap;           /* 0 = default (new knob, additive) */
    int64_t max_index_file_bytes; /* lib.rs:644, 660: 100 MiB; < 0 = None; 0 is a ConfigurationError (lib.rs:740-746) */
    int64_t max_index_entries;    /* lib.rs:646, 661: 1 000 000; < 0 = None; 0 is a ConfigurationError (lib.rs:747-753) */
    /* GPUs of this node the engine spreads every collection over (new knob, additive).  0: `device` alone.  1: devices[0]
     * alone.  >= 2: each mirror is ONE nmn_sharded index (include/neumann_gpu.h, NMN_SHARDED_LAYOUT_CYCLIC: 64-row blocks dealt
     * round-robin, so the rows held — not the capacity — are spread evenly and stay so under appends) whose shards sit on
     * devices[0..n_devices): a search runs on all of them at once, the per-GPU top-k blocks are gathered (RCCL all-gather
     * over xGMI, or peer copies when an ordinal repeats) and merged on devices[0] with ResultMerger::merge_top_k's rule
     * (query_router/src/distributed.rs:413-433).  Results are the single-GPU results.  Metadata columns, IVF indexes and
     * the compute_similarity slot stay on devices[0]; a predicate's bitmap is evaluated there and sliced per shard. */
    uint32_t n_devices;
    int32_t devices[16]; /* NMN_ENGINE_MAX_DEVICES */
} nmn_engine_config;
#define NMN_ENGINE_MAX_DEVICES 16u

/* ScalarValue / FilterValue payload (tensor_store ScalarValue; lib.rs:342-353). */
#define NMN_VAL_NULL 0
#define NMN_VAL_BOOL 1
#define NMN_VAL_INT 2
#define NMN_VAL_FLOAT 3
#define NMN_VAL_STRING 4
typedef struct nmn_value {
    int32_t kind;
    int32_t b;
    int64_t i;
    double f;
    const char* s;
} nmn_value;
typedef struct nmn_meta_field {
    const char* name;
    nmn_value value;
} nmn_meta_field;

/* FilterStrategy (lib.rs:386-397) / FilteredSearchConfig (lib.rs:399-449). */
#define NMN_FILTER_AUTO 0
#define NMN_FILTER_PRE 1
#define NMN_FILTER_POST 2
typedef struct nmn_filtered_config {
    int32_t strategy;
    float selectivity_threshold; /* 0.1 */
    uint64_t oversample_factor;  /* 3 */
} nmn_filtered_config;

void nmn_engine_config_default(nmn_engine_config* c);
void nmn_filtered_config_default(nmn_filtered_config* c);

/* VectorEngine::new / with_config (lib.rs:1162-1203); config NULL = default. */
nmn_status nmn_engine_create(const nmn_engine_config* config, nmn_engine** out);
void nmn_engine_destroy(nmn_engine* e);
const char* nmn_engine_last_error(void);

/* store_embedding / store_embedding_with_metadata (lib.rs:1840-1868, 3272-3310) */
nmn_status nmn_engine_store_embedding(nmn_engine* e, const char* key, const float* v, uint64_t dim);
nmn_status nmn_engine_store_embedding_with_metadata(nmn_engine* e, const char* key, const float* v, uint64_t dim,
                                                    const nmn_meta_field* meta, uint32_t n_meta);
/* get_embedding (lib.rs:1895-1908): *dim_out receives the length; copies min(cap, len) floats. */
nmn_status nmn_engine_get_embedding(nmn_engine* e, const char* key, float* out, uint64_t cap, uint64_t* dim_out);
nmn_status nmn_engine_delete_embedding(nmn_engine* e, const char* key); /* lib.rs:1915-1925 */
int32_t nmn_engine_exists(nmn_engine* e, const char* key);              /* lib.rs:1929-1932 */
uint64_t nmn_engine_count(nmn_engine* e);                               /* lib.rs:1936-1938 */
nmn_strlist* nmn_engine_list_keys(nmn_engine* e);                       /* lib.rs:2312-2329 (bounded by max_keys_per_scan) */
/* list_keys_paginated (lib.rs:2945-2980): limit -1 = None; *total_count -1 = None */
nmn_strlist* nmn_engine_list_keys_paginated(nmn_engine* e, uint64_t skip, int64_t limit, int32_t count_total,
                                            int64_t* total_count, int32_t* has_more);
/* clear (lib.rs:2340-2354): with max_keys_per_scan set and more keys stored, deletes that many ("call again until 0") */
nmn_status nmn_engine_clear(nmn_engine* e, uint64_t* removed);
nmn_status nmn_engine_batch_store(nmn_engine* e, const char* const* keys, const float* rows, uint64_t n,
                                  uint64_t dim); /* batch_store_embeddings (lib.rs:2865-2913), uniform dim */

/* search_similar / search_similar_with_metric (lib.rs:1950-2101) */
nmn_status nmn_engine_search_similar(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k,
                                     nmn_results** out);
nmn_status nmn_engine_search_similar_with_metric(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k,
                                                 int32_t metric, nmn_results** out);
/* search_similar_filtered (lib.rs:3429-3477); config NULL = default */
nmn_status nmn_engine_search_similar_filtered(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k,
                                              const nmn_filter* filter, const nmn_filtered_config* config,
                                              nmn_results** out);
/* compute_similarity (lib.rs:2278-2295) — scored by the exact GPU kernel */
nmn_status nmn_engine_compute_similarity(nmn_engine* e, const float* a, uint64_t na, const float* b, uint64_t nb,
                                         float* out);

/* collections (lib.rs:1369-1582) */
nmn_status nmn_engine_create_collection(nmn_engine* e, const char* name, uint64_t dimension /*0=None*/,
                                        int32_t metric);
nmn_status nmn_engine_delete_collection(nmn_engine* e, const char* name);
int32_t nmn_engine_collection_exists(nmn_engine* e, const char* name);
uint64_t nmn_engine_collection_count(nmn_engine* e, const char* name);
nmn_strlist* nmn_engine_list_collections(nmn_engine* e);
nmn_status nmn_engine_store_in_collection(nmn_engine* e, const char* coll, const char* key, const float* v,
                                          uint64_t dim, const nmn_meta_field* meta, uint32_t n_meta);
nmn_status nmn_engine_get_from_collection(nmn_engine* e, const char* coll, const char* key, float* out,
                                          uint64_t cap, uint64_t* dim_out);
nmn_status nmn_engine_delete_from_collection(nmn_engine* e, const char* coll, const char* key);
/* search_in_collection / search_filtered_in_collection (lib.rs:1585-1829) */
nmn_status nmn_engine_search_in_collection(nmn_engine* e, const char* coll, const float* q, uint64_t dim,
                                           uint64_t top_k, nmn_results** out);
nmn_status nmn_engine_search_filtered_in_collection(nmn_engine* e, const char* coll, const float* q, uint64_t dim,
                                                    uint64_t top_k, const nmn_filter* filter,
                                                    const nmn_filtered_config* config, nmn_results** out);

/* Measurement aid (bench.py's `published_shapes` leg: the shapes of vector_engine/benches/vector_engine_bench.rs:40-77): `calls`
 * back-to-back nmn_engine_search_similar calls from ONE host thread — call i with query i % n_queries of `queries` (HOST,
 * n_queries x dim), results taken and freed as a caller would — each timed on the host's steady clock around the whole call
 * (validation, H2D of the query, kernels, D2H, key strings): out_us[i] = microseconds of call i.  No Python in the loop. */
nmn_status nmn_engine_search_probe(nmn_engine* e, const float* queries, uint64_t n_queries, uint64_t dim, uint64_t top_k,
                                   uint64_t calls, float* out_us);

/* results */
uint64_t nmn_results_len(const nmn_results* r);
const char* nmn_results_key(const nmn_results* r, uint64_t i);
float nmn_results_score(const nmn_results* r, uint64_t i);
void nmn_results_free(nmn_results* r);
uint64_t nmn_strlist_len(const nmn_strlist* l);
const char* nmn_strlist_get(const nmn_strlist* l, uint64_t i);
void nmn_strlist_free(nmn_strlist* l);

/* FilterCondition builders (lib.rs:296-340); ops for nmn_filter_cmp */
#define NMN_OP_EQ 0
#define NMN_OP_NE 1
#define NMN_OP_LT 2
#define NMN_OP_LE 3
#define NMN_OP_GT 4
#define NMN_OP_GE 5
nmn_filter* nmn_filter_cmp(int32_t op, const char* field, const nmn_value* value);
nmn_filter* nmn_filter_and(nmn_filter* a, nmn_filter* b); /* takes ownership of a and b */
nmn_filter* nmn_filter_or(nmn_filter* a, nmn_filter* b);
nmn_filter* nmn_filter_true(void);
nmn_filter* nmn_filter_exists(const char* field);
nmn_filter* nmn_filter_contains(const char* field, const char* substr);
nmn_filter* nmn_filter_starts_with(const char* field, const char* prefix);
nmn_filter* nmn_filter_in(const char* field, const nmn_value* values, uint32_t n);
void nmn_filter_free(nmn_filter* f);
/* ---- metadata CRUD (lib.rs:3311-3385) and the remaining helpers of the filtered / paginated surface ---- */
typedef struct nmn_metalist nmn_metalist;  /* HashMap<String, TensorValue> */
nmn_metalist* nmn_engine_get_metadata(nmn_engine* e, const char* key, nmn_status* status);      /* 3312-3327 */
uint64_t nmn_metalist_len(const nmn_metalist* l);
const char* nmn_metalist_name(const nmn_metalist* l, uint64_t i);
nmn_status nmn_metalist_value(const nmn_metalist* l, uint64_t i, nmn_value* out);
void nmn_metalist_free(nmn_metalist* l);
nmn_status nmn_engine_update_metadata(nmn_engine* e, const char* key, const nmn_meta_field* meta, uint32_t n_meta); /* 3329-3351 */
nmn_status nmn_engine_remove_metadata_field(nmn_engine* e, const char* key, const char* field);                    /* 3353-3364 */
int32_t nmn_engine_has_metadata_field(nmn_engine* e, const char* key, const char* field);                          /* 3366-3372 */
nmn_status nmn_engine_get_metadata_field(nmn_engine* e, const char* key, const char* field, nmn_value* out,
                                         int32_t* present);                                                         /* 3374-3383 */
nmn_status nmn_engine_estimate_filter_selectivity(nmn_engine* e, const nmn_filter* f, float* out);                  /* 3695-3711 */
nmn_strlist* nmn_engine_list_keys_matching(nmn_engine* e, const nmn_filter* f);                                     /* 3720-3725 */
nmn_status nmn_engine_batch_delete(nmn_engine* e, const char* const* keys, uint64_t n, uint64_t* deleted);         /* 2924-2940 */
uint64_t nmn_engine_dimension(nmn_engine* e);                                            /* 2298-2308; 0 = None */
int32_t nmn_engine_exists_in_collection(nmn_engine* e, const char* coll, const char* key);                          /* 1537-1540 */
nmn_strlist* nmn_engine_list_collection_keys(nmn_engine* e, const char* coll);                                      /* 1543-1550 */
/* Pagination{skip, limit (-1 = None), count_total} -> PagedResult{items, total_count (-1 = None), has_more} (lib.rs:1052-1112) */
nmn_status nmn_engine_search_similar_paginated(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, uint64_t skip,
                                               int64_t limit, int32_t count_total, nmn_results** out, int64_t* total_count,
                                               int32_t* has_more);                                                  /* 2988-3019 */
nmn_status nmn_engine_search_entities_paginated(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, uint64_t skip,
                                                int64_t limit, int32_t count_total, nmn_results** out, int64_t* total_count,
                                                int32_t* has_more);                                                 /* 3027-3058 */

/* ---- tensor_blob artifact similarity (tensor_blob/src/lib.rs:520-625) ----------------------------- */
/* set_embedding (529-556) for the artifact record `_blob:meta:{artifact_id}`; filename = its `_filename`. */
nmn_status nmn_engine_blob_set_embedding(nmn_engine* e, const char* artifact_id, const char* filename, const float* v,
                                         uint64_t dim);
nmn_status nmn_engine_blob_remove(nmn_engine* e, const char* artifact_id);
/* search_by_embedding (591-625): SimilarArtifact{id = key, filename = aux, similarity = score}, f64 sparse cosine */
nmn_status nmn_engine_blob_search_by_embedding(nmn_engine* e, const float* q, uint64_t dim, uint64_t k, nmn_results** out);
/* similar (563-583) */
nmn_status nmn_engine_blob_similar(nmn_engine* e, const char* artifact_id, uint64_t k, nmn_results** out);
const char* nmn_results_aux(const nmn_results* r, uint64_t i);

/* ---- IVF-Flat (lib.rs:2641-2812; tensor_store/src/ivf.rs) ---------------------------------------- */
#define NMN_KMEANS_INIT_RANDOM 0   /* KMeansInit::Random          (delta_vector.rs:781-800) */
#define NMN_KMEANS_INIT_PLUSPLUS 1 /* KMeansInit::KMeansPlusPlus  (delta_vector.rs:805-853) */
typedef struct nmn_ivf_options {   /* IVFBuildOptions / IVFConfig (Flat storage) + KMeansConfig */
    uint64_t num_clusters;         /* 100 */
    uint64_t nprobe;               /* 0 = default_nprobe(num_clusters) = ceil(sqrt(num_clusters)) */
    uint64_t max_iterations;       /* 100 */
    float convergence_threshold;   /* 1e-4 */
    uint64_t seed;                 /* 42 */
    int32_t init_method;           /* NMN_KMEANS_INIT_PLUSPLUS */
} nmn_ivf_options;
typedef struct nmn_engine_ivf nmn_engine_ivf; /* (IVFIndex, Vec<String> key_mapping) */
void nmn_ivf_options_default(nmn_ivf_options* o);
/* build_ivf_index (lib.rs:2641-2694): k-means (bit for bit) and list assignment on the GPU, see nmn_ivf_build */
nmn_status nmn_engine_build_ivf_index(nmn_engine* e, const nmn_ivf_options* options, nmn_engine_ivf** out);
void nmn_engine_ivf_free(nmn_engine_ivf* ivf);
uint64_t nmn_engine_ivf_len(const nmn_engine_ivf* ivf);
uint32_t nmn_engine_ivf_clusters(const nmn_engine_ivf* ivf);
uint64_t nmn_engine_ivf_nprobe(const nmn_engine_ivf* ivf);
const char* nmn_engine_ivf_key(const nmn_engine_ivf* ivf, uint64_t id);
nmn_status nmn_engine_ivf_centroids(const nmn_engine_ivf* ivf, float* out, uint64_t cap_floats);
nmn_status nmn_engine_ivf_cluster_sizes(nmn_engine_ivf* ivf, uint64_t* out);
/* search_with_ivf (nprobe = 0) / search_with_ivf_nprobe (lib.rs:2731-2812): score = 1 / (1 + distance) */
nmn_status nmn_engine_search_with_ivf(nmn_engine* e, nmn_engine_ivf* ivf, const float* q, uint64_t dim, uint64_t top_k,
                                      uint64_t nprobe, nmn_results** out);

/* ---- unified entity mode (lib.rs:3060-3237): vectors in the `_embedding` field of entity keys ---- */
nmn_status nmn_engine_set_entity_embedding(nmn_engine* e, const char* entity_key, const float* v, uint64_t dim);
nmn_status nmn_engine_get_entity_embedding(nmn_engine* e, const char* entity_key, float* out, uint64_t cap,
                                           uint64_t* dim_out);
int32_t nmn_engine_entity_has_embedding(nmn_engine* e, const char* entity_key);
nmn_status nmn_engine_remove_entity_embedding(nmn_engine* e, const char* entity_key);
nmn_strlist* nmn_engine_scan_entities_with_embeddings(nmn_engine* e);
uint64_t nmn_engine_count_entities_with_embeddings(nmn_engine* e);
/* search_entities (lib.rs:3155-3219): cosine TOP-K over every entity that has an embedding */
nmn_status nmn_engine_search_entities(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, nmn_results** out);

/* ---- index persistence (lib.rs:500-623, 3733-4000) ------------------------------------------------- */
/* `collection`: VectorEngine::DEFAULT_COLLECTION = "default" for the non-collection embeddings.
 * save_index / load_index: PersistentVectorIndex as serde_json writes it (either side reads the other's file).
 * save_index_binary / load_index_binary: the reference's second format is bitcode (an absent third-party crate's
 * bit-packed encoding); the binary format here is this library's own — the same snapshot with the vectors as flat
 * shard sections (include/neumann_gpu.h "persistence of the device layout"): a load is bulk reads + H2D copies, the
 * GPU mirror is rebuilt at once and every row's device-computed magnitude is checked against the stored one.
 * Loads apply max_index_file_bytes before reading and max_index_entries after decoding (ConfigurationError,
 * "index file size {} exceeds limit {}" / "index entry count {} exceeds limit {}").  name_out (nullable) receives the
 * collection's name, NUL-terminated, truncated to name_cap - 1 bytes. */
nmn_status nmn_engine_save_index(nmn_engine* e, const char* collection, const char* path);                 /* 3794-3801 */
nmn_status nmn_engine_load_index(nmn_engine* e, const char* path, char* name_out, uint64_t name_cap);      /* 3827-3866 */
nmn_status nmn_engine_save_index_binary(nmn_engine* e, const char* collection, const char* path);          /* 3811-3817 */
nmn_status nmn_engine_load_index_binary(nmn_engine* e, const char* path, char* name_out, uint64_t name_cap); /* 3868-3899 */
nmn_strlist* nmn_engine_save_all_indices(nmn_engine* e, const char* dir, nmn_status* status);              /* 3944-3971 */
nmn_strlist* nmn_engine_load_all_indices(nmn_engine* e, const char* dir, nmn_status* status);              /* 3980-3999 */
/* The (IVFIndex, key_mapping) pair of build_ivf_index with its trained centroids and lists: a restart restores it
 * without k-means (the reference rebuilds: lib.rs:2641-2694). */
nmn_status nmn_engine_ivf_save(nmn_engine_ivf* ivf, const char* path);
nmn_status nmn_engine_ivf_load(nmn_engine* e, const char* path, nmn_engine_ivf** out);

/* count_matching / estimate_filter_selectivity (lib.rs:3698-3722) */
uint64_t nmn_engine_count_matching(nmn_engine* e, const nmn_filter* f);

/* Mirror bookkeeping (for the cache-protocol tests, lib.rs:9686-9944): number of GPU mirror builds
 * so far and whether a mirror is currently cached for `coll` (NULL = default collection). */
uint64_t nmn_engine_mirror_builds(nmn_engine* e);
/* rows the GPU mirror of (default collection, dim) holds on each GPU: out[0 .. min(cap, shards)); returns the shard count */
uint32_t nmn_engine_mirror_shard_rows(nmn_engine* e, uint64_t dim, uint64_t* out, uint32_t cap);
/* device memory of that mirror, summed over its shards: out[0] = the f32 rows, out[1] = the 8-bit / bf16 mirrors that exist right
 * now (which ones a shard builds: nmn_index_set_mirror in neumann_gpu.h), out[2] = per-row factors; returns the shard count */
uint32_t nmn_engine_mirror_hbm_bytes(nmn_engine* e, uint64_t dim, uint64_t out[3]);
/* Pre-filter predicates evaluated by the GPU predicate kernel over the metadata columns, and how many
 * times a column set was (re)built from the store (instrumentation of SURVEY.md §8f-2). */
uint64_t nmn_engine_device_filter_evals(nmn_engine* e);
uint64_t nmn_engine_column_builds(nmn_engine* e);
int32_t nmn_engine_mirror_cached(nmn_engine* e, const char* coll);

#ifdef __cplusplus
}
#endif
#endif /* NEUMANN_ENGINE_H */
