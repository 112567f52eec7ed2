// nmn_engine.cpp — host-side mirror of Neumann's `VectorEngine` for the SIMILAR TOP-K path
// (C ABI: include/neumann_engine.h).  C++ because the reference is compiled code and the image has no
// Rust toolchain; names, argument meaning, validation order and error texts follow
// vector_engine/src/lib.rs (cited per function).  ALL scoring happens on the GPU through
// include/neumann_gpu.h; this file holds the key/value bookkeeping the reference keeps in
// `TensorStore`, the validation rules, the metadata predicate evaluator and the GPU-mirror cache.
//
// Built with -ffp-contract=off: the zero-magnitude-query rule needs `simd::magnitude(query)` in
// reference order on the host (validation, not the hot path).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>
#include <hip/hip_runtime_api.h>  // hipMemcpy of a selection bitmap / the magnitudes (no kernels in this file)

#include "../../include/neumann_engine.h"

namespace {

thread_local std::string g_err;

nmn_status fail(nmn_status code, std::string msg) {
    g_err = std::move(msg);
    return code;
}
// Display texts of VectorError (lib.rs:151-183)
nmn_status err_not_found(const std::string& key) { return fail(NMN_ERR_NOT_FOUND, "Embedding not found: " + key); }
nmn_status err_dim(uint64_t expected, uint64_t got) {
    return fail(NMN_ERR_DIMENSION_MISMATCH,
                "Dimension mismatch: expected " + std::to_string(expected) + ", got " + std::to_string(got));
}
nmn_status err_empty() { return fail(NMN_ERR_EMPTY_VECTOR, "Empty vector provided"); }
nmn_status err_topk() { return fail(NMN_ERR_INVALID_TOP_K, "Invalid top_k value (must be > 0)"); }
nmn_status err_timeout(const char* op, int64_t ms) {
    return fail(NMN_ERR_SEARCH_TIMEOUT, std::string("search timeout: ") + op + " exceeded " + std::to_string(ms) + "ms");
}
nmn_status err_gpu(nmn_status st) {
    // StorageError(String) carries the shim's message (lib.rs:185-189: From<TensorStoreError>)
    const char* detail = nmn_last_error();
    return fail(st, std::string("Storage error: ") + nmn_status_str(st) + (detail && *detail ? std::string(": ") + detail : ""));
}

// ---- ScalarValue / FilterValue -------------------------------------------------------------------
struct Value {
    int kind = NMN_VAL_NULL;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    static Value from(const nmn_value& v) {
        Value o;
        o.kind = v.kind;
        o.b = v.b != 0;
        o.i = v.i;
        o.f = v.f;
        if (v.kind == NMN_VAL_STRING && v.s) o.s = v.s;
        return o;
    }
};

// compare_tensor_value_to_filter (lib.rs:3648-3670): ordering if the types are compatible.
// returns false when incomparable; *ord = -1/0/+1 otherwise.
bool compare_values(const Value& stored, const Value& flt, int* ord) {
    auto cmp3 = [](auto a, auto b) { return a < b ? -1 : (a > b ? 1 : 0); };
    if (stored.kind == NMN_VAL_INT && flt.kind == NMN_VAL_INT) { *ord = cmp3(stored.i, flt.i); return true; }
    if (stored.kind == NMN_VAL_FLOAT && flt.kind == NMN_VAL_FLOAT) {
        if (std::isnan(stored.f) || std::isnan(flt.f)) return false;  // partial_cmp -> None
        *ord = cmp3(stored.f, flt.f); return true;
    }
    if (stored.kind == NMN_VAL_FLOAT && flt.kind == NMN_VAL_INT) {
        if (std::isnan(stored.f)) return false;
        *ord = cmp3(stored.f, (double)flt.i); return true;
    }
    if (stored.kind == NMN_VAL_INT && flt.kind == NMN_VAL_FLOAT) {
        if (std::isnan(flt.f)) return false;
        *ord = cmp3((double)stored.i, flt.f); return true;
    }
    if (stored.kind == NMN_VAL_STRING && flt.kind == NMN_VAL_STRING) { *ord = cmp3(stored.s.compare(flt.s), 0); return true; }
    if (stored.kind == NMN_VAL_BOOL && flt.kind == NMN_VAL_BOOL) { *ord = cmp3((int)stored.b, (int)flt.b); return true; }
    if (stored.kind == NMN_VAL_NULL && flt.kind == NMN_VAL_NULL) { *ord = 0; return true; }
    return false;
}

}  // namespace

// FilterCondition (lib.rs:296-324)
struct nmn_filter {
    enum Kind { Cmp, And, Or, True, Exists, Contains, StartsWith, In } kind = True;
    int op = NMN_OP_EQ;
    std::string field, text;
    Value value;
    std::vector<Value> values;
    std::unique_ptr<nmn_filter> a, b;
};

struct nmn_results {
    std::vector<std::string> keys;
    std::vector<float> scores;
    std::vector<std::string> aux;  // SimilarArtifact::filename for the artifact searches, otherwise empty
};
struct nmn_strlist {
    std::vector<std::string> items;
};
struct nmn_metalist {  // HashMap<String, TensorValue> of get_metadata (lib.rs:3312-3327)
    std::vector<std::string> names;
    std::vector<Value> values;
};
// (IVFIndex, Vec<String>) as build_ivf_index returns it (lib.rs:2641-2694): the GPU index + id -> key mapping
struct nmn_engine_ivf {
    nmn_ivf* index = nullptr;           // null = untrained (built from an empty store)
    std::vector<std::string> keys;      // key_mapping: id -> key
    std::vector<float> centroids;
    uint32_t n_clusters = 0;
    uint64_t dim = 0, nprobe = 0;
    ~nmn_engine_ivf() {
        if (index) nmn_ivf_destroy(index);
    }
};

namespace {

using Meta = std::map<std::string, Value>;

// evaluate_filter (lib.rs:3592-3630)
bool evaluate_filter(const Meta& meta, const nmn_filter& f) {
    switch (f.kind) {
        case nmn_filter::True: return true;
        case nmn_filter::And: return evaluate_filter(meta, *f.a) && evaluate_filter(meta, *f.b);
        case nmn_filter::Or: return evaluate_filter(meta, *f.a) || evaluate_filter(meta, *f.b);
        case nmn_filter::Exists: return meta.count(f.field) != 0;
        case nmn_filter::Cmp: {
            auto it = meta.find(f.field);
            if (it == meta.end()) return false;
            int ord = 0;
            if (!compare_values(it->second, f.value, &ord)) return false;
            switch (f.op) {
                case NMN_OP_EQ: return ord == 0;
                case NMN_OP_NE: return ord != 0;
                case NMN_OP_LT: return ord < 0;
                case NMN_OP_LE: return ord <= 0;
                case NMN_OP_GT: return ord > 0;
                default: return ord >= 0;
            }
        }
        case nmn_filter::Contains: {
            auto it = meta.find(f.field);
            return it != meta.end() && it->second.kind == NMN_VAL_STRING && it->second.s.find(f.text) != std::string::npos;
        }
        case nmn_filter::StartsWith: {
            auto it = meta.find(f.field);
            return it != meta.end() && it->second.kind == NMN_VAL_STRING && it->second.s.compare(0, f.text.size(), f.text) == 0;
        }
        case nmn_filter::In: {
            auto it = meta.find(f.field);
            if (it == meta.end()) return false;
            for (const auto& v : f.values) {
                int ord = 0;
                if (compare_values(it->second, v, &ord) && ord == 0) return true;
            }
            return false;
        }
    }
    return false;
}

// simd::sum_of_squares in reference order (hnsw.rs:198-222) — host-side, validation only.
float sumsq8_host(const float* v, uint64_t n) {
    const uint64_t chunks = n / 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint64_t c = 0; c < chunks; c++)
        for (int l = 0; l < 8; l++) {
            const float p = v[8 * c + l] * v[8 * c + l];
            acc[l] = acc[l] + p;
        }
    float r = -0.0f;
    for (int l = 0; l < 8; l++) r = r + acc[l];
    for (uint64_t i = chunks * 8; i < n; i++) {
        const float p = v[i] * v[i];
        r = r + p;
    }
    return r;
}
bool zero_magnitude(const float* q, uint64_t n) { return std::sqrt(sumsq8_host(q, n)) == 0.0f; }

struct Entry {
    std::string key;
    std::vector<float> vec;
    Meta meta;
    bool live = false;
    int64_t mrow = -1;  // row of this entry in the GPU mirror of its dimension (-1: not mirrored yet)
};

// GPU mirror of the rows of one collection that have one dimension: the `hnsw_cache` slot of the
// reference (lib.rs:98,1311-1328) with a flat GPU index instead of an HNSW graph.
//
// Maintenance: the reference drops its cache on every store/delete
// (lib.rs:1497,1532,1866,1923) because an HNSW graph cannot be patched cheaply; a flat matrix can.
// New keys are APPENDED into spare capacity (one 3 KB H2D copy + one norm), overwrites rewrite their
// row in place, deletes clear the row's bit in a `live` bitmap that every search passes as (part of)
// the predicate mask.  The mirror is rebuilt only when the spare capacity is exhausted or more than a
// quarter of its rows are dead.  Searches always see exactly the store's current contents.
//
// Metadata: on the first pre-filtered search the metadata fields of the mirrored rows
// are laid out as typed columns in HBM (`cols`, one (kind, payload) cell per row and field; strings
// dictionary-encoded per field) and kept current by the same append / overwrite / tombstone hooks, so
// the predicate of every later filtered search is one kernel over the columns instead of a
// `store.get` + `evaluate_filter` per key (lib.rs:3526-3530).
struct FieldColumn {
    uint32_t id = 0;                                  // column id inside `cols`
    std::unordered_map<std::string, uint32_t> dict;   // string value -> dictionary id
    std::vector<std::string> strings;                 // dictionary id -> string value
};

struct Mirror {
    nmn_index* idx = nullptr;    // one GPU (nmn_engine_config.n_devices <= 1)
    nmn_sharded* sh = nullptr;   // several: the rows split into equal ranges over config.devices[] (include/neumann_gpu.h)
    bool has_rows() const { return idx || sh; }
    uint64_t rows() const { return idx ? nmn_index_rows(idx) : sh ? nmn_sharded_rows(sh) : 0; }
    nmn_status upload(const float* rows_host, uint64_t row0, uint64_t n) {
        return idx ? nmn_index_upload(idx, rows_host, row0, n) : nmn_sharded_upload(sh, rows_host, row0, n);
    }
    std::vector<uint32_t> row_to_slot;
    std::vector<uint64_t> live;  // bit r of word r/64: row r takes part
    uint64_t cap = 0, n_dead = 0;
    int32_t device = -1;
    // Rows whose vector changed on the host since the last upload (stores of existing keys, appended keys).  A store
    // only records the row; the next search of this mirror uploads all of them in runs of consecutive rows (one store =
    // one H2D + two kernels + two stream waits, ~25 us: 45k stores/s against 350k without a mirror).  Rows beyond
    // nmn_index_rows() are always dirty (appends are contiguous).
    std::vector<uint64_t> dirty;
    nmn_columns* cols = nullptr;  // null until a filtered search needs it (or after a device error)
    std::unordered_map<std::string, FieldColumn> fields;
    void drop_columns() {
        if (cols) nmn_columns_destroy(cols);
        cols = nullptr;
        fields.clear();
    }
    ~Mirror() {
        drop_columns();
        if (idx) nmn_index_destroy(idx);
        if (sh) nmn_sharded_destroy(sh);
    }
};

struct Collection {
    std::vector<Entry> slots;
    std::unordered_map<std::string, uint32_t> by_key;
    std::vector<uint32_t> free_slots;
    uint64_t live = 0;
    std::unordered_map<uint64_t, std::unique_ptr<Mirror>> mirrors;  // by dimension
    bool has_dirty = false;                                          // some mirror here has stores pending
    void invalidate() { mirrors.clear(); }  // full drop (delete_collection / clear)
};

struct CollectionConfig {  // VectorCollectionConfig (lib.rs:455-475)
    uint64_t dimension = 0;  // 0 = None
    int32_t metric = NMN_METRIC_COSINE;
    bool auto_index = false;               // HNSW auto-indexing: carried through index files, not acted on (CPU structure)
    uint64_t auto_index_threshold = 1000;
};

struct Deadline {  // lib.rs:216-249
    bool has = false;
    std::chrono::steady_clock::time_point at;
    int64_t ms = 0;
    explicit Deadline(int64_t timeout_ms) {
        if (timeout_ms >= 0) {
            has = true;
            ms = timeout_ms;
            at = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
        }
    }
    bool expired() const { return has && std::chrono::steady_clock::now() >= at; }
};

}  // namespace

struct nmn_engine {
    nmn_engine_config cfg;
    // `&self` from many threads is safe.  Searches whose GPU mirror already exists run under the SHARED lock (the shim
    // merges them into query batches); everything that writes the store or builds a mirror / its columns takes the
    // exclusive lock.  The lock is phase-fair: a waiting writer stops NEW readers (glibc's rwlock prefers readers: a
    // steady stream of searches would starve every store), and when it is done the readers that waited go first, all
    // together (plain writer preference starved the searches under a steady stream of stores: measured 3 q/s).  One
    // mutex + two condition variables: a spin-and-yield ticket turnstile in front of a shared_mutex made 128 waiting
    // searches steal the CPU from the store they waited for (14 ms per store), and a hand-over queue serialised the
    // readers' entry (64 wake-ups in a chain per cohort).
    struct PhaseFairLock {
        std::mutex m;
        std::condition_variable readers_cv, writers_cv;
        int active_readers = 0, waiting_readers = 0, waiting_writers = 0;
        bool writer_active = false, readers_turn = false;
        void read_lock() {
            std::unique_lock<std::mutex> lk(m);
            if (writer_active || (waiting_writers > 0 && !readers_turn)) {
                waiting_readers++;
                readers_cv.wait(lk, [&] { return !writer_active && (waiting_writers == 0 || readers_turn); });
                if (--waiting_readers == 0) readers_turn = false;  // the cohort is in: the next writer's turn
            }
            active_readers++;
            reader_entries++;
        }
        void read_unlock() {
            std::lock_guard<std::mutex> g(m);
            if (--active_readers == 0 && waiting_writers > 0) writers_cv.notify_one();
        }
        void write_lock() {
            std::unique_lock<std::mutex> lk(m);
            waiting_writers++;
            writers_cv.wait(lk, [&] { return !writer_active && active_readers == 0 && !(readers_turn && waiting_readers > 0); });
            waiting_writers--;
            writer_active = true;
        }
        uint64_t reader_entries = 0;  // searches that entered so far
        // searches are waiting right now, or some ran since the caller last looked (*seen is updated)
        bool readers_around(uint64_t* seen) {
            std::lock_guard<std::mutex> g(m);
            const bool around = waiting_readers > 0 || reader_entries != *seen;
            *seen = reader_entries;
            return around;
        }
        void write_unlock() {
            std::lock_guard<std::mutex> g(m);
            writer_active = false;
            if (waiting_readers > 0) {
                readers_turn = true;
                readers_cv.notify_all();
            } else if (waiting_writers > 0) {
                writers_cv.notify_one();
            }
        }
    } rw;
    // A writer that finds searches waiting for the lock — or that searches ran since the previous write — uploads the
    // pending stores (Mirror::dirty, Collection::has_dirty) before it leaves: it already holds the exclusive lock the
    // upload needs.  With no search around they stay pending and a burst of stores becomes one upload.
    uint64_t reader_entries_seen = 0;
    Collection dflt;
    Collection entities;                              // unified entity mode: keys whose TensorData has `_embedding`
    Collection artifacts;                             // tensor_blob: `_blob:meta:{id}` records that carry `_embedding`
    std::map<std::string, Collection> colls;          // storage of named collections
    std::map<std::string, CollectionConfig> configs;  // `collections` map (configured ones only)
    uint64_t mirror_builds = 0;
    uint64_t column_builds = 0;    // metadata column sets built from the store
    std::atomic<uint64_t> device_filters{0};   // predicates evaluated by the GPU kernel (bumped under the shared lock too)
    std::unordered_map<uint64_t, std::unique_ptr<Mirror>> scratch;  // compute_similarity, by dim

    Collection* storage(const char* coll, bool create) {
        if (!coll) return &dflt;
        auto it = colls.find(coll);
        if (it != colls.end()) return &it->second;
        if (!create) return nullptr;
        return &colls[coll];
    }
};

namespace {

// exclusive / shared lock of the engine (nmn_engine::PhaseFairLock)
void flush_dirty_mirrors(nmn_engine* e);
struct WriteLock {
    nmn_engine* e;
    explicit WriteLock(nmn_engine* e_) : e(e_) { e->rw.write_lock(); }
    ~WriteLock() {
        if (e->rw.readers_around(&e->reader_entries_seen)) flush_dirty_mirrors(e);
        e->rw.write_unlock();
    }
    WriteLock(const WriteLock&) = delete;
    WriteLock& operator=(const WriteLock&) = delete;
};
struct ReadLock {
    nmn_engine* e;
    explicit ReadLock(nmn_engine* e_) : e(e_) { e->rw.read_lock(); }
    ~ReadLock() { e->rw.read_unlock(); }
    ReadLock(const ReadLock&) = delete;
    ReadLock& operator=(const ReadLock&) = delete;
};

Mirror* mirror_of(Collection* c, uint64_t dim) {
    auto it = c->mirrors.find(dim);
    if (it == c->mirrors.end()) return nullptr;
    if (!it->second->has_rows()) {  // built when no row of this dimension existed: nothing to patch, rebuild lazily
        c->mirrors.erase(it);
        return nullptr;
    }
    return it->second.get();
}

// ---- metadata columns of a mirror -------------------------------------------------------------------
// ScalarValue -> (kind, payload) cell; strings get (or take) an id in the field's dictionary
void encode_cell(FieldColumn& fc, const Value& v, uint8_t* kind, uint64_t* payload) {
    *payload = 0;
    switch (v.kind) {
        case NMN_VAL_BOOL: *kind = NMN_CELL_BOOL; *payload = v.b ? 1 : 0; break;
        case NMN_VAL_INT: *kind = NMN_CELL_INT; memcpy(payload, &v.i, 8); break;
        case NMN_VAL_FLOAT: *kind = NMN_CELL_FLOAT; memcpy(payload, &v.f, 8); break;
        case NMN_VAL_STRING: {
            *kind = NMN_CELL_STRING;
            auto it = fc.dict.find(v.s);
            if (it == fc.dict.end()) {
                it = fc.dict.emplace(v.s, (uint32_t)fc.strings.size()).first;
                fc.strings.push_back(v.s);
            }
            *payload = it->second;
            break;
        }
        default: *kind = NMN_CELL_NULL; break;
    }
}

// keep the device copy of one word of the live bitmap current (no-op without columns)
void columns_sync_valid(Mirror* m, uint64_t word) {
    if (!m->cols) return;
    if (nmn_columns_write_valid(m->cols, word, 1, &m->live[word]) != NMN_OK) m->drop_columns();
}

// write the metadata cells of one (new or overwritten) row; `overwrite` first drops the old cells
void columns_write_row(Mirror* m, uint64_t row, const Meta& meta, bool overwrite) {
    if (!m->cols) return;
    if (overwrite && nmn_columns_clear_row(m->cols, row) != NMN_OK) return m->drop_columns();
    for (const auto& kv : meta) {
        auto it = m->fields.find(kv.first);
        if (it == m->fields.end()) {
            FieldColumn fc;
            if (nmn_columns_add(m->cols, &fc.id) != NMN_OK) return m->drop_columns();
            it = m->fields.emplace(kv.first, std::move(fc)).first;
        }
        uint8_t kind;
        uint64_t payload;
        encode_cell(it->second, kv.second, &kind, &payload);
        if (nmn_columns_write(m->cols, it->second.id, row, 1, &kind, &payload) != NMN_OK) return m->drop_columns();
    }
}

// build the column set of a mirror from the store (first filtered search, or after drop_columns)
nmn_status columns_build(nmn_engine* e, Collection* c, Mirror* m) {
    if (m->cols) return NMN_OK;
    nmn_status st = nmn_columns_create(m->device, m->cap, &m->cols);
    if (st != NMN_OK) return err_gpu(st);
    const uint64_t n = m->row_to_slot.size();
    struct Staging {
        std::vector<uint8_t> kinds;
        std::vector<uint64_t> payload;
    };
    std::unordered_map<std::string, Staging> staging;
    for (uint64_t r = 0; r < n; r++) {
        if (!((m->live[r >> 6] >> (r & 63)) & 1ull)) continue;  // a dead row's slot may belong to another key now
        for (const auto& kv : c->slots[m->row_to_slot[r]].meta) {
            Staging& sg = staging[kv.first];
            if (sg.kinds.empty()) {
                sg.kinds.assign(n, NMN_CELL_ABSENT);
                sg.payload.assign(n, 0ull);
            }
            encode_cell(m->fields[kv.first], kv.second, &sg.kinds[r], &sg.payload[r]);
        }
    }
    for (auto& kv : staging) {
        FieldColumn& fc = m->fields[kv.first];
        st = nmn_columns_add(m->cols, &fc.id);
        if (st == NMN_OK) st = nmn_columns_write(m->cols, fc.id, 0, n, kv.second.kinds.data(), kv.second.payload.data());
        if (st != NMN_OK) {
            m->drop_columns();
            return err_gpu(st);
        }
    }
    st = nmn_columns_write_valid(m->cols, 0, m->live.size(), m->live.data());
    if (st != NMN_OK) {
        m->drop_columns();
        return err_gpu(st);
    }
    e->column_builds++;
    return NMN_OK;
}

// mark a mirrored row dead; rebuild later once a quarter of the mirror is dead
void mirror_tombstone(Collection* c, uint64_t dim, int64_t row) {
    Mirror* m = mirror_of(c, dim);
    if (!m || row < 0) return;
    m->live[(uint64_t)row >> 6] &= ~(1ull << ((uint64_t)row & 63));
    m->n_dead++;
    if (m->n_dead * 4 > m->row_to_slot.size()) c->mirrors.erase(dim);
    else columns_sync_valid(m, (uint64_t)row >> 6);
}

void mark_dirty(Collection* c, Mirror* m, uint64_t dim, uint64_t row) {
    (void)dim;
    m->dirty.push_back(row);
    c->has_dirty = true;
}

// append one vector to the mirror of its dimension; returns its row, or -1 (mirror absent / dropped)
int64_t mirror_append(Collection* c, uint64_t dim, const float* v, uint32_t slot, const Meta& meta) {
    Mirror* m = mirror_of(c, dim);
    if (!m) return -1;
    const uint64_t row = m->row_to_slot.size();
    (void)v;  // uploaded by mirror_flush from the slot's host copy
    if (row >= m->cap) {
        c->mirrors.erase(dim);  // out of spare capacity: rebuild on the next search
        return -1;
    }
    mark_dirty(c, m, dim, row);
    m->row_to_slot.push_back(slot);
    if ((row >> 6) >= m->live.size()) m->live.push_back(0ull);
    m->live[row >> 6] |= 1ull << (row & 63);
    columns_write_row(m, row, meta, false);  // a fresh row's cells are still ABSENT
    columns_sync_valid(m, row >> 6);
    return (int64_t)row;
}

nmn_status store_into(nmn_engine* e, Collection* c, const char* key, const float* v, uint64_t dim,
                      const nmn_meta_field* meta, uint32_t n_meta) {
    Entry ent;
    ent.key = key;
    ent.vec.assign(v, v + dim);
    for (uint32_t i = 0; i < n_meta; i++)
        if (meta[i].name) ent.meta[meta[i].name] = Value::from(meta[i].value);
    ent.live = true;
    auto it = c->by_key.find(ent.key);
    if (it != c->by_key.end()) {
        // put() overwrites the whole TensorData (vector and metadata)
        Entry& old = c->slots[it->second];
        const uint64_t old_dim = old.vec.size();
        Mirror* m = old.mrow >= 0 ? mirror_of(c, old_dim) : nullptr;
        if (m && old_dim == dim) {
            mark_dirty(c, m, dim, (uint64_t)old.mrow);
            ent.mrow = old.mrow;
            columns_write_row(m, (uint64_t)old.mrow, ent.meta, true);
        } else {
            if (m) mirror_tombstone(c, old_dim, old.mrow);
            ent.mrow = mirror_append(c, dim, v, it->second, ent.meta);
        }
        old = std::move(ent);
    } else {
        uint32_t slot;
        if (!c->free_slots.empty()) {
            slot = c->free_slots.back();
            c->free_slots.pop_back();
        } else {
            slot = (uint32_t)c->slots.size();
            c->slots.emplace_back();
        }
        ent.mrow = mirror_append(c, dim, v, slot, ent.meta);
        c->slots[slot] = std::move(ent);
        c->by_key[c->slots[slot].key] = slot;
        c->live++;
    }
    (void)e;
    return NMN_OK;
}

nmn_status delete_from(Collection* c, const std::string& key, const std::string& shown) {
    auto it = c->by_key.find(key);
    if (it == c->by_key.end()) return err_not_found(shown);
    Entry& ent = c->slots[it->second];
    mirror_tombstone(c, ent.vec.size(), ent.mrow);
    ent = Entry();
    c->free_slots.push_back(it->second);
    c->by_key.erase(it);
    c->live--;
    return NMN_OK;
}

// Upload the vectors recorded in m->dirty (exclusive lock held): runs of consecutive rows, one nmn_index_upload each.
// A row that died meanwhile still gets a vector (appends must stay contiguous; the live bitmap keeps it out of every
// scan).  On a device error the mirror is dropped and rebuilt by the caller's get_mirror.
bool mirror_flush(Collection* c, Mirror* m, uint64_t dim) {
    if (m->dirty.empty() || !m->has_rows()) {
        m->dirty.clear();
        return true;
    }
    std::sort(m->dirty.begin(), m->dirty.end());
    m->dirty.erase(std::unique(m->dirty.begin(), m->dirty.end()), m->dirty.end());
    std::vector<float> buf;
    size_t i = 0;
    while (i < m->dirty.size()) {
        size_t j = i + 1;
        while (j < m->dirty.size() && m->dirty[j] == m->dirty[j - 1] + 1 && j - i < 65536) j++;
        buf.assign((j - i) * dim, 0.0f);
        for (size_t r = i; r < j; r++) {
            const uint64_t row = m->dirty[r];
            if (row >= m->row_to_slot.size()) continue;
            const Entry& ent = c->slots[m->row_to_slot[row]];
            if (ent.live && ent.mrow == (int64_t)row && ent.vec.size() == dim)
                memcpy(buf.data() + (r - i) * dim, ent.vec.data(), dim * sizeof(float));
        }
        if (m->upload(buf.data(), m->dirty[i], j - i) != NMN_OK) return false;
        i = j;
    }
    m->dirty.clear();
    return true;
}

void flush_collection(Collection* c) {
    if (!c->has_dirty) return;
    c->has_dirty = false;
    for (auto it = c->mirrors.begin(); it != c->mirrors.end();) {
        if (!it->second->dirty.empty() && !mirror_flush(c, it->second.get(), it->first)) it = c->mirrors.erase(it);
        else ++it;
    }
}
void flush_dirty_mirrors(nmn_engine* e) {  // exclusive lock held
    flush_collection(&e->dflt);
    flush_collection(&e->entities);
    flush_collection(&e->artifacts);
    for (auto& kv : e->colls) flush_collection(&kv.second);
}

// Lazily (re)build the GPU mirror of the rows of `c` with dimension `dim`.
nmn_status get_mirror(nmn_engine* e, Collection* c, uint64_t dim, Mirror** out) {
    auto it = c->mirrors.find(dim);
    if (it != c->mirrors.end()) {
        if (mirror_flush(c, it->second.get(), dim)) {
            *out = it->second.get();
            return NMN_OK;
        }
        c->mirrors.erase(it);  // device error while patching: rebuild from the store
    }
    auto m = std::make_unique<Mirror>();
    m->device = e->cfg.device;
    uint64_t n = 0;
    for (const auto& ent : c->slots)
        if (ent.live && ent.vec.size() == dim) n++;  // `if stored_vec.len() != query.len() { return None }`
    if (n > 0) {
        // (an FFI caller's u64 dimension must not be truncated into a smaller, valid one; nmn_index_create's own limit is
        // one query in LDS: 40960 floats)
        if (dim > 0xFFFFFFFFull) return fail(NMN_ERR_INVALID_ARGUMENT, "dimension does not fit the device index (> 2^32 - 1)");
        nmn_index_desc d{};
        d.dim = (uint32_t)dim;
        // an engine serves concurrent search_similar callers, which share sweeps: rows of 300 / 200 / 100 floats are
        // stored with a stride of 384 / 256 / 128 so those batches take the matrix-core sweep (10M x 300, 64 queries:
        // 2.3 k -> 39.4 k q/s; one query alone 800 -> 718 q/s).  NMN_ENGINE_TIGHT_ROWS=1 keeps the stride at dim.
        static const bool tight_rows = getenv("NMN_ENGINE_TIGHT_ROWS") != nullptr;
        d.flags = tight_rows ? 0u : NMN_INDEX_WIDE_ROWS;
        m->cap = n + std::max<uint64_t>(n / 2, 1024);  // spare rows for appended keys
        d.capacity_rows = m->cap;
        d.row_base = 0;
        d.device = e->cfg.device;
        d.cand_cap = e->cfg.cand_cap;
        nmn_status st;
        if (e->cfg.n_devices >= 2) {
            // one logical index over the configured GPUs: the rows dealt evenly over them, every search on all of them at once,
            // per-shard top-k gathered (RCCL over xGMI / peer copies) and merged on devices[0] (nmn_sharded.hip).  The
            // metadata columns of the collection stay on devices[0]; a predicate's bitmap is sliced per shard by the search.
            nmn_sharded_desc sd{};
            sd.dim = d.dim;
            sd.flags = d.flags;
            sd.capacity_rows = m->cap;
            sd.row_base = 0;
            sd.n_shards = e->cfg.n_devices;
            sd.gather = NMN_GATHER_AUTO;
            sd.devices = e->cfg.devices;
            sd.cand_cap = e->cfg.cand_cap;
            // 64-row blocks dealt round-robin: the rows HELD (n of the n + 50 % capacity) and every later append are spread
            // evenly over the GPUs; contiguous ranges of the capacity would fill GPU 0, then GPU 1, ... and leave the last idle
            sd.layout = NMN_SHARDED_LAYOUT_CYCLIC;
            m->device = e->cfg.devices[0];
            st = nmn_sharded_create(&sd, &m->sh);
        } else {
            st = nmn_index_create(&d, &m->idx);
        }
        if (st != NMN_OK) return err_gpu(st);
        m->row_to_slot.reserve(n);
        // stage in chunks so the host copy stays small next to the store itself
        const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (dim * sizeof(float)));
        std::vector<float> buf;
        buf.reserve((size_t)std::min(chunk, n) * dim);
        uint64_t row0 = 0;
        auto flush = [&]() -> nmn_status {
            if (buf.empty()) return NMN_OK;
            const uint64_t cnt = buf.size() / dim;
            nmn_status s2 = m->upload(buf.data(), row0, cnt);
            row0 += cnt;
            buf.clear();
            return s2;
        };
        for (uint32_t s = 0; s < c->slots.size(); s++) {
            Entry& ent = c->slots[s];
            if (!ent.live || ent.vec.size() != dim) continue;
            buf.insert(buf.end(), ent.vec.begin(), ent.vec.end());
            ent.mrow = (int64_t)m->row_to_slot.size();
            m->row_to_slot.push_back(s);
            if (buf.size() >= chunk * dim) {
                st = flush();
                if (st != NMN_OK) return err_gpu(st);
            }
        }
        st = flush();
        if (st != NMN_OK) return err_gpu(st);
        m->live.assign((n + 63) / 64, ~0ull);
        if (n & 63) m->live.back() = (1ull << (n & 63)) - 1ull;
    }
    e->mirror_builds++;
    *out = m.get();
    c->mirrors[dim] = std::move(m);
    return NMN_OK;
}

// Run the GPU search over a mirror and map rows back to keys.
// `selected` (optional): a device bitmap of the rows that take part and how many bits it has set — the
// output of the predicate kernel, already ANDed with the live bitmap.
struct DeviceSelection {
    const uint64_t* mask_dev;
    uint64_t count;
};
// the same as a HOST bitmap over the mirror's rows (a subset of the live rows), e.g. the first max_keys_per_scan keys
struct HostSelection {
    const uint64_t* mask;
    uint64_t count;
};

nmn_status gpu_topk(Collection* c, Mirror* m, const float* q, uint64_t top_k, int32_t metric,
                    const DeviceSelection* selected, nmn_results* res, const HostSelection* host_sel = nullptr) {
    if (!m->has_rows()) return NMN_OK;  // no rows of this dimension
    const uint64_t rows = m->rows();
    const uint64_t taking_part = host_sel ? host_sel->count : selected ? selected->count : rows - std::min<uint64_t>(m->n_dead, rows);
    uint64_t k = std::min<uint64_t>(top_k, taking_part);
    if (k == 0) return NMN_OK;
    std::vector<uint64_t> out_rows(k);
    std::vector<float> out_scores(k);
    uint32_t count = 0;
    nmn_status st;
    if (m->sh) {
        // several GPUs: the bitmap travels as a host bitmap over global rows (the search slices it per shard); a
        // predicate's selection is read back from devices[0] first (rows / 8 bytes)
        std::vector<uint64_t> sel_host;
        const uint64_t* mask = host_sel ? host_sel->mask : m->n_dead ? m->live.data() : nullptr;
        if (selected) {
            sel_host.resize((size_t)((rows + 63) / 64));
            if (hipMemcpy(sel_host.data(), selected->mask_dev, sel_host.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(NMN_ERR_STORAGE, "Storage error: reading the selection bitmap back");
            mask = sel_host.data();
        }
        st = nmn_sharded_search(m->sh, q, 1, (uint32_t)k, (nmn_metric)metric, mask, out_rows.data(), out_scores.data(), &count,
                                nullptr);
    } else if (selected) {
        st = nmn_index_search_dmask_hint(m->idx, q, 1, (uint32_t)k, (nmn_metric)metric, selected->mask_dev, selected->count,
                                         out_rows.data(), out_scores.data(), &count, nullptr);
    } else {
        // deleted rows stay in the matrix until the next rebuild: the live bitmap keeps them out of every scan
        st = nmn_index_search(m->idx, q, 1, (uint32_t)k, (nmn_metric)metric,
                              host_sel ? host_sel->mask : m->n_dead ? m->live.data() : nullptr,
                              out_rows.data(), out_scores.data(), &count, nullptr);
    }
    if (st != NMN_OK) return err_gpu(st);
    for (uint32_t i = 0; i < count; i++) {
        res->keys.push_back(c->slots[m->row_to_slot[out_rows[i]]].key);
        res->scores.push_back(out_scores[i]);
    }
    return NMN_OK;
}

// shared body of search_similar / search_similar_with_metric / search_in_collection
nmn_status search_common(nmn_engine* e, Collection* c, const float* q, uint64_t dim, uint64_t top_k, int32_t metric,
                         const char* op, const Deadline& dl, const DeviceSelection* selected,
                         Mirror* prebuilt, nmn_results* res) {
    Mirror* m = prebuilt;
    if (!m) {
        nmn_status st = get_mirror(e, c, dim, &m);  // "let keys = self.store.scan(prefix)" stage
        if (st != NMN_OK) return st;
    }
    if (dl.expired()) return err_timeout(op, dl.ms);  // lib.rs:2005-2010
    nmn_status st = gpu_topk(c, m, q, top_k, metric, selected, res);
    if (st != NMN_OK) return st;
    if (dl.expired()) return err_timeout(op, dl.ms);  // lib.rs:2019-2024
    return NMN_OK;
}

// search_common under the shared lock when the mirror of (collection, dim) exists, else under the exclusive lock
// (which builds it).  `resolve` returns the collection (or null: nothing stored there) and runs under the lock.
template <typename Resolve>
nmn_status locked_search(nmn_engine* e, Resolve resolve, const float* q, uint64_t dim, uint64_t top_k, int32_t metric,
                         const char* op, const Deadline& dl, nmn_results* res) {
    for (int attempt = 0; attempt < 2; attempt++) {
        {
            ReadLock rd(e);
            Collection* c = resolve();
            if (!c) return NMN_OK;
            auto it = c->mirrors.find(dim);
            if (it != c->mirrors.end() && it->second->dirty.empty())
                return search_common(e, c, q, dim, top_k, metric, op, dl, nullptr, it->second.get(), res);
        }
        // the mirror is missing or has stores pending: build / patch it under the exclusive lock, then search under
        // the shared one like everybody else (a search under the exclusive lock cannot share a sweep with anybody)
        WriteLock wr(e);
        Collection* c = resolve();
        if (!c) return NMN_OK;
        Mirror* m = nullptr;
        nmn_status st = get_mirror(e, c, dim, &m);
        if (st != NMN_OK) return st;
    }
    WriteLock wr(e);  // stores keep arriving between our two locks: serve this one exclusively
    Collection* c = resolve();
    if (!c) return NMN_OK;
    return search_common(e, c, q, dim, top_k, metric, op, dl, nullptr, nullptr, res);
}

nmn_status validate_query(const nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, bool check_max_dim) {
    if (!q || dim == 0) return err_empty();          // query.is_empty() first (lib.rs:1953)
    if (top_k == 0) return err_topk();               // then top_k (lib.rs:1956)
    if (check_max_dim && e->cfg.max_dimension && dim > e->cfg.max_dimension)
        return err_dim(e->cfg.max_dimension, dim);   // lib.rs:1960-1967
    return NMN_OK;
}

nmn_results* new_results() { return new (std::nothrow) nmn_results(); }

// ---- FilterCondition -> predicate program (include/neumann_gpu.h, NMN_PRED_*) ------------------------
struct Program {
    std::vector<nmn_pred_op> ops;
    std::vector<uint64_t> consts;
};
struct Code {
    std::vector<nmn_pred_op> ops;
    uint32_t need = 1;  // stack slots the fragment uses
};

nmn_pred_op make_op(uint32_t op, uint32_t column = 0, uint32_t cmp = 0, uint32_t vkind = 0, uint64_t a = 0, uint64_t b = 0) {
    nmn_pred_op o;
    o.op = op;
    o.cmp = cmp;
    o.vkind = vkind;
    o.column = column;
    o.a = a;
    o.b = b;
    return o;
}

// the (kind, payload) a FilterValue compares as; strings never get here
bool filter_value_cell(const Value& v, uint32_t* kind, uint64_t* payload) {
    *payload = 0;
    switch (v.kind) {
        case NMN_VAL_NULL: *kind = NMN_CELL_NULL; return true;
        case NMN_VAL_BOOL: *kind = NMN_CELL_BOOL; *payload = v.b ? 1 : 0; return true;
        case NMN_VAL_INT: *kind = NMN_CELL_INT; memcpy(payload, &v.i, 8); return true;
        case NMN_VAL_FLOAT: *kind = NMN_CELL_FLOAT; memcpy(payload, &v.f, 8); return true;
        default: return false;
    }
}

// bitset over the dictionary ids of one string column: bit i = test(strings[i])
template <typename Test>
nmn_pred_op string_set(const FieldColumn& fc, Program& p, Test test) {
    const uint64_t n = fc.strings.size(), off = p.consts.size();
    p.consts.resize(off + (n + 63) / 64, 0ull);
    for (uint64_t i = 0; i < n; i++)
        if (test(fc.strings[i])) p.consts[off + (i >> 6)] |= 1ull << (i & 63);
    return make_op(NMN_PRED_STRSET, fc.id, 0, 0, off, n);
}

// Postfix code of a condition.  And/Or are pure here, so the deeper operand is emitted first
// (Sethi-Ullman order): the stack need grows with log2 of the leaf count, not with the nesting depth.
Code compile_filter(const nmn_filter& f, const Mirror& m, Program& p) {
    Code c;
    auto field = [&](const std::string& name) -> const FieldColumn* {
        auto it = m.fields.find(name);
        return it == m.fields.end() ? nullptr : &it->second;
    };
    switch (f.kind) {
        case nmn_filter::True: c.ops.push_back(make_op(NMN_PRED_TRUE)); return c;
        case nmn_filter::And:
        case nmn_filter::Or: {
            Code a = compile_filter(*f.a, m, p), b = compile_filter(*f.b, m, p);
            if (a.need < b.need) std::swap(a, b);
            c.ops = std::move(a.ops);
            c.ops.insert(c.ops.end(), b.ops.begin(), b.ops.end());
            c.ops.push_back(make_op(f.kind == nmn_filter::And ? NMN_PRED_AND : NMN_PRED_OR));
            c.need = std::max(a.need, b.need + 1);
            return c;
        }
        default: break;
    }
    const FieldColumn* fc = field(f.field);
    if (!fc) {  // no mirrored row has the field: `tensor.get(..)` is None for every row
        c.ops.push_back(make_op(NMN_PRED_FALSE));
        return c;
    }
    switch (f.kind) {
        case nmn_filter::Exists: c.ops.push_back(make_op(NMN_PRED_EXISTS, fc->id)); break;
        case nmn_filter::Contains:
            c.ops.push_back(string_set(*fc, p, [&](const std::string& s) { return s.find(f.text) != std::string::npos; }));
            break;
        case nmn_filter::StartsWith:
            c.ops.push_back(string_set(*fc, p, [&](const std::string& s) { return s.compare(0, f.text.size(), f.text) == 0; }));
            break;
        case nmn_filter::Cmp:
            if (f.value.kind == NMN_VAL_STRING) {
                const int op = f.op;
                c.ops.push_back(string_set(*fc, p, [&](const std::string& s) {
                    const int ord = s.compare(f.value.s);
                    switch (op) {
                        case NMN_OP_EQ: return ord == 0;
                        case NMN_OP_NE: return ord != 0;
                        case NMN_OP_LT: return ord < 0;
                        case NMN_OP_LE: return ord <= 0;
                        case NMN_OP_GT: return ord > 0;
                        default: return ord >= 0;
                    }
                }));
            } else {
                uint32_t vk;
                uint64_t vp;
                if (filter_value_cell(f.value, &vk, &vp)) c.ops.push_back(make_op(NMN_PRED_CMP, fc->id, (uint32_t)f.op, vk, vp));
                else c.ops.push_back(make_op(NMN_PRED_FALSE));
            }
            break;
        case nmn_filter::In: {
            // scalar members -> one IN list; string members -> one bitset; In = any member equal
            const uint64_t off = p.consts.size();
            uint64_t n_scalar = 0;
            bool any_string = false;
            for (const auto& v : f.values) {
                uint32_t vk;
                uint64_t vp;
                if (v.kind == NMN_VAL_STRING) any_string = true;
                else if (filter_value_cell(v, &vk, &vp)) {
                    p.consts.push_back(vk);
                    p.consts.push_back(vp);
                    n_scalar++;
                }
            }
            if (n_scalar) c.ops.push_back(make_op(NMN_PRED_IN, fc->id, 0, 0, off, n_scalar));
            if (any_string) {
                c.ops.push_back(string_set(*fc, p, [&](const std::string& s) {
                    for (const auto& v : f.values)
                        if (v.kind == NMN_VAL_STRING && v.s == s) return true;
                    return false;
                }));
                if (n_scalar) {
                    c.ops.push_back(make_op(NMN_PRED_OR));
                    c.need = 2;
                }
            }
            if (c.ops.empty()) c.ops.push_back(make_op(NMN_PRED_FALSE));  // `values.iter().any(..)` of nothing
            break;
        }
        default: c.ops.push_back(make_op(NMN_PRED_FALSE)); break;
    }
    return c;
}

// Pre-filter strategy (lib.rs:3514-3557 / 1776-1796): the predicate runs on the GPU over the mirror's
// metadata columns and leaves the selection bitmap in HBM; the masked scan reads it in place.  Exact.
// A filtered search first runs under the SHARED lock (so that many of them overlap: each predicate evaluation gets a
// bitmap of its own, and the searches that consume them share corpus sweeps); anything that would have to change the
// engine — building the mirror or its metadata columns — makes it return this and run again under the exclusive lock.
constexpr nmn_status kRetryExclusive = 0x7e7e;

nmn_status pre_filter_search(nmn_engine* e, Collection* c, const float* q, uint64_t dim, uint64_t top_k,
                             const nmn_filter& f, const char* op, const Deadline& dl, nmn_results* res, bool shared) {
    Mirror* m = nullptr;
    if (shared) {
        auto it = c->mirrors.find(dim);
        if (it == c->mirrors.end()) return kRetryExclusive;
        m = it->second.get();
        if (!m->dirty.empty() || (m->has_rows() && !m->cols)) return kRetryExclusive;
    } else {
        nmn_status st = get_mirror(e, c, dim, &m);
        if (st != NMN_OK) return st;
    }
    if (!m->has_rows()) return NMN_OK;
    nmn_status st = shared ? NMN_OK : columns_build(e, c, m);
    if (st != NMN_OK) return st;
    Program p;
    Code code = compile_filter(f, *m, p);
    p.ops = std::move(code.ops);
    // predicate and search in one call (nmn_index_search_pred): concurrent filtered searches then wait for ONE batch —
    // their predicates are evaluated together on the batch's stream, right before the sweep that serves them all
    if (dl.expired()) return err_timeout(op, dl.ms);  // lib.rs:2005-2010
    const uint64_t rows = m->rows();
    const uint64_t k = std::min<uint64_t>(std::min<uint64_t>(top_k, rows), NMN_MAX_TOP_K);
    if (k == 0 || rows != m->row_to_slot.size()) {
        if (k == 0) return NMN_OK;
        return fail(NMN_ERR_STORAGE, "mirror and metadata columns disagree on the row count");
    }
    if (top_k > NMN_MAX_TOP_K || m->sh) {
        // beyond the candidate pipeline: evaluate, then the large-k path over the bitmap (the two-step form); also the
        // form of a mirror that spans several GPUs (evaluate on devices[0], search every shard under its slice)
        DeviceSelection sel{nullptr, 0};
        uint32_t slot = 0;
        st = nmn_columns_eval_acquire(m->cols, p.ops.data(), (uint32_t)p.ops.size(), p.consts.data(), p.consts.size(),
                                      m->row_to_slot.size(), &sel.count, &slot, &sel.mask_dev);
        if (st != NMN_OK) return err_gpu(st);
        e->device_filters++;
        st = NMN_OK;
        if (sel.count != 0)  // else `if matching_keys.is_empty() { return Vec::new() }`
            st = search_common(e, c, q, dim, top_k, NMN_METRIC_COSINE, op, dl, &sel, m, res);
        (void)nmn_columns_eval_release(m->cols, slot);
        return st;
    }
    std::vector<uint64_t> out_rows(k);
    std::vector<float> out_scores(k);
    uint32_t count = 0;
    uint64_t selected = 0;
    st = nmn_index_search_pred(m->idx, m->cols, p.ops.data(), (uint32_t)p.ops.size(), p.consts.data(), p.consts.size(), q, 1,
                               (uint32_t)k, NMN_METRIC_COSINE, out_rows.data(), out_scores.data(), &count, &selected, nullptr);
    if (st != NMN_OK) return err_gpu(st);
    e->device_filters++;
    for (uint32_t i = 0; i < count; i++) {
        res->keys.push_back(c->slots[m->row_to_slot[out_rows[i]]].key);
        res->scores.push_back(out_scores[i]);
    }
    if (dl.expired()) return err_timeout(op, dl.ms);  // lib.rs:2019-2024
    return NMN_OK;
}

// what a shared-lock filtered search needs to find in place (exclusive lock held)
nmn_status prepare_filtered(nmn_engine* e, Collection* c, uint64_t dim) {
    Mirror* m = nullptr;
    nmn_status st = get_mirror(e, c, dim, &m);
    if (st != NMN_OK || !m->has_rows()) return st;
    return columns_build(e, c, m);
}

// search_common for the post-filter arm: under the shared lock only when the mirror exists already
nmn_status search_common_mode(nmn_engine* e, Collection* c, const float* q, uint64_t dim, uint64_t top_k, int32_t metric,
                              const char* op, const Deadline& dl, nmn_results* res, bool shared) {
    Mirror* prebuilt = nullptr;
    if (shared) {
        auto it = c->mirrors.find(dim);
        if (it == c->mirrors.end() || !it->second->dirty.empty()) return kRetryExclusive;
        prebuilt = it->second.get();
    }
    return search_common(e, c, q, dim, top_k, metric, op, dl, nullptr, prebuilt, res);
}

}  // namespace

// ================================================================================================
extern "C" {

void nmn_engine_config_default(nmn_engine_config* c) {  // lib.rs:648-664
    if (!c) return;
    c->default_dimension = 0;
    c->sparse_threshold = 0.5f;
    c->parallel_threshold = 5000;
    c->default_metric = NMN_METRIC_COSINE;
    c->max_dimension = 0;
    c->max_keys_per_scan = 0;
    c->search_timeout_ms = -1;
    c->device = -1;
    c->cand_cap = 0;
    c->max_index_file_bytes = 100ll * 1024 * 1024;  // lib.rs:660
    c->max_index_entries = 1000000;                 // lib.rs:661
    c->n_devices = 0;
    for (uint32_t i = 0; i < NMN_ENGINE_MAX_DEVICES; i++) c->devices[i] = -1;
}

void nmn_filtered_config_default(nmn_filtered_config* c) {  // lib.rs:412-420
    if (!c) return;
    c->strategy = NMN_FILTER_AUTO;
    c->selectivity_threshold = 0.1f;
    c->oversample_factor = 3;
}

const char* nmn_engine_last_error(void) { return g_err.c_str(); }

nmn_status nmn_engine_create(const nmn_engine_config* config, nmn_engine** out) {
    if (!out) return fail(NMN_ERR_INVALID_ARGUMENT, "out is null");
    nmn_engine* e = new (std::nothrow) nmn_engine();
    if (!e) return fail(NMN_ERR_OUT_OF_MEMORY, "engine alloc");
    if (config) e->cfg = *config;
    else nmn_engine_config_default(&e->cfg);
    // VectorEngineConfig::validate (lib.rs:710-740)
    if (!(e->cfg.sparse_threshold >= 0.0f && e->cfg.sparse_threshold <= 1.0f)) {
        delete e;
        return fail(NMN_ERR_CONFIGURATION, "Configuration error: sparse_threshold must be between 0.0 and 1.0");
    }
    if (e->cfg.parallel_threshold == 0) {
        delete e;
        return fail(NMN_ERR_CONFIGURATION, "Configuration error: parallel_threshold must be greater than 0");
    }
    if (e->cfg.max_index_file_bytes == 0) {  // lib.rs:740-746
        delete e;
        return fail(NMN_ERR_CONFIGURATION, "Configuration error: max_index_file_bytes must be greater than 0");
    }
    if (e->cfg.max_index_entries == 0) {  // lib.rs:747-753
        delete e;
        return fail(NMN_ERR_CONFIGURATION, "Configuration error: max_index_entries must be greater than 0");
    }
    if (e->cfg.n_devices > NMN_ENGINE_MAX_DEVICES) {
        delete e;
        return fail(NMN_ERR_CONFIGURATION, "Configuration error: n_devices exceeds NMN_ENGINE_MAX_DEVICES");
    }
    // devices[]: one entry = that GPU; two or more = every mirror is one index over all of them; the collection-level
    // device objects that are not sharded (metadata columns, IVF indexes, the compute_similarity slot) live on devices[0]
    if (e->cfg.n_devices >= 1) e->cfg.device = e->cfg.devices[0];
    *out = e;
    return NMN_OK;
}

void nmn_engine_destroy(nmn_engine* e) { delete e; }

nmn_status nmn_engine_store_embedding_with_metadata(nmn_engine* e, const char* key, const float* v, uint64_t dim,
                                                    const nmn_meta_field* meta, uint32_t n_meta) {
    if (!e || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!v || dim == 0) return err_empty();  // lib.rs:1841-1843
    if (e->cfg.max_dimension && dim > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, dim);
    WriteLock g(e);
    return store_into(e, &e->dflt, key, v, dim, meta, n_meta);
}

nmn_status nmn_engine_store_embedding(nmn_engine* e, const char* key, const float* v, uint64_t dim) {
    return nmn_engine_store_embedding_with_metadata(e, key, v, dim, nullptr, 0);
}

nmn_status nmn_engine_batch_store(nmn_engine* e, const char* const* keys, const float* rows, uint64_t n,
                                  uint64_t dim) {
    if (!e || !keys || (!rows && n)) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (n && dim == 0) return err_empty();
    if (e->cfg.max_dimension && dim > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, dim);
    WriteLock g(e);
    for (uint64_t i = 0; i < n; i++) {
        if (!keys[i]) return fail(NMN_ERR_INVALID_ARGUMENT, "null key");
        store_into(e, &e->dflt, keys[i], rows + i * dim, dim, nullptr, 0);
    }
    return NMN_OK;
}

static nmn_status get_from(Collection* c, const std::string& key, const std::string& shown, float* out, uint64_t cap,
                           uint64_t* dim_out) {
    if (!c) return err_not_found(shown);
    auto it = c->by_key.find(key);
    if (it == c->by_key.end()) return err_not_found(shown);
    const Entry& ent = c->slots[it->second];
    if (dim_out) *dim_out = ent.vec.size();
    if (out) memcpy(out, ent.vec.data(), std::min<uint64_t>(cap, ent.vec.size()) * sizeof(float));
    return NMN_OK;
}

nmn_status nmn_engine_get_embedding(nmn_engine* e, const char* key, float* out, uint64_t cap, uint64_t* dim_out) {
    if (!e || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return get_from(&e->dflt, key, key, out, cap, dim_out);
}

nmn_status nmn_engine_delete_embedding(nmn_engine* e, const char* key) {
    if (!e || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return delete_from(&e->dflt, key, key);
}

int32_t nmn_engine_exists(nmn_engine* e, const char* key) {
    if (!e || !key) return 0;
    WriteLock g(e);
    return e->dflt.by_key.count(key) ? 1 : 0;
}

uint64_t nmn_engine_count(nmn_engine* e) {
    if (!e) return 0;
    WriteLock g(e);
    return e->dflt.live;
}

// max_keys_per_scan (lib.rs:638, 2322, 2341, 2948, 3179, 3225): every unbounded walk over the store stops after that many
// keys of the scan.  The reference's scan order is its HashSet's (slab_router.rs:287-305: unspecified); here it is slot
// order, so "the first N keys" is a deterministic prefix.  0 = None.
static uint64_t scan_limit(const nmn_engine* e) { return e->cfg.max_keys_per_scan ? e->cfg.max_keys_per_scan : UINT64_MAX; }

nmn_strlist* nmn_engine_list_keys(nmn_engine* e) {  // list_keys_bounded, lib.rs:2321-2329
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !l) return l;
    WriteLock g(e);
    const uint64_t limit = scan_limit(e);
    for (const auto& ent : e->dflt.slots) {
        if (l->items.size() >= limit) break;
        if (ent.live) l->items.push_back(ent.key);
    }
    return l;
}

// list_keys_paginated (lib.rs:2945-2980): scan -> take(min(skip + limit.unwrap_or(max_scan), max_scan)) -> skip -> take(limit);
// total_count = count() when asked for; has_more = skip + items < total, or (no total) items == limit.unwrap_or(0)
nmn_strlist* nmn_engine_list_keys_paginated(nmn_engine* e, uint64_t skip, int64_t limit, int32_t count_total,
                                            int64_t* total_count, int32_t* has_more) {
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (total_count) *total_count = -1;
    if (has_more) *has_more = 0;
    if (!e || !l) return l;
    WriteLock g(e);
    const uint64_t max_scan = scan_limit(e);
    uint64_t fetch = skip + (limit >= 0 ? (uint64_t)limit : max_scan);
    if (fetch < skip) fetch = UINT64_MAX;  // saturating_add
    fetch = std::min(fetch, max_scan);
    uint64_t seen = 0;
    for (const auto& ent : e->dflt.slots) {
        if (seen >= fetch) break;
        if (!ent.live) continue;
        if (seen++ < skip) continue;
        if (limit >= 0 && l->items.size() >= (uint64_t)limit) break;
        l->items.push_back(ent.key);
    }
    const uint64_t n = l->items.size();
    if (count_total) {
        if (total_count) *total_count = (int64_t)e->dflt.live;
        uint64_t reach = skip + n;
        if (reach < skip) reach = UINT64_MAX;
        if (has_more) *has_more = reach < e->dflt.live ? 1 : 0;
    } else if (has_more) {
        *has_more = n == (limit >= 0 ? (uint64_t)limit : 0ull) ? 1 : 0;
    }
    return l;
}

nmn_status nmn_engine_clear(nmn_engine* e, uint64_t* removed) {  // lib.rs:2340-2354
    if (!e) return fail(NMN_ERR_INVALID_ARGUMENT, "null engine");
    WriteLock g(e);
    const uint64_t max_keys = scan_limit(e);
    if (e->dflt.live <= max_keys) {
        if (removed) *removed = e->dflt.live;
        e->dflt = Collection();
        return NMN_OK;
    }
    // more keys than one bounded scan covers: the first max_keys go ("call again until 0 is returned")
    std::vector<std::string> keys;
    for (const auto& ent : e->dflt.slots) {
        if (keys.size() >= max_keys) break;
        if (ent.live) keys.push_back(ent.key);
    }
    for (const auto& k : keys) delete_from(&e->dflt, k, k);
    if (removed) *removed = keys.size();
    return NMN_OK;
}

// ---- searches ------------------------------------------------------------------------------------
nmn_status nmn_engine_search_similar_with_metric(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k,
                                                 int32_t metric, nmn_results** out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, /*check_max_dim=*/false);  // lib.rs:2056-2061
    if (st != NMN_OK) return st;
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    // zero-magnitude queries: empty for cosine/dot, scored for euclidean (lib.rs:2063-2068)
    if (!(zero_magnitude(q, dim) && metric != NMN_METRIC_EUCLIDEAN)) {
        st = locked_search(e, [&] { return &e->dflt; }, q, dim, top_k, metric, "search_similar_with_metric", dl, res);
        if (st != NMN_OK) {
            delete res;
            return st;
        }
    }
    *out = res;
    return NMN_OK;
}

nmn_status nmn_engine_search_similar(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, nmn_results** out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, /*check_max_dim=*/true);  // lib.rs:1953-1967
    if (st != NMN_OK) return st;
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    if (!zero_magnitude(q, dim)) {  // lib.rs:1970-1974
        st = locked_search(e, [&] { return &e->dflt; }, q, dim, top_k, NMN_METRIC_COSINE, "search_similar", dl, res);
        if (st != NMN_OK) {
            delete res;
            return st;
        }
    }
    *out = res;
    return NMN_OK;
}

// ---- metadata CRUD of stored embeddings (lib.rs:3311-3385): filtered searches see the change at once ----
static void value_out(const Value& v, nmn_value* out) {
    out->kind = v.kind;
    out->b = v.b ? 1 : 0;
    out->i = v.i;
    out->f = v.f;
    out->s = v.kind == NMN_VAL_STRING ? v.s.c_str() : nullptr;
}

nmn_metalist* nmn_engine_get_metadata(nmn_engine* e, const char* key, nmn_status* status) {  // lib.rs:3312-3327
    nmn_status dummy;
    if (!status) status = &dummy;
    if (!e || !key) {
        *status = fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
        return nullptr;
    }
    WriteLock g(e);
    auto it = e->dflt.by_key.find(key);
    if (it == e->dflt.by_key.end()) {
        *status = err_not_found(key);
        return nullptr;
    }
    nmn_metalist* l = new (std::nothrow) nmn_metalist();
    if (!l) {
        *status = fail(NMN_ERR_OUT_OF_MEMORY, "metadata list alloc");
        return nullptr;
    }
    for (const auto& kv : e->dflt.slots[it->second].meta) {
        l->names.push_back(kv.first);
        l->values.push_back(kv.second);
    }
    *status = NMN_OK;
    return l;
}
uint64_t nmn_metalist_len(const nmn_metalist* l) { return l ? l->names.size() : 0; }
const char* nmn_metalist_name(const nmn_metalist* l, uint64_t i) { return (l && i < l->names.size()) ? l->names[i].c_str() : nullptr; }
nmn_status nmn_metalist_value(const nmn_metalist* l, uint64_t i, nmn_value* out) {
    if (!l || !out || i >= l->values.size()) return fail(NMN_ERR_INVALID_ARGUMENT, "bad metadata index");
    value_out(l->values[i], out);  // out->s points into the list: valid until nmn_metalist_free
    return NMN_OK;
}
void nmn_metalist_free(nmn_metalist* l) { delete l; }

// update_metadata (lib.rs:3329-3351): set the given fields, keep the others and the vector
nmn_status nmn_engine_update_metadata(nmn_engine* e, const char* key, const nmn_meta_field* meta, uint32_t n_meta) {
    if (!e || !key || (n_meta && !meta)) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    auto it = e->dflt.by_key.find(key);
    if (it == e->dflt.by_key.end()) return err_not_found(key);
    Entry& ent = e->dflt.slots[it->second];
    Meta changed;
    for (uint32_t i = 0; i < n_meta; i++)
        if (meta[i].name) {
            ent.meta[meta[i].name] = Value::from(meta[i].value);
            changed[meta[i].name] = ent.meta[meta[i].name];
        }
    Mirror* m = ent.mrow >= 0 ? mirror_of(&e->dflt, ent.vec.size()) : nullptr;
    if (m) columns_write_row(m, (uint64_t)ent.mrow, changed, /*overwrite=*/false);  // only the touched cells
    return NMN_OK;
}

// remove_metadata_field (lib.rs:3353-3364): NotFound only for a missing key; a missing field is not an error
nmn_status nmn_engine_remove_metadata_field(nmn_engine* e, const char* key, const char* field) {
    if (!e || !key || !field) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    auto it = e->dflt.by_key.find(key);
    if (it == e->dflt.by_key.end()) return err_not_found(key);
    Entry& ent = e->dflt.slots[it->second];
    if (ent.meta.erase(field) == 0) return NMN_OK;
    Mirror* m = ent.mrow >= 0 ? mirror_of(&e->dflt, ent.vec.size()) : nullptr;
    if (m && m->cols) {
        auto fc = m->fields.find(field);
        if (fc != m->fields.end()) {
            const uint8_t kind = NMN_CELL_ABSENT;
            const uint64_t payload = 0;
            if (nmn_columns_write(m->cols, fc->second.id, (uint64_t)ent.mrow, 1, &kind, &payload) != NMN_OK) m->drop_columns();
        }
    }
    return NMN_OK;
}

int32_t nmn_engine_has_metadata_field(nmn_engine* e, const char* key, const char* field) {  // lib.rs:3366-3372
    if (!e || !key || !field) return 0;
    WriteLock g(e);
    auto it = e->dflt.by_key.find(key);
    return (it != e->dflt.by_key.end() && e->dflt.slots[it->second].meta.count(field)) ? 1 : 0;
}

// get_metadata_field (lib.rs:3374-3383): Ok(None) for a missing field -> *present = 0.  out->s (strings) points to
// thread-local storage that stays valid until this thread's next call of this function.
nmn_status nmn_engine_get_metadata_field(nmn_engine* e, const char* key, const char* field, nmn_value* out,
                                         int32_t* present) {
    if (!e || !key || !field || !out || !present) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    static thread_local Value hold;
    *present = 0;
    WriteLock g(e);
    auto it = e->dflt.by_key.find(key);
    if (it == e->dflt.by_key.end()) return err_not_found(key);
    const Meta& meta = e->dflt.slots[it->second].meta;
    auto f = meta.find(field);
    if (f == meta.end()) return NMN_OK;
    hold = f->second;
    value_out(hold, out);
    *present = 1;
    return NMN_OK;
}

// estimate_filter_selectivity (lib.rs:3695-3711): the first min(100, count) keys
nmn_status nmn_engine_estimate_filter_selectivity(nmn_engine* e, const nmn_filter* f, float* out) {
    if (!e || !f || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    uint64_t sample = 0, matches = 0;
    for (const auto& ent : e->dflt.slots) {
        if (!ent.live) continue;
        if (sample == 100) break;
        sample++;
        if (evaluate_filter(ent.meta, *f)) matches++;
    }
    *out = sample ? (float)matches / (float)sample : 0.0f;
    return NMN_OK;
}

nmn_strlist* nmn_engine_list_keys_matching(nmn_engine* e, const nmn_filter* f) {  // lib.rs:3720-3725
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !f || !l) return l;
    WriteLock g(e);
    for (const auto& ent : e->dflt.slots)
        if (ent.live && evaluate_filter(ent.meta, *f)) l->items.push_back(ent.key);
    return l;
}

// batch_delete_embeddings (lib.rs:2924-2940): missing keys are skipped, the number deleted is returned
nmn_status nmn_engine_batch_delete(nmn_engine* e, const char* const* keys, uint64_t n, uint64_t* deleted) {
    if (!e || (n && !keys)) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; i++)
        if (keys[i] && e->dflt.by_key.count(keys[i]) && delete_from(&e->dflt, keys[i], keys[i]) == NMN_OK) cnt++;
    if (deleted) *deleted = cnt;
    return NMN_OK;
}

uint64_t nmn_engine_dimension(nmn_engine* e) {  // lib.rs:2298-2308: the first stored vector's length; 0 = None
    if (!e) return 0;
    WriteLock g(e);
    for (const auto& ent : e->dflt.slots)
        if (ent.live) return ent.vec.size();
    return 0;
}

int32_t nmn_engine_exists_in_collection(nmn_engine* e, const char* coll, const char* key) {  // lib.rs:1537-1540
    if (!e || !coll || !key) return 0;
    WriteLock g(e);
    Collection* c = e->storage(coll, false);
    return (c && c->by_key.count(key)) ? 1 : 0;
}

nmn_strlist* nmn_engine_list_collection_keys(nmn_engine* e, const char* coll) {  // lib.rs:1543-1550
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !coll || !l) return l;
    WriteLock g(e);
    Collection* c = e->storage(coll, false);
    if (c)
        for (const auto& ent : c->slots)
            if (ent.live) l->items.push_back(ent.key);
    return l;
}

// search_similar_paginated / search_entities_paginated (lib.rs:2988-3058): search min(skip + limit.unwrap_or(top_k),
// top_k) results, then skip / take; total_count (when asked for) is the number of results FOUND, not of rows stored.
static nmn_status paginate(nmn_status st, nmn_results* all, uint64_t skip, int64_t limit, int32_t count_total,
                           nmn_results** out, int64_t* total_count, int32_t* has_more) {
    if (st != NMN_OK) return st;
    const uint64_t found = all->keys.size();
    if (total_count) *total_count = count_total ? (int64_t)found : -1;
    nmn_results* res = new_results();
    if (!res) {
        delete all;
        return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    }
    for (uint64_t i = skip; i < found && (limit < 0 || res->keys.size() < (uint64_t)limit); i++) {
        res->keys.push_back(all->keys[i]);
        res->scores.push_back(all->scores[i]);
    }
    if (has_more) {
        uint64_t reach = skip + res->keys.size();
        if (reach < skip) reach = UINT64_MAX;  // saturating_add
        *has_more = (limit >= 0 && count_total && reach < found) ? 1 : 0;
    }
    delete all;
    *out = res;
    return NMN_OK;
}

static uint64_t total_needed(uint64_t top_k, uint64_t skip, int64_t limit) {
    const uint64_t want = limit >= 0 ? (uint64_t)limit : top_k;
    uint64_t need = skip + want;
    if (need < skip) need = UINT64_MAX;  // saturating_add
    return std::min(need, top_k);
}

nmn_status nmn_engine_search_similar_paginated(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, uint64_t skip,
                                               int64_t limit, int32_t count_total, nmn_results** out, int64_t* total_count,
                                               int32_t* has_more) {
    if (!out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_results* all = nullptr;
    nmn_status st = nmn_engine_search_similar(e, q, dim, total_needed(top_k, skip, limit), &all);
    return paginate(st, all, skip, limit, count_total, out, total_count, has_more);
}

nmn_status nmn_engine_search_entities_paginated(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, uint64_t skip,
                                                int64_t limit, int32_t count_total, nmn_results** out, int64_t* total_count,
                                                int32_t* has_more) {
    if (!out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_results* all = nullptr;
    nmn_status st = nmn_engine_search_entities(e, q, dim, total_needed(top_k, skip, limit), &all);
    return paginate(st, all, skip, limit, count_total, out, total_count, has_more);
}

// ---- tensor_blob artifact similarity (tensor_blob/src/lib.rs:520-625) -----------------------------------
// The blob store keeps an optional `_embedding` (+ `_id`, `_filename`) in each artifact's `_blob:meta:{id}`
// record; `search_by_embedding` scans those records, keeps the ones of the query's dimension and ranks them by
// the f64 sparse cosine (NMN_METRIC_SPARSE_COSINE_F64).  Here the embeddings of the artifacts are one more
// mirrored key space; the artifact bytes, chunks, tags and links are the blob store's business, not this path's.
nmn_status nmn_engine_blob_set_embedding(nmn_engine* e, const char* artifact_id, const char* filename, const float* v,
                                         uint64_t dim) {
    if (!e || !artifact_id) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!v || dim == 0) return err_empty();
    nmn_meta_field mf{};
    mf.name = "_filename";
    mf.value.kind = NMN_VAL_STRING;
    mf.value.s = filename ? filename : "";
    WriteLock g(e);
    return store_into(e, &e->artifacts, artifact_id, v, dim, &mf, 1);
}

nmn_status nmn_engine_blob_remove(nmn_engine* e, const char* artifact_id) {  // BlobStore::delete drops the record
    if (!e || !artifact_id) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return delete_from(&e->artifacts, artifact_id, artifact_id);
}

static void fill_filenames(Collection* c, nmn_results* res) {
    res->aux.clear();
    for (const auto& key : res->keys) {
        auto it = c->by_key.find(key);
        std::string fn;
        if (it != c->by_key.end()) {
            auto m = c->slots[it->second].meta.find("_filename");
            if (m != c->slots[it->second].meta.end()) fn = m->second.s;  // get_string(..).unwrap_or_default()
        }
        res->aux.push_back(std::move(fn));
    }
}

// search_by_embedding (lib.rs:591-625): no validation in the reference — an empty query or k == 0 simply finds nothing
nmn_status nmn_engine_blob_search_by_embedding(nmn_engine* e, const float* q, uint64_t dim, uint64_t k, nmn_results** out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    if (q && dim && k) {
        const Deadline dl(-1);
        nmn_status st = locked_search(e, [&] { return &e->artifacts; }, q, dim, k, NMN_METRIC_SPARSE_COSINE_F64,
                                      "search_by_embedding", dl, res);
        if (st != NMN_OK) {
            delete res;
            return st;
        }
        ReadLock rd(e);
        fill_filenames(&e->artifacts, res);
    }
    *out = res;
    return NMN_OK;
}

// similar (lib.rs:563-583): search k + 1 with the artifact's own embedding, drop the artifact itself, take k
nmn_status nmn_engine_blob_similar(nmn_engine* e, const char* artifact_id, uint64_t k, nmn_results** out) {
    if (!e || !artifact_id || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    std::vector<float> emb;
    {
        WriteLock g(e);
        auto it = e->artifacts.by_key.find(artifact_id);
        if (it == e->artifacts.by_key.end()) return err_not_found(artifact_id);  // BlobError::NotFound
        emb = e->artifacts.slots[it->second].vec;
    }
    nmn_results* all = nullptr;
    nmn_status st = nmn_engine_blob_search_by_embedding(e, emb.data(), emb.size(), k + 1, &all);
    if (st != NMN_OK) return st;
    nmn_results* res = new_results();
    if (!res) {
        delete all;
        return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    }
    for (size_t i = 0; i < all->keys.size() && res->keys.size() < k; i++) {
        if (all->keys[i] == artifact_id) continue;
        res->keys.push_back(all->keys[i]);
        res->scores.push_back(all->scores[i]);
        res->aux.push_back(all->aux[i]);
    }
    delete all;
    *out = res;
    return NMN_OK;
}

// ---- IVF (lib.rs:2641-2812) --------------------------------------------------------------------------
void nmn_ivf_options_default(nmn_ivf_options* o) {  // IVFConfig::default + KMeansConfig::default
    if (!o) return;
    o->num_clusters = 100;
    o->nprobe = 0;  // default_nprobe(num_clusters) = ceil(sqrt(num_clusters)), ivf.rs:46-56
    o->max_iterations = 100;
    o->convergence_threshold = 1e-4f;
    o->seed = 42;
    o->init_method = NMN_KMEANS_INIT_PLUSPLUS;
}

nmn_status nmn_engine_build_ivf_index(nmn_engine* e, const nmn_ivf_options* options, nmn_engine_ivf** out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_ivf_options o;
    if (options) o = *options;
    else nmn_ivf_options_default(&o);
    auto res = std::unique_ptr<nmn_engine_ivf>(new (std::nothrow) nmn_engine_ivf());
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "ivf alloc");
    res->nprobe = o.nprobe ? o.nprobe : (uint64_t)std::ceil(std::sqrt((float)o.num_clusters));
    WriteLock g(e);
    // `let keys = self.list_keys()` then every vector in that order; one dimension only (lib.rs:2653-2668)
    std::vector<float> rows;
    uint64_t dim = 0;
    for (const auto& ent : e->dflt.slots) {
        if (!ent.live) continue;
        if (dim == 0) dim = ent.vec.size();
        else if (ent.vec.size() != dim) return err_dim(dim, ent.vec.size());
        res->keys.push_back(ent.key);
        rows.insert(rows.end(), ent.vec.begin(), ent.vec.end());
    }
    const uint64_t n = res->keys.size();
    if (n == 0) {  // `return Ok((IVFIndex::new(options.config), Vec::new()))`: untrained, searches find nothing
        *out = res.release();
        return NMN_OK;
    }
    if (e->cfg.max_dimension && dim > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, dim);  // lib.rs:2671-2680
    res->dim = dim;
    // index.train(&vectors) then index.add(v) for every vector — both on the GPU, the k-means bit for bit
    // (nmn_ivf_build: exact centroid sweeps for the assignments, sequential per-(cluster, dimension) sums for the update)
    if (dim > 0xFFFFFFFFull) return fail(NMN_ERR_INVALID_ARGUMENT, "dimension does not fit the device index (> 2^32 - 1)");
    nmn_index_desc d{};
    d.dim = (uint32_t)dim;
    d.capacity_rows = n;
    d.device = e->cfg.device;
    d.cand_cap = e->cfg.cand_cap;
    nmn_kmeans_options ko{};
    ko.max_iterations = o.max_iterations;
    ko.convergence_threshold = o.convergence_threshold;
    ko.seed = o.seed;
    ko.init_method = o.init_method == NMN_KMEANS_INIT_RANDOM ? 0 : 1;
    nmn_status st = nmn_ivf_build(&d, rows.data(), n, (uint32_t)std::min<uint64_t>(o.num_clusters, UINT32_MAX), &ko, &res->index);
    if (st != NMN_OK) return err_gpu(st);
    res->n_clusters = nmn_ivf_clusters(res->index);
    res->centroids.resize((size_t)res->n_clusters * dim);
    st = nmn_ivf_centroids(res->index, res->centroids.data(), res->centroids.size());
    if (st != NMN_OK) return err_gpu(st);
    *out = res.release();
    return NMN_OK;
}

void nmn_engine_ivf_free(nmn_engine_ivf* ivf) { delete ivf; }
uint64_t nmn_engine_ivf_len(const nmn_engine_ivf* ivf) { return (ivf && ivf->index) ? nmn_ivf_len(ivf->index) : 0; }
uint32_t nmn_engine_ivf_clusters(const nmn_engine_ivf* ivf) { return ivf ? ivf->n_clusters : 0; }
uint64_t nmn_engine_ivf_nprobe(const nmn_engine_ivf* ivf) { return ivf ? ivf->nprobe : 0; }
const char* nmn_engine_ivf_key(const nmn_engine_ivf* ivf, uint64_t id) {
    return (ivf && id < ivf->keys.size()) ? ivf->keys[id].c_str() : nullptr;
}
nmn_status nmn_engine_ivf_centroids(const nmn_engine_ivf* ivf, float* out, uint64_t cap_floats) {
    if (!ivf || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (cap_floats < ivf->centroids.size()) return fail(NMN_ERR_BUFFER_TOO_SMALL, "centroid buffer too small");
    memcpy(out, ivf->centroids.data(), ivf->centroids.size() * sizeof(float));
    return NMN_OK;
}
nmn_status nmn_engine_ivf_cluster_sizes(nmn_engine_ivf* ivf, uint64_t* out) {
    if (!ivf || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!ivf->index) return NMN_OK;
    nmn_status st = nmn_ivf_cluster_sizes(ivf->index, out);
    return st == NMN_OK ? NMN_OK : err_gpu(st);
}

// search_with_ivf / search_with_ivf_nprobe (lib.rs:2731-2812); nprobe == 0 = the index's own
nmn_status nmn_engine_search_with_ivf(nmn_engine* e, nmn_engine_ivf* ivf, const float* q, uint64_t dim, uint64_t top_k,
                                      uint64_t nprobe, nmn_results** out) {
    if (!e || !ivf || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    if (!q || dim == 0) return err_empty();  // lib.rs:2717-2719
    if (top_k == 0) return err_topk();       // lib.rs:2720-2722
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    if (ivf->index) {  // untrained index: `return Vec::new()` (ivf.rs:326-328)
        if (dim != ivf->dim) {
            delete res;
            return err_dim(ivf->dim, dim);  // the reference zips and silently truncates; refused here
        }
        const uint64_t len = nmn_ivf_len(ivf->index);
        const uint32_t k = (uint32_t)std::min<uint64_t>(top_k, std::max<uint64_t>(len, 1));
        std::vector<uint64_t> ids(k);
        std::vector<float> dist(k);
        uint32_t count = 0;
        nmn_status st = nmn_ivf_search(ivf->index, q, 1, k, (uint32_t)std::min<uint64_t>(nprobe ? nprobe : ivf->nprobe, UINT32_MAX),
                                       ids.data(), dist.data(), &count, nullptr);
        if (st != NMN_OK) {
            delete res;
            return err_gpu(st);
        }
        if (dl.expired()) {  // lib.rs:2726-2731
            delete res;
            return err_timeout(nprobe ? "search_with_ivf_nprobe" : "search_with_ivf", dl.ms);
        }
        for (uint32_t i = 0; i < count; i++) {
            if (ids[i] >= ivf->keys.size()) continue;  // `key_mapping.get(vector_id)` -> filter_map
            res->keys.push_back(ivf->keys[ids[i]]);
            const float denom = 1.0f + dist[i];
            res->scores.push_back(1.0f / denom);  // "IVF returns distances, convert to similarity"
        }
    }
    *out = res;
    return NMN_OK;
}

// ---- unified entity mode (lib.rs:3060-3237) --------------------------------------------------------
// Entity keys ("user:1") carry their vector in the `_embedding` field of the entity's TensorData
// (fields::EMBEDDING, tensor_store/src/lib.rs:183).  On this path only that field matters, so the
// entities are one more key space with its own GPU mirror; `search_entities` is the same cosine scan
// over it (lib.rs:3155-3219: rows without `_embedding` or of another dimension are skipped).
nmn_status nmn_engine_set_entity_embedding(nmn_engine* e, const char* entity_key, const float* v, uint64_t dim) {
    if (!e || !entity_key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!v || dim == 0) return err_empty();  // lib.rs:3073-3075
    if (e->cfg.max_dimension && dim > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, dim);  // 3077-3084
    WriteLock g(e);
    return store_into(e, &e->entities, entity_key, v, dim, nullptr, 0);
}

nmn_status nmn_engine_get_entity_embedding(nmn_engine* e, const char* entity_key, float* out, uint64_t cap,
                                           uint64_t* dim_out) {
    if (!e || !entity_key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return get_from(&e->entities, entity_key, entity_key, out, cap, dim_out);  // NotFound(entity_key), lib.rs:3110-3119
}

int32_t nmn_engine_entity_has_embedding(nmn_engine* e, const char* entity_key) {  // lib.rs:3123-3128
    if (!e || !entity_key) return 0;
    WriteLock g(e);
    return e->entities.by_key.count(entity_key) ? 1 : 0;
}

nmn_status nmn_engine_remove_entity_embedding(nmn_engine* e, const char* entity_key) {  // lib.rs:3135-3147
    if (!e || !entity_key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return delete_from(&e->entities, entity_key, entity_key);
}

nmn_strlist* nmn_engine_scan_entities_with_embeddings(nmn_engine* e) {  // lib.rs:3224-3232
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !l) return l;
    WriteLock g(e);
    // `.scan("").take(max_scan).filter(entity_has_embedding)`: the reference's bound counts every key of its store in an
    // unspecified order; this key space only holds entities WITH an embedding, so the bound counts those
    const uint64_t limit = scan_limit(e);
    for (const auto& ent : e->entities.slots) {
        if (l->items.size() >= limit) break;
        if (ent.live) l->items.push_back(ent.key);
    }
    return l;
}

uint64_t nmn_engine_count_entities_with_embeddings(nmn_engine* e) {  // lib.rs:3235-3237: scan_entities_with_embeddings().len()
    if (!e) return 0;
    WriteLock g(e);
    return std::min<uint64_t>(e->entities.live, scan_limit(e));
}

nmn_status nmn_engine_search_entities(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k, nmn_results** out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, /*check_max_dim=*/true);  // lib.rs:3158-3173
    if (st != NMN_OK) return st;
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    if (!zero_magnitude(q, dim)) {  // lib.rs:3175-3178
        bool bounded = false;
        {
            ReadLock rd(e);
            bounded = e->entities.live > scan_limit(e);
        }
        if (bounded) {
            // `let keys = self.store.scan("").into_iter().take(max_scan)` (lib.rs:3179-3180): only the first max_scan keys of
            // the scan take part (then the `_embedding` / dimension filter): a selection bitmap over the mirror's rows.
            // (The reference's scan("") counts EVERY key of the store towards max_scan; this mirror of the engine holds entity
            //  keys only, so the bound is applied to them — INTEGRATION.md §5 states the deviation.)
            // Locking as locked_search: the mirror is built / patched under the exclusive lock, the bitmap and the GPU search
            // run under the shared one, so bounded searches overlap with every other reader.
            auto bounded_search = [&](Collection* c, Mirror* m) -> nmn_status {
                if (dl.expired()) return err_timeout("search_entities", dl.ms);
                if (!m->has_rows()) return NMN_OK;
                std::vector<uint64_t> sel(m->live.size(), 0ull);
                uint64_t seen = 0, picked = 0;
                const uint64_t limit = scan_limit(e);
                for (const auto& ent : c->slots) {
                    if (seen >= limit) break;
                    if (!ent.live) continue;
                    seen++;
                    if (ent.vec.size() == dim && ent.mrow >= 0 && ((m->live[(uint64_t)ent.mrow >> 6] >> ((uint64_t)ent.mrow & 63)) & 1ull)) {
                        sel[(uint64_t)ent.mrow >> 6] |= 1ull << ((uint64_t)ent.mrow & 63);
                        picked++;
                    }
                }
                const HostSelection hs{sel.data(), picked};
                nmn_status s2 = gpu_topk(c, m, q, top_k, NMN_METRIC_COSINE, nullptr, res, &hs);
                if (s2 == NMN_OK && dl.expired()) s2 = err_timeout("search_entities", dl.ms);
                return s2;
            };
            bool served = false;
            for (int attempt = 0; attempt < 2 && !served && st == NMN_OK; attempt++) {
                {
                    ReadLock rd(e);
                    Collection* c = &e->entities;
                    auto it = c->mirrors.find(dim);
                    if (it != c->mirrors.end() && it->second->dirty.empty()) {
                        st = bounded_search(c, it->second.get());
                        served = true;
                        break;
                    }
                }
                WriteLock wr(e);  // the mirror is missing or has stores pending: build / patch it exclusively, search shared
                Mirror* m = nullptr;
                st = get_mirror(e, &e->entities, dim, &m);
            }
            if (!served && st == NMN_OK) {  // stores keep arriving between our two locks: serve this one exclusively
                WriteLock wr(e);
                Mirror* m = nullptr;
                st = get_mirror(e, &e->entities, dim, &m);
                if (st == NMN_OK) st = bounded_search(&e->entities, m);
            }
        } else {
            st = locked_search(e, [&] { return &e->entities; }, q, dim, top_k, NMN_METRIC_COSINE, "search_entities", dl, res);
        }
        if (st != NMN_OK) {
            delete res;
            return st;
        }
    }
    *out = res;
    return NMN_OK;
}

static int choose_strategy(Collection* c, const nmn_filter& f, const nmn_filtered_config& cfg) {
    // choose_filter_strategy (lib.rs:3480-3511): True -> post; sample the first 100 keys
    if (f.kind == nmn_filter::True) return NMN_FILTER_POST;
    uint64_t sample = 0, matches = 0;
    for (const auto& ent : c->slots) {
        if (!ent.live) continue;
        if (sample == 100) break;
        sample++;
        if (evaluate_filter(ent.meta, f)) matches++;
    }
    if (sample == 0) return NMN_FILTER_POST;
    const float sel = (float)matches / (float)sample;
    return sel < cfg.selectivity_threshold ? NMN_FILTER_PRE : NMN_FILTER_POST;
}

static nmn_status post_filter(Collection* c, const nmn_filter& f, nmn_results* cand, uint64_t top_k, bool truncate,
                              nmn_results* res) {
    for (size_t i = 0; i < cand->keys.size(); i++) {
        auto it = c->by_key.find(cand->keys[i]);
        if (it == c->by_key.end()) continue;
        if (!evaluate_filter(c->slots[it->second].meta, f)) continue;
        if (truncate && res->keys.size() >= top_k) break;  // `.take(top_k)`
        res->keys.push_back(cand->keys[i]);
        res->scores.push_back(cand->scores[i]);
    }
    return NMN_OK;
}

nmn_status nmn_engine_search_similar_filtered(nmn_engine* e, const float* q, uint64_t dim, uint64_t top_k,
                                              const nmn_filter* filter, const nmn_filtered_config* config,
                                              nmn_results** out) {
    if (!e || !out || !filter) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, true);  // lib.rs:3438-3453
    if (st != NMN_OK) return st;
    nmn_filtered_config cfg;
    if (config) cfg = *config;
    else nmn_filtered_config_default(&cfg);
    auto run = [&](bool shared) -> nmn_status {
        nmn_results* res = new_results();
        if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
        int strategy = cfg.strategy == NMN_FILTER_AUTO ? choose_strategy(&e->dflt, *filter, cfg) : cfg.strategy;
        if (dl.expired()) {
            delete res;
            return err_timeout("search_similar_filtered", dl.ms);
        }
        nmn_status rs = NMN_OK;
        if (strategy == NMN_FILTER_POST) {
            // search_with_post_filter (lib.rs:3560-3579): search_similar(k * oversample) then filter, take k
            uint64_t over = top_k * cfg.oversample_factor;
            if (cfg.oversample_factor && over / cfg.oversample_factor != top_k) over = UINT64_MAX;  // saturating_mul
            over = std::max(over, top_k);
            nmn_results cand;
            if (!zero_magnitude(q, dim))
                rs = search_common_mode(e, &e->dflt, q, dim, over, NMN_METRIC_COSINE, "search_similar", dl, &cand, shared);
            if (rs == NMN_OK) post_filter(&e->dflt, *filter, &cand, top_k, true, res);
        } else if (!zero_magnitude(q, dim)) {  // search_with_pre_filter (lib.rs:3520-3523)
            rs = pre_filter_search(e, &e->dflt, q, dim, top_k, *filter, "search_similar_filtered", dl, res, shared);
        }
        if (rs != NMN_OK) {
            delete res;
            return rs;
        }
        *out = res;
        return NMN_OK;
    };
    {
        ReadLock rd(e);
        st = run(true);
    }
    if (st == kRetryExclusive) {  // build / patch the mirror and its columns exclusively, then search shared again
        {
            WriteLock g(e);
            (void)prepare_filtered(e, &e->dflt, dim);
        }
        ReadLock rd(e);
        st = run(true);
    }
    if (st == kRetryExclusive) {
        WriteLock g(e);
        st = run(false);
    }
    return st;
}

nmn_status nmn_engine_compute_similarity(nmn_engine* e, const float* a, uint64_t na, const float* b, uint64_t nb,
                                         float* out) {
    if (!e || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!a || !b || na == 0 || nb == 0) return err_empty();  // lib.rs:2279-2281
    if (na != nb) return err_dim(na, nb);                    // lib.rs:2282-2287
    if (na > 0xFFFFFFFFull) return fail(NMN_ERR_INVALID_ARGUMENT, "dimension does not fit the device index (> 2^32 - 1)");
    WriteLock g(e);
    auto& slot = e->scratch[na];
    if (!slot) {
        slot = std::make_unique<Mirror>();
        nmn_index_desc d{};
        d.dim = (uint32_t)na;
        d.capacity_rows = 1;
        d.device = e->cfg.device;
        nmn_status st = nmn_index_create(&d, &slot->idx);
        if (st != NMN_OK) {
            e->scratch.erase(na);
            return err_gpu(st);
        }
    }
    nmn_status st = nmn_index_upload(slot->idx, b, 0, 1);
    if (st != NMN_OK) return err_gpu(st);
    const uint64_t row = 0;
    // a_magnitude == 0 -> 0.0 (lib.rs:2289-2292) is what the exact kernel returns for |q| == 0
    st = nmn_index_score_rows(slot->idx, a, 1, NMN_METRIC_COSINE, &row, 1, out);
    return st == NMN_OK ? NMN_OK : err_gpu(st);
}

// ---- collections ---------------------------------------------------------------------------------
nmn_status nmn_engine_create_collection(nmn_engine* e, const char* name, uint64_t dimension, int32_t metric) {
    if (!e || !name) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    if (e->configs.count(name)) return fail(NMN_ERR_COLLECTION_EXISTS, std::string("Collection already exists: ") + name);
    CollectionConfig cc;
    cc.dimension = dimension;
    cc.metric = metric;
    e->configs[name] = cc;
    return NMN_OK;
}

nmn_status nmn_engine_delete_collection(nmn_engine* e, const char* name) {
    if (!e || !name) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    if (!e->configs.count(name)) return fail(NMN_ERR_COLLECTION_NOT_FOUND, std::string("Collection not found: ") + name);
    e->configs.erase(name);
    e->colls.erase(name);  // "Delete all embeddings in the collection"
    return NMN_OK;
}

int32_t nmn_engine_collection_exists(nmn_engine* e, const char* name) {
    if (!e || !name) return 0;
    WriteLock g(e);
    return e->configs.count(name) ? 1 : 0;
}

uint64_t nmn_engine_collection_count(nmn_engine* e, const char* name) {
    if (!e || !name) return 0;
    WriteLock g(e);
    Collection* c = e->storage(name, false);
    return c ? c->live : 0;
}

nmn_strlist* nmn_engine_list_collections(nmn_engine* e) {
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !l) return l;
    WriteLock g(e);
    for (const auto& kv : e->configs) l->items.push_back(kv.first);
    return l;
}

nmn_status nmn_engine_store_in_collection(nmn_engine* e, const char* coll, const char* key, const float* v,
                                          uint64_t dim, const nmn_meta_field* meta, uint32_t n_meta) {
    if (!e || !coll || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (!v || dim == 0) return err_empty();  // lib.rs:1454-1456
    WriteLock g(e);
    auto cit = e->configs.find(coll);
    if (cit != e->configs.end() && cit->second.dimension && dim != cit->second.dimension)
        return err_dim(cit->second.dimension, dim);  // lib.rs:1459-1469
    if (e->cfg.max_dimension && dim > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, dim);
    return store_into(e, e->storage(coll, true), key, v, dim, meta, n_meta);
}

nmn_status nmn_engine_get_from_collection(nmn_engine* e, const char* coll, const char* key, float* out,
                                          uint64_t cap, uint64_t* dim_out) {
    if (!e || !coll || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    return get_from(e->storage(coll, false), key, std::string(coll) + ":" + key, out, cap, dim_out);
}

nmn_status nmn_engine_delete_from_collection(nmn_engine* e, const char* coll, const char* key) {
    if (!e || !coll || !key) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    Collection* c = e->storage(coll, false);
    const std::string shown = std::string(coll) + ":" + key;
    if (!c) return err_not_found(shown);
    return delete_from(c, key, shown);
}

nmn_status nmn_engine_search_in_collection(nmn_engine* e, const char* coll, const float* q, uint64_t dim,
                                           uint64_t top_k, nmn_results** out) {
    if (!e || !out || !coll) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, false);  // lib.rs:1593-1598
    if (st != NMN_OK) return st;
    int32_t metric = NMN_METRIC_COSINE;
    {
        ReadLock rd(e);
        auto cit = e->configs.find(coll);
        if (cit != e->configs.end()) {
            if (cit->second.dimension && dim != cit->second.dimension) return err_dim(cit->second.dimension, dim);
            metric = cit->second.metric;  // lib.rs:1614-1616
        }
    }
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    if (!(zero_magnitude(q, dim) && metric == NMN_METRIC_COSINE)) {  // lib.rs:1617-1620
        st = locked_search(e, [&] { return e->storage(coll, false); }, q, dim, top_k, metric, "search_in_collection", dl, res);
        if (st != NMN_OK) {
            delete res;
            return st;
        }
    }
    *out = res;
    return NMN_OK;
}

nmn_status nmn_engine_search_filtered_in_collection(nmn_engine* e, const char* coll, const float* q, uint64_t dim,
                                                    uint64_t top_k, const nmn_filter* filter,
                                                    const nmn_filtered_config* config, nmn_results** out) {
    if (!e || !out || !coll || !filter) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const Deadline dl(e->cfg.search_timeout_ms);
    nmn_status st = validate_query(e, q, dim, top_k, false);  // lib.rs:1708-1713
    if (st != NMN_OK) return st;
    nmn_filtered_config cfg;
    if (config) cfg = *config;
    else nmn_filtered_config_default(&cfg);
    auto run = [&](bool shared) -> nmn_status {
    auto cit = e->configs.find(coll);
    int32_t coll_metric = NMN_METRIC_COSINE;
    if (cit != e->configs.end()) {
        if (cit->second.dimension && dim != cit->second.dimension) return err_dim(cit->second.dimension, dim);
        coll_metric = cit->second.metric;
    }
    nmn_results* res = new_results();
    if (!res) return fail(NMN_ERR_OUT_OF_MEMORY, "results alloc");
    Collection* c = e->storage(coll, false);
    if (!c || zero_magnitude(q, dim)) {  // lib.rs:1729-1733
        *out = res;
        return NMN_OK;
    }
    int strategy = cfg.strategy;
    if (strategy == NMN_FILTER_AUTO) {  // lib.rs:1737-1764 (no special case for True here)
        uint64_t sample = 0, matches = 0;
        for (const auto& ent : c->slots) {
            if (!ent.live) continue;
            if (sample == 100) break;
            sample++;
            if (evaluate_filter(ent.meta, *filter)) matches++;
        }
        strategy = (sample == 0 || !((float)matches / (float)sample < cfg.selectivity_threshold)) ? NMN_FILTER_POST
                                                                                                  : NMN_FILTER_PRE;
    }
    if (dl.expired()) {
        delete res;
        return err_timeout("search_filtered_in_collection", dl.ms);
    }
    if (strategy == NMN_FILTER_POST) {
        // lib.rs:1797-1813: search_in_collection(k*oversample) with the COLLECTION's metric, filter, then the
        // common sort + truncate(k)
        uint64_t over = top_k * cfg.oversample_factor;
        if (cfg.oversample_factor && over / cfg.oversample_factor != top_k) over = UINT64_MAX;
        over = std::max(over, top_k);
        nmn_results cand;
        nmn_status rs = search_common_mode(e, c, q, dim, over, coll_metric, "search_in_collection", dl, &cand, shared);
        if (rs != NMN_OK) {
            delete res;
            return rs;
        }
        post_filter(c, *filter, &cand, top_k, false, res);
        if (res->keys.size() > top_k) {  // results.truncate(top_k) — candidates are already sorted
            res->keys.resize(top_k);
            res->scores.resize(top_k);
        }
    } else {
        nmn_status rs = pre_filter_search(e, c, q, dim, top_k, *filter, "search_filtered_in_collection", dl, res, shared);
        if (rs != NMN_OK) {
            delete res;
            return rs;
        }
    }
    if (dl.expired()) {
        delete res;
        return err_timeout("search_filtered_in_collection", dl.ms);
    }
    *out = res;
    return NMN_OK;
    };
    {
        ReadLock rd(e);
        st = run(true);
    }
    if (st == kRetryExclusive) {
        {
            WriteLock g(e);
            Collection* c = e->storage(coll, false);
            if (c) (void)prepare_filtered(e, c, dim);
        }
        ReadLock rd(e);
        st = run(true);
    }
    if (st == kRetryExclusive) {
        WriteLock g(e);
        st = run(false);
    }
    return st;
}

// ---- measurement aid: per-call latency of search_similar from one host thread (bench.py published_shapes) ----------
nmn_status nmn_engine_search_probe(nmn_engine* e, const float* queries, uint64_t n_queries, uint64_t dim, uint64_t top_k,
                                   uint64_t calls, float* out_us) {
    if (!e || !queries || !out_us || n_queries == 0) return NMN_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < calls; i++) {
        nmn_results* r = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        const nmn_status st = nmn_engine_search_similar(e, queries + (i % n_queries) * dim, dim, top_k, &r);
        uint64_t got = nmn_results_len(r);
        volatile float sink = got ? nmn_results_score(r, got - 1) : 0.f;  // (the caller reads its answer)
        (void)sink;
        nmn_results_free(r);
        out_us[i] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (st != NMN_OK) return st;
    }
    return NMN_OK;
}

// ---- results / lists / filters -------------------------------------------------------------------
uint64_t nmn_results_len(const nmn_results* r) { return r ? r->keys.size() : 0; }
const char* nmn_results_key(const nmn_results* r, uint64_t i) { return (r && i < r->keys.size()) ? r->keys[i].c_str() : nullptr; }
float nmn_results_score(const nmn_results* r, uint64_t i) { return (r && i < r->scores.size()) ? r->scores[i] : NAN; }
const char* nmn_results_aux(const nmn_results* r, uint64_t i) { return (r && i < r->aux.size()) ? r->aux[i].c_str() : nullptr; }
void nmn_results_free(nmn_results* r) { delete r; }
uint64_t nmn_strlist_len(const nmn_strlist* l) { return l ? l->items.size() : 0; }
const char* nmn_strlist_get(const nmn_strlist* l, uint64_t i) { return (l && i < l->items.size()) ? l->items[i].c_str() : nullptr; }
void nmn_strlist_free(nmn_strlist* l) { delete l; }

nmn_filter* nmn_filter_cmp(int32_t op, const char* field, const nmn_value* value) {
    if (!field || !value || op < NMN_OP_EQ || op > NMN_OP_GE) return nullptr;
    nmn_filter* f = new (std::nothrow) nmn_filter();
    if (!f) return nullptr;
    f->kind = nmn_filter::Cmp;
    f->op = op;
    f->field = field;
    f->value = Value::from(*value);
    return f;
}
static nmn_filter* binary(nmn_filter::Kind k, nmn_filter* a, nmn_filter* b) {
    if (!a || !b) {
        delete a;
        delete b;
        return nullptr;
    }
    nmn_filter* f = new (std::nothrow) nmn_filter();
    if (!f) return nullptr;
    f->kind = k;
    f->a.reset(a);
    f->b.reset(b);
    return f;
}
nmn_filter* nmn_filter_and(nmn_filter* a, nmn_filter* b) { return binary(nmn_filter::And, a, b); }
nmn_filter* nmn_filter_or(nmn_filter* a, nmn_filter* b) { return binary(nmn_filter::Or, a, b); }
nmn_filter* nmn_filter_true(void) { return new (std::nothrow) nmn_filter(); }
static nmn_filter* text_filter(nmn_filter::Kind k, const char* field, const char* text) {
    if (!field) return nullptr;
    nmn_filter* f = new (std::nothrow) nmn_filter();
    if (!f) return nullptr;
    f->kind = k;
    f->field = field;
    if (text) f->text = text;
    return f;
}
nmn_filter* nmn_filter_exists(const char* field) { return text_filter(nmn_filter::Exists, field, nullptr); }
nmn_filter* nmn_filter_contains(const char* field, const char* s) { return text_filter(nmn_filter::Contains, field, s); }
nmn_filter* nmn_filter_starts_with(const char* field, const char* s) { return text_filter(nmn_filter::StartsWith, field, s); }
nmn_filter* nmn_filter_in(const char* field, const nmn_value* values, uint32_t n) {
    if (!field || (!values && n)) return nullptr;
    nmn_filter* f = new (std::nothrow) nmn_filter();
    if (!f) return nullptr;
    f->kind = nmn_filter::In;
    f->field = field;
    for (uint32_t i = 0; i < n; i++) f->values.push_back(Value::from(values[i]));
    return f;
}
void nmn_filter_free(nmn_filter* f) { delete f; }

uint64_t nmn_engine_count_matching(nmn_engine* e, const nmn_filter* f) {
    if (!e || !f) return 0;
    WriteLock g(e);
    uint64_t n = 0;
    for (const auto& ent : e->dflt.slots)
        if (ent.live && evaluate_filter(ent.meta, *f)) n++;
    return n;
}

uint64_t nmn_engine_device_filter_evals(nmn_engine* e) {
    if (!e) return 0;
    WriteLock g(e);
    return e->device_filters;
}

uint64_t nmn_engine_column_builds(nmn_engine* e) {
    if (!e) return 0;
    WriteLock g(e);
    return e->column_builds;
}

uint64_t nmn_engine_mirror_builds(nmn_engine* e) {
    if (!e) return 0;
    WriteLock g(e);
    return e->mirror_builds;
}

// rows the mirror of (default collection, dim) holds on each of its GPUs: out[0 .. min(cap, shards)); returns the number of
// shards (1 for a single-GPU mirror, 0 when no mirror of that dimension exists)
uint32_t nmn_engine_mirror_shard_rows(nmn_engine* e, uint64_t dim, uint64_t* out, uint32_t cap) {
    if (!e) return 0;
    WriteLock g(e);
    auto it = e->dflt.mirrors.find(dim);
    if (it == e->dflt.mirrors.end() || !it->second->has_rows()) return 0;
    Mirror* m = it->second.get();
    if (!mirror_flush(&e->dflt, m, dim)) {  // (a failed flush leaves a mirror nobody may trust: dropped, as every other caller does)
        e->dflt.mirrors.erase(it);
        return 0;
    }
    if (m->idx) {
        if (out && cap) out[0] = nmn_index_rows(m->idx);
        return 1;
    }
    const uint32_t G = nmn_sharded_shards(m->sh);
    for (uint32_t i = 0; i < G && i < cap && out; i++) out[i] = nmn_index_rows(nmn_sharded_shard(m->sh, i));
    return G;
}

// device memory of the mirror of (default collection, dim), summed over its shards: out[0] = f32 rows, out[1] = the 8-bit / bf16
// mirrors that exist right now, out[2] = per-row factors (nmn_index_hbm_bytes); returns the number of shards (0: no such mirror)
uint32_t nmn_engine_mirror_hbm_bytes(nmn_engine* e, uint64_t dim, uint64_t out[3]) {
    if (!e || !out) return 0;
    // an accounting query: the shared lock, and no flush of pending rows — the sizes depend on the shards' CAPACITY, not on which
    // rows have been copied (it used to take the exclusive lock, flush, and ignore the flush's verdict)
    ReadLock g(e);
    out[0] = out[1] = out[2] = 0;
    auto it = e->dflt.mirrors.find(dim);
    if (it == e->dflt.mirrors.end() || !it->second->has_rows()) return 0;
    Mirror* m = it->second.get();
    const uint32_t G = m->idx ? 1u : nmn_sharded_shards(m->sh);
    for (uint32_t i = 0; i < G; i++) {
        uint64_t a = 0, b = 0, c = 0;
        (void)nmn_index_hbm_bytes(m->idx ? m->idx : nmn_sharded_shard(m->sh, i), &a, &b, &c);
        out[0] += a;
        out[1] += b;
        out[2] += c;
    }
    return G;
}

int32_t nmn_engine_mirror_cached(nmn_engine* e, const char* coll) {
    if (!e) return 0;
    WriteLock g(e);
    Collection* c = e->storage(coll, false);
    return (c && !c->mirrors.empty()) ? 1 : 0;
}


// ---- index persistence (lib.rs:500-623, 3733-4000) --------------------------------------------------------------------------
// save_index / load_index: PersistentVectorIndex { collection, config, vectors: [{key, vector, metadata}], created_at,
// version } as serde_json writes it (externally tagged MetadataValue: "Null", {"Bool": b}, {"Int": i}, {"Float": f},
// {"String": s}; DistanceMetric as "Cosine" / "Euclidean" / "DotProduct"; Option::None as null) — a file either side can
// read.  save_index_binary / load_index_binary: the reference's second format is bitcode, an undocumented bit-packed
// encoding of an absent third-party crate; the binary format here is this library's own (nmn_persist.h): the same
// snapshot with the vectors as flat shard sections — the matrix as the GPU mirror holds it, magnitudes included — so a
// load is sequential reads + bulk H2D copies, and the mirror is rebuilt (and checked against the stored magnitudes) at
// once instead of on the first search.  Both loads apply max_index_file_bytes BEFORE reading and max_index_entries after
// decoding, with the reference's texts (lib.rs:3831-3856).
}  // extern "C"

#include "nmn_persist.h"

namespace {

constexpr const char* kDefaultCollection = "default";  // VectorEngine::DEFAULT_COLLECTION (lib.rs:1359)
constexpr uint32_t kPersistVersion = 1;                // PersistentVectorIndex::CURRENT_VERSION (lib.rs:579)

nmn_status err_serialization(const std::string& what) { return fail(NMN_ERR_SERIALIZATION, "Serialization error: " + what); }
nmn_status err_io(const std::string& what) { return fail(NMN_ERR_IO, "IO error: " + what); }
nmn_status err_config(const std::string& what) { return fail(NMN_ERR_CONFIGURATION, "Configuration error: " + what); }

// ---- a small JSON reader / writer: exactly what serde_json's output of PersistentVectorIndex needs ---------------------
struct JVal {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    bool is_int = false;  // the number token had no fraction / exponent and fits i64
    int64_t i = 0;
    double d = 0.0;
    std::string s;
    std::vector<double> nums;                          // an array of numbers only (a vector): kept flat
    std::vector<JVal> arr;                             // any other array
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;
    bool fail_at(const char* what) {
        if (err.empty()) err = std::string(what) + " at byte " + std::to_string((size_t)(p - begin));
        return false;
    }
    const char* begin;
    JParser(const char* b, size_t n) : p(b), end(b + n), begin(b) {}
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    }
    bool lit(const char* w) {
        const size_t n = strlen(w);
        if ((size_t)(end - p) < n || memcmp(p, w, n) != 0) return fail_at("invalid literal");
        p += n;
        return true;
    }
    static void utf8(std::string& o, uint32_t c) {
        if (c < 0x80) o += (char)c;
        else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
        else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t* out) {
        if (end - p < 4) return fail_at("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = p[k];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail_at("bad \\u escape");
        }
        p += 4;
        *out = v;
        return true;
    }
    bool str(std::string* o) {
        if (p >= end || *p != '"') return fail_at("expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail_at("bad escape");
                const char c = *p++;
                switch (c) {
                    case '"': *o += '"'; break;
                    case '\\': *o += '\\'; break;
                    case '/': *o += '/'; break;
                    case 'b': *o += '\b'; break;
                    case 'f': *o += '\f'; break;
                    case 'n': *o += '\n'; break;
                    case 'r': *o += '\r'; break;
                    case 't': *o += '\t'; break;
                    case 'u': {
                        uint32_t c1;
                        if (!hex4(&c1)) return false;
                        if (c1 >= 0xD800 && c1 < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2;
                            uint32_t c2;
                            if (!hex4(&c2)) return false;
                            c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00);
                        }
                        utf8(*o, c1);
                        break;
                    }
                    default: return fail_at("bad escape");
                }
            } else {
                *o += *p++;
            }
        }
        if (p >= end) return fail_at("unterminated string");
        p++;
        return true;
    }
    bool num(JVal* v) {
        const char* s0 = p;
        bool integral = true;
        if (p < end && *p == '-') p++;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
            if (*p == '.' || *p == 'e' || *p == 'E') integral = false;
            p++;
        }
        if (p == s0) return fail_at("expected value");
        const std::string tok(s0, p);
        char* e2 = nullptr;
        v->t = JVal::Num;
        v->d = strtod(tok.c_str(), &e2);
        if (!e2 || *e2) return fail_at("bad number");
        if (integral) {
            errno = 0;
            const long long ll = strtoll(tok.c_str(), &e2, 10);
            if (errno == 0 && e2 && !*e2) {
                v->is_int = true;
                v->i = ll;
            }
        }
        return true;
    }
    bool value(JVal* v, int depth) {
        if (depth > 64) return fail_at("nesting too deep");
        ws();
        if (p >= end) return fail_at("unexpected end");
        switch (*p) {
            case '{': {
                v->t = JVal::Obj;
                p++;
                ws();
                if (p < end && *p == '}') { p++; return true; }
                for (;;) {
                    ws();
                    std::string k;
                    if (!str(&k)) return false;
                    ws();
                    if (p >= end || *p != ':') return fail_at("expected ':'");
                    p++;
                    v->obj.emplace_back(std::move(k), JVal());
                    if (!value(&v->obj.back().second, depth + 1)) return false;
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == '}') { p++; return true; }
                    return fail_at("expected ',' or '}'");
                }
            }
            case '[': {
                v->t = JVal::Arr;
                p++;
                ws();
                if (p < end && *p == ']') { p++; return true; }
                bool flat = true;  // numbers only so far: stays in `nums`
                for (;;) {
                    ws();
                    const bool numeric = p < end && (*p == '-' || (*p >= '0' && *p <= '9'));
                    if (flat && numeric) {
                        JVal n;
                        if (!num(&n)) return false;
                        v->nums.push_back(n.d);
                    } else {
                        if (flat) {  // first non-number: move what we have into the general form
                            for (double d : v->nums) {
                                JVal n;
                                n.t = JVal::Num;
                                n.d = d;
                                v->arr.push_back(std::move(n));
                            }
                            v->nums.clear();
                            flat = false;
                        }
                        v->arr.emplace_back();
                        if (!value(&v->arr.back(), depth + 1)) return false;
                    }
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == ']') { p++; return true; }
                    return fail_at("expected ',' or ']'");
                }
            }
            case '"': v->t = JVal::Str; return str(&v->s);
            case 't': v->t = JVal::Bool; v->b = true; return lit("true");
            case 'f': v->t = JVal::Bool; v->b = false; return lit("false");
            case 'n': v->t = JVal::Null; return lit("null");
            default: return num(v);
        }
    }
};

void json_str(std::string& o, const std::string& s) {
    o += '"';
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    snprintf(b, sizeof b, "\\u%04x", c);
                    o += b;
                } else {
                    o += (char)c;
                }
        }
    }
    o += '"';
}
// shortest decimal that reads back as the same value (what serde_json's ryu prints), always with a fraction or exponent;
// non-finite values become null as serde_json writes them
void json_f32(std::string& o, float v) {
    if (!std::isfinite(v)) { o += "null"; return; }
    char b[40];
    for (int prec = 1; prec <= 9; prec++) {
        snprintf(b, sizeof b, "%.*g", prec, (double)v);
        if (strtof(b, nullptr) == v) break;
    }
    o += b;
    if (!strpbrk(b, ".eE")) o += ".0";
}
void json_f64(std::string& o, double v) {
    if (!std::isfinite(v)) { o += "null"; return; }
    char b[48];
    for (int prec = 1; prec <= 17; prec++) {
        snprintf(b, sizeof b, "%.*g", prec, v);
        if (strtod(b, nullptr) == v) break;
    }
    o += b;
    if (!strpbrk(b, ".eE")) o += ".0";
}
const char* metric_name(int32_t m) {
    return m == NMN_METRIC_EUCLIDEAN ? "Euclidean" : m == NMN_METRIC_DOT_PRODUCT ? "DotProduct" : "Cosine";
}
bool metric_from(const std::string& s, int32_t* m) {
    if (s == "Cosine") *m = NMN_METRIC_COSINE;
    else if (s == "Euclidean") *m = NMN_METRIC_EUCLIDEAN;
    else if (s == "DotProduct") *m = NMN_METRIC_DOT_PRODUCT;
    else return false;
    return true;
}
void json_meta_value(std::string& o, const Value& v) {  // MetadataValue, externally tagged (lib.rs:534-546)
    switch (v.kind) {
        case NMN_VAL_BOOL: o += v.b ? "{\"Bool\": true}" : "{\"Bool\": false}"; break;
        case NMN_VAL_INT: o += "{\"Int\": " + std::to_string(v.i) + "}"; break;
        case NMN_VAL_FLOAT: o += "{\"Float\": "; json_f64(o, v.f); o += "}"; break;
        case NMN_VAL_STRING: o += "{\"String\": "; json_str(o, v.s); o += "}"; break;
        default: o += "\"Null\""; break;
    }
}
bool meta_value_from(const JVal& j, Value* out) {
    if (j.t == JVal::Str && j.s == "Null") { out->kind = NMN_VAL_NULL; return true; }
    if (j.t != JVal::Obj || j.obj.size() != 1) return false;
    const std::string& tag = j.obj[0].first;
    const JVal& x = j.obj[0].second;
    if (tag == "Bool" && x.t == JVal::Bool) { out->kind = NMN_VAL_BOOL; out->b = x.b; return true; }
    if (tag == "Int" && x.t == JVal::Num && x.is_int) { out->kind = NMN_VAL_INT; out->i = x.i; return true; }
    if (tag == "Float" && x.t == JVal::Num) { out->kind = NMN_VAL_FLOAT; out->f = x.d; return true; }
    if (tag == "Float" && x.t == JVal::Null) { out->kind = NMN_VAL_FLOAT; out->f = std::nan(""); return true; }
    if (tag == "String" && x.t == JVal::Str) { out->kind = NMN_VAL_STRING; out->s = x.s; return true; }
    return false;
}

// snapshot_collection (lib.rs:3738-3784): the live entries of one collection, in storage order
struct Snapshot {
    std::string collection;
    CollectionConfig config;
    bool has_config = false;
    uint64_t created_at = 0;
    std::vector<const Entry*> entries;
};
Snapshot snapshot_collection(nmn_engine* e, const char* collection) {
    Snapshot s;
    s.collection = collection;
    const bool is_default = s.collection == kDefaultCollection;
    auto cit = e->configs.find(s.collection);
    if (!is_default && cit != e->configs.end()) {  // get_collection_config(..).unwrap_or_default()
        s.config = cit->second;
        s.has_config = true;
    }
    s.created_at = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    Collection* c = is_default ? &e->dflt : e->storage(collection, false);
    if (c)
        for (const Entry& ent : c->slots)
            if (ent.live) s.entries.push_back(&ent);
    return s;
}
void json_header(std::string& o, const Snapshot& s) {
    o += "{\n  \"collection\": ";
    json_str(o, s.collection);
    o += ",\n  \"config\": {\n    \"dimension\": ";
    o += s.config.dimension ? std::to_string(s.config.dimension) : "null";
    o += ",\n    \"distance_metric\": \"";
    o += metric_name(s.config.metric);
    o += "\",\n    \"auto_index\": ";
    o += s.config.auto_index ? "true" : "false";
    o += ",\n    \"auto_index_threshold\": " + std::to_string(s.config.auto_index_threshold) + "\n  },\n";
}
void json_metadata(std::string& o, const Meta& m, const char* indent) {
    if (m.empty()) { o += "null"; return; }  // `if metadata.is_empty() { None }`
    o += "{";
    bool first = true;
    for (auto& kv : m) {
        o += first ? "\n" : ",\n";
        first = false;
        o += indent;
        o += "  ";
        json_str(o, kv.first);
        o += ": ";
        json_meta_value(o, kv.second);
    }
    o += "\n";
    o += indent;
    o += "}";
}

nmn_status write_file(const char* path, const std::string& bytes) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return err_io(std::string("cannot create '") + path + "': " + strerror(errno));
    const bool ok = fwrite(bytes.data(), 1, bytes.size(), fp) == bytes.size();
    if (fclose(fp) != 0 || !ok) return err_io(std::string("cannot write '") + path + "': " + strerror(errno));
    return NMN_OK;
}

// decoded PersistentVectorIndex, before it is restored
struct Decoded {
    std::string collection;
    CollectionConfig config;
    struct Ent {
        std::string key;
        std::vector<float> vec;
        Meta meta;
    };
    std::vector<Ent> entries;
};
bool decode_config(const JVal* cfg, CollectionConfig* out, std::string* why) {
    if (!cfg || cfg->t != JVal::Obj) { *why = "missing field `config`"; return false; }
    const JVal* d = cfg->get("dimension");
    const JVal* m = cfg->get("distance_metric");
    if (!d || !m) { *why = "missing field in `config`"; return false; }
    if (d->t == JVal::Null) out->dimension = 0;
    else if (d->t == JVal::Num && d->is_int && d->i >= 0) out->dimension = (uint64_t)d->i;
    else { *why = "invalid `dimension`"; return false; }
    if (m->t != JVal::Str || !metric_from(m->s, &out->metric)) { *why = "unknown variant of DistanceMetric"; return false; }
    const JVal* ai = cfg->get("auto_index");
    const JVal* at = cfg->get("auto_index_threshold");
    if (!ai || ai->t != JVal::Bool || !at || at->t != JVal::Num || !at->is_int || at->i < 0) { *why = "missing field in `config`"; return false; }
    out->auto_index = ai->b;
    out->auto_index_threshold = (uint64_t)at->i;
    return true;
}
bool decode_metadata(const JVal* mj, Meta* out, std::string* why) {
    if (!mj || mj->t == JVal::Null) return true;
    if (mj->t != JVal::Obj) { *why = "invalid `metadata`"; return false; }
    for (auto& kv : mj->obj) {
        Value v;
        if (!meta_value_from(kv.second, &v)) { *why = "invalid MetadataValue for `" + kv.first + "`"; return false; }
        (*out)[kv.first] = v;
    }
    return true;
}

// restore_from_index (lib.rs:3902-3934): collection config (named collections), then every entry through the ordinary
// store path — the same validation (EmptyVector, collection dimension, max_dimension) and the same overwrite semantics
nmn_status restore_from_index(nmn_engine* e, Decoded& d) {
    const bool is_default = d.collection == kDefaultCollection;
    if (!is_default) e->configs[d.collection] = d.config;
    Collection* c = is_default ? &e->dflt : e->storage(d.collection.c_str(), true);
    for (auto& ent : d.entries) {
        if (ent.vec.empty()) return err_empty();
        if (!is_default && d.config.dimension && ent.vec.size() != d.config.dimension) return err_dim(d.config.dimension, ent.vec.size());
        if (e->cfg.max_dimension && ent.vec.size() > e->cfg.max_dimension) return err_dim(e->cfg.max_dimension, ent.vec.size());
        std::vector<nmn_meta_field> mf;
        mf.reserve(ent.meta.size());
        for (auto& kv : ent.meta) {
            nmn_meta_field f{};
            f.name = kv.first.c_str();
            f.value.kind = kv.second.kind;
            f.value.b = kv.second.b ? 1 : 0;
            f.value.i = kv.second.i;
            f.value.f = kv.second.f;
            f.value.s = kv.second.s.c_str();
            mf.push_back(f);
        }
        nmn_status st = store_into(e, c, ent.key.c_str(), ent.vec.data(), ent.vec.size(), mf.data(), (uint32_t)mf.size());
        if (st != NMN_OK) return st;
    }
    return NMN_OK;
}

nmn_status check_file_limit(nmn_engine* e, const char* path) {  // lib.rs:3831-3840
    const nmn_status st = nmn::persist_check_file_size(path, e->cfg.max_index_file_bytes > 0 ? (uint64_t)e->cfg.max_index_file_bytes : 0, nullptr);
    if (st == NMN_ERR_CONFIGURATION) return err_config(nmn_last_error());
    if (st != NMN_OK) return err_io(nmn_last_error());
    return NMN_OK;
}
nmn_status check_entry_limit(nmn_engine* e, uint64_t n) {  // lib.rs:3847-3856
    if (e->cfg.max_index_entries > 0 && n > (uint64_t)e->cfg.max_index_entries)
        return err_config("index entry count " + std::to_string(n) + " exceeds limit " + std::to_string(e->cfg.max_index_entries));
    return NMN_OK;
}
void copy_name(const std::string& s, char* out, uint64_t cap) {
    if (!out || !cap) return;
    const size_t n = std::min<size_t>(s.size(), cap - 1);
    memcpy(out, s.data(), n);
    out[n] = 0;
}

}  // namespace

extern "C" {

// save_index (lib.rs:3794-3801)
nmn_status nmn_engine_save_index(nmn_engine* e, const char* collection, const char* path) {
    if (!e || !collection || !path) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::string o;
    {
        WriteLock g(e);
        const Snapshot s = snapshot_collection(e, collection);
        json_header(o, s);
        o += "  \"vectors\": [";
        bool first = true;
        for (const Entry* ent : s.entries) {
            o += first ? "\n" : ",\n";
            first = false;
            o += "    {\n      \"key\": ";
            json_str(o, ent->key);
            o += ",\n      \"vector\": [";
            for (size_t i = 0; i < ent->vec.size(); i++) {
                o += i ? ",\n        " : "\n        ";
                json_f32(o, ent->vec[i]);
            }
            o += ent->vec.empty() ? "]" : "\n      ]";
            o += ",\n      \"metadata\": ";
            json_metadata(o, ent->meta, "      ");
            o += "\n    }";
        }
        o += s.entries.empty() ? "]" : "\n  ]";
        o += ",\n  \"created_at\": " + std::to_string(s.created_at) + ",\n  \"version\": " + std::to_string(kPersistVersion) + "\n}";
    }
    return write_file(path, o);
}

// load_index (lib.rs:3827-3866); the collection's name goes to name_out (truncated to cap - 1 bytes)
nmn_status nmn_engine_load_index(nmn_engine* e, const char* path, char* name_out, uint64_t name_cap) {
    if (!e || !path) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    nmn_status st = check_file_limit(e, path);
    if (st != NMN_OK) return st;
    std::string text;
    {
        FILE* fp = fopen(path, "rb");
        if (!fp) return err_io(std::string("cannot open '") + path + "': " + strerror(errno));
        char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
        fclose(fp);
    }
    JVal root;
    JParser jp(text.data(), text.size());
    if (!jp.value(&root, 0)) return err_serialization(jp.err);
    jp.ws();
    if (jp.p != jp.end) return err_serialization("trailing characters");
    if (root.t != JVal::Obj) return err_serialization("invalid type: expected struct PersistentVectorIndex");
    Decoded d;
    const JVal* jc = root.get("collection");
    const JVal* jv = root.get("vectors");
    if (!jc || jc->t != JVal::Str) return err_serialization("missing field `collection`");
    if (!jv || jv->t != JVal::Arr) return err_serialization("missing field `vectors`");
    if (!root.get("created_at") || !root.get("version")) return err_serialization("missing field `created_at` / `version`");
    d.collection = jc->s;
    std::string why;
    if (!decode_config(root.get("config"), &d.config, &why)) return err_serialization(why);
    if (!jv->nums.empty()) return err_serialization("invalid type in `vectors`");
    d.entries.resize(jv->arr.size());
    for (size_t i = 0; i < jv->arr.size(); i++) {
        const JVal& je = jv->arr[i];
        const JVal* k = je.t == JVal::Obj ? je.get("key") : nullptr;
        const JVal* v = je.t == JVal::Obj ? je.get("vector") : nullptr;
        if (!k || k->t != JVal::Str || !v || v->t != JVal::Arr || !v->arr.empty()) return err_serialization("invalid VectorEntry");
        d.entries[i].key = k->s;
        d.entries[i].vec.resize(v->nums.size());
        for (size_t x = 0; x < v->nums.size(); x++) d.entries[i].vec[x] = (float)v->nums[x];  // f64 token -> `as f32`
        if (!decode_metadata(je.get("metadata"), &d.entries[i].meta, &why)) return err_serialization(why);
    }
    st = check_entry_limit(e, d.entries.size());
    if (st != NMN_OK) return st;
    WriteLock g(e);
    st = restore_from_index(e, d);
    if (st == NMN_OK) copy_name(d.collection, name_out, name_cap);
    return st;
}

// save_index_binary (lib.rs:3811-3817), in this library's own format (see the section comment):
// Header{kind = engine, rows = entries, aux = bytes of the text block} | text block = the JSON snapshot with every
// "vector" replaced by its "dim" | one flat shard section per distinct dimension, ascending, rows in entry order
nmn_status nmn_engine_save_index_binary(nmn_engine* e, const char* collection, const char* path) {
    if (!e || !collection || !path) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    WriteLock g(e);
    const Snapshot s = snapshot_collection(e, collection);
    std::string o;
    json_header(o, s);
    o += "  \"vectors\": [";
    std::map<uint64_t, std::vector<const Entry*>> by_dim;
    bool first = true;
    for (const Entry* ent : s.entries) {
        o += first ? "\n" : ",\n";
        first = false;
        o += "    {\"key\": ";
        json_str(o, ent->key);
        o += ", \"dim\": " + std::to_string(ent->vec.size()) + ", \"metadata\": ";
        json_metadata(o, ent->meta, "    ");
        o += "}";
        by_dim[ent->vec.size()].push_back(ent);
    }
    o += s.entries.empty() ? "]" : "\n  ]";
    o += ",\n  \"created_at\": " + std::to_string(s.created_at) + ",\n  \"version\": " + std::to_string(kPersistVersion) + "\n}";
    FILE* fp = fopen(path, "wb");
    if (!fp) return err_io(std::string("cannot create '") + path + "': " + strerror(errno));
    nmn::PersistHeader h{};
    memcpy(h.magic, "NMNIDX\0\1", 8);
    h.version = 1;
    h.kind = nmn::kPersistEngine;
    h.rows = s.entries.size();
    h.aux = o.size();
    h.reserved = by_dim.size();
    nmn_status st = NMN_OK;
    if (fwrite(&h, sizeof h, 1, fp) != 1 || fwrite(o.data(), 1, o.size(), fp) != o.size()) st = err_io(std::string("cannot write '") + path + "'");
    std::vector<float> rows, norms;
    for (auto& kv : by_dim) {
        if (st != NMN_OK) break;
        const uint64_t dim = kv.first, n = kv.second.size();
        if (dim > 0xFFFFFFFFull) { st = fail(NMN_ERR_INVALID_ARGUMENT, "dimension does not fit the index file"); break; }
        rows.resize((size_t)n * dim);
        norms.resize((size_t)n);
        for (uint64_t r = 0; r < n; r++) {
            memcpy(rows.data() + r * dim, kv.second[r]->vec.data(), dim * sizeof(float));
            // simd::magnitude in reference order (hnsw.rs:198-229): the load compares it, bit for bit, with what the GPU
            // computes from the same row — a per-row checksum that doubles as a host/device parity check
            norms[r] = std::sqrt(sumsq8_host(kv.second[r]->vec.data(), dim));
        }
        if (nmn::persist_write_rows_host(fp, path, (uint32_t)dim, n, 0, rows.data(), norms.data()) != NMN_OK) st = err_io(nmn_last_error());
    }
    if (fclose(fp) != 0 && st == NMN_OK) st = err_io(std::string("cannot close '") + path + "'");
    return st;
}

// load_index_binary (lib.rs:3868-3899)
nmn_status nmn_engine_load_index_binary(nmn_engine* e, const char* path, char* name_out, uint64_t name_cap) {
    if (!e || !path) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    nmn_status st = check_file_limit(e, path);
    if (st != NMN_OK) return st;
    FILE* fp = fopen(path, "rb");
    if (!fp) return err_io(std::string("cannot open '") + path + "': " + strerror(errno));
    struct Closer {
        FILE* f;
        ~Closer() { fclose(f); }
    } closer{fp};
    nmn::PersistHeader h{};
    if (nmn::persist_read_header(fp, path, &h) != NMN_OK || h.kind != nmn::kPersistEngine) return err_serialization("not a neumann_gpu collection file");
    st = check_entry_limit(e, h.rows);
    if (st != NMN_OK) return st;
    if (h.aux > (1ull << 40) || h.aux > nmn::persist_bytes_left(fp)) return err_serialization("file truncated (text block)");
    std::string text((size_t)h.aux, '\0');
    if (h.aux && fread(&text[0], 1, (size_t)h.aux, fp) != h.aux) return err_serialization("file truncated (text block)");
    JVal root;
    JParser jp(text.data(), text.size());
    if (!jp.value(&root, 0) || root.t != JVal::Obj) return err_serialization(jp.err.empty() ? "invalid text block" : jp.err);
    Decoded d;
    const JVal* jc = root.get("collection");
    const JVal* jv = root.get("vectors");
    if (!jc || jc->t != JVal::Str || !jv || jv->t != JVal::Arr) return err_serialization("missing field `collection` / `vectors`");
    d.collection = jc->s;
    std::string why;
    if (!decode_config(root.get("config"), &d.config, &why)) return err_serialization(why);
    if (jv->arr.size() != h.rows) return err_serialization("entry count differs from the header");
    d.entries.resize(jv->arr.size());
    std::map<uint64_t, std::vector<size_t>> by_dim;
    for (size_t i = 0; i < jv->arr.size(); i++) {
        const JVal& je = jv->arr[i];
        const JVal* k = je.t == JVal::Obj ? je.get("key") : nullptr;
        const JVal* dj = je.t == JVal::Obj ? je.get("dim") : nullptr;
        if (!k || k->t != JVal::Str || !dj || dj->t != JVal::Num || !dj->is_int || dj->i < 0) return err_serialization("invalid entry");
        d.entries[i].key = k->s;
        if (!decode_metadata(je.get("metadata"), &d.entries[i].meta, &why)) return err_serialization(why);
        by_dim[(uint64_t)dj->i].push_back(i);
    }
    // the matrix sections, ascending dimension: rows go straight into the entries; magnitudes are kept for the check below
    std::map<uint64_t, std::vector<float>> stored_norms;
    std::vector<float> rows;
    for (auto& kv : by_dim) {
        nmn::PersistHeader hs{};
        if (nmn::persist_read_header(fp, path, &hs) != NMN_OK) return err_serialization(nmn_last_error());
        if (hs.dim != kv.first || hs.rows != kv.second.size()) return err_serialization("matrix section does not match the entries");
        if (nmn::persist_read_rows_host(fp, hs, &rows, &stored_norms[kv.first]) != NMN_OK) return err_serialization(nmn_last_error());
        for (size_t r = 0; r < kv.second.size(); r++)
            d.entries[kv.second[r]].vec.assign(rows.begin() + r * kv.first, rows.begin() + (r + 1) * kv.first);
    }
    WriteLock g(e);
    st = restore_from_index(e, d);
    if (st != NMN_OK) return st;
    // Rebuild the GPU mirrors now (a restart should not pay for it on the first query) and check the device layout against
    // the file: the magnitude the GPU computed for every restored row must equal the stored one bit for bit.
    const bool is_default = d.collection == kDefaultCollection;
    Collection* c = is_default ? &e->dflt : e->storage(d.collection.c_str(), false);
    for (auto& kv : by_dim) {
        if (!c || kv.first == 0) continue;
        Mirror* m = nullptr;
        st = get_mirror(e, c, kv.first, &m);
        if (st != NMN_OK) return st;
        if (!m || !m->has_rows()) continue;
        const uint64_t n_rows = m->rows();
        std::vector<float> dev((size_t)n_rows);
        // (device pointer of the magnitudes through the public accessor; one D2H per shard)
        const uint32_t n_sh = m->sh ? nmn_sharded_shards(m->sh) : 1u;
        std::vector<float> part_norms;
        for (uint32_t g = 0; g < n_sh; g++) {
            const nmn_index* part = m->sh ? nmn_sharded_shard(m->sh, g) : m->idx;
            const uint64_t cnt = nmn_index_rows(part);
            if (cnt == 0) continue;
            part_norms.resize((size_t)cnt);
            if (hipMemcpy(part_norms.data(), nmn_index_norms_device(part), (size_t)cnt * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(NMN_ERR_STORAGE, "Storage error: reading the magnitudes back");
            for (uint64_t l = 0; l < cnt; l++) {  // the shard's local rows back to mirror rows (ranges or 64-row blocks dealt round-robin)
                const uint64_t row = m->sh ? nmn_sharded_global_row(m->sh, g, l) : l;
                if (row >= n_rows) return fail(NMN_ERR_STORAGE, "Storage error: shard rows beyond the mirror");
                dev[(size_t)row] = part_norms[(size_t)l];
            }
        }
        const std::vector<float>& want = stored_norms[kv.first];
        for (size_t r = 0; r < kv.second.size(); r++) {
            auto it = c->by_key.find(d.entries[kv.second[r]].key);
            if (it == c->by_key.end()) continue;
            const Entry& ent = c->slots[it->second];
            if (ent.mrow < 0 || (uint64_t)ent.mrow >= n_rows || ent.vec.size() != kv.first) continue;  // overwritten by a later duplicate key
            uint32_t a, b;
            memcpy(&a, &dev[(size_t)ent.mrow], 4);
            memcpy(&b, &want[r], 4);
            if (a != b && d.entries[kv.second[r]].vec == ent.vec)
                return err_serialization("index file corrupt: magnitude of '" + ent.key + "' differs from the stored one");
        }
    }
    copy_name(d.collection, name_out, name_cap);
    return NMN_OK;
}

// save_all_indices (lib.rs:3944-3971): default.json + one {collection}.json per non-empty collection
nmn_strlist* nmn_engine_save_all_indices(nmn_engine* e, const char* dir, nmn_status* status) {
    nmn_status dummy;
    if (!status) status = &dummy;
    *status = NMN_OK;
    if (!e || !dir) { *status = fail(NMN_ERR_INVALID_ARGUMENT, "null argument"); return nullptr; }
    std::string cmd_dir = dir;
    // fs::create_dir_all
    for (size_t i = 1; i <= cmd_dir.size(); i++)
        if (i == cmd_dir.size() || cmd_dir[i] == '/') {
            const std::string part = cmd_dir.substr(0, i);
            if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) { *status = err_io("cannot create '" + part + "': " + strerror(errno)); return nullptr; }
        }
    auto* out = new (std::nothrow) nmn_strlist();
    if (!out) { *status = fail(NMN_ERR_OUT_OF_MEMORY, "list alloc"); return nullptr; }
    std::vector<std::string> names;
    {
        WriteLock g(e);
        if (e->dflt.live > 0) names.push_back(kDefaultCollection);
        for (auto& kv : e->configs) {  // list_collections(): the configured ones
            Collection* c = e->storage(kv.first.c_str(), false);
            if (c && c->live > 0) names.push_back(kv.first);
        }
    }
    for (auto& n : names) {
        const std::string path = cmd_dir + "/" + n + ".json";
        *status = nmn_engine_save_index(e, n.c_str(), path.c_str());
        if (*status != NMN_OK) { delete out; return nullptr; }
        out->items.push_back(n);
    }
    return out;
}

// load_all_indices (lib.rs:3980-3999): every *.json of the directory; a file that fails to load is skipped
nmn_strlist* nmn_engine_load_all_indices(nmn_engine* e, const char* dir, nmn_status* status) {
    nmn_status dummy;
    if (!status) status = &dummy;
    *status = NMN_OK;
    if (!e || !dir) { *status = fail(NMN_ERR_INVALID_ARGUMENT, "null argument"); return nullptr; }
    DIR* dp = opendir(dir);
    if (!dp) { *status = err_io(std::string("cannot read '") + dir + "': " + strerror(errno)); return nullptr; }
    std::vector<std::string> files;
    while (dirent* de = readdir(dp)) {
        const std::string f = de->d_name;
        if (f.size() > 5 && f.compare(f.size() - 5, 5, ".json") == 0) files.push_back(f);
    }
    closedir(dp);
    std::sort(files.begin(), files.end());
    auto* out = new (std::nothrow) nmn_strlist();
    if (!out) { *status = fail(NMN_ERR_OUT_OF_MEMORY, "list alloc"); return nullptr; }
    for (auto& f : files) {
        char name[1024];
        if (nmn_engine_load_index(e, (std::string(dir) + "/" + f).c_str(), name, sizeof name) == NMN_OK) out->items.push_back(name);
    }
    return out;
}

// (IVFIndex, key_mapping) of build_ivf_index, persisted: Header{kind = engine, aux = text bytes} | {"nprobe", "dim", "keys"}
// | the IVF section (centroids, lists, vectors) — a restart restores the trained index without k-means
nmn_status nmn_engine_ivf_save(nmn_engine_ivf* ivf, const char* path) {
    if (!ivf || !path) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::string o = "{\"nprobe\": " + std::to_string(ivf->nprobe) + ", \"dim\": " + std::to_string(ivf->dim) + ", \"trained\": " +
                    (ivf->index ? "true" : "false") + ", \"keys\": [";
    for (size_t i = 0; i < ivf->keys.size(); i++) {
        if (i) o += ", ";
        json_str(o, ivf->keys[i]);
    }
    o += "]}";
    FILE* fp = fopen(path, "wb");
    if (!fp) return err_io(std::string("cannot create '") + path + "': " + strerror(errno));
    nmn::PersistHeader h{};
    memcpy(h.magic, "NMNIDX\0\1", 8);
    h.version = 1;
    h.kind = nmn::kPersistEngine;
    h.rows = ivf->keys.size();
    h.aux = o.size();
    h.flags = 1;  // an IVF index follows
    nmn_status st = NMN_OK;
    if (fwrite(&h, sizeof h, 1, fp) != 1 || fwrite(o.data(), 1, o.size(), fp) != o.size()) st = err_io(std::string("cannot write '") + path + "'");
    if (st == NMN_OK && ivf->index && nmn::persist_write_ivf(ivf->index, fp, path) != NMN_OK) st = err_gpu(NMN_ERR_STORAGE);
    if (fclose(fp) != 0 && st == NMN_OK) st = err_io(std::string("cannot close '") + path + "'");
    return st;
}

nmn_status nmn_engine_ivf_load(nmn_engine* e, const char* path, nmn_engine_ivf** out) {
    if (!e || !path || !out) return fail(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_status st = check_file_limit(e, path);
    if (st != NMN_OK) return st;
    FILE* fp = fopen(path, "rb");
    if (!fp) return err_io(std::string("cannot open '") + path + "': " + strerror(errno));
    struct Closer {
        FILE* f;
        ~Closer() { fclose(f); }
    } closer{fp};
    nmn::PersistHeader h{};
    if (nmn::persist_read_header(fp, path, &h) != NMN_OK || h.kind != nmn::kPersistEngine || h.flags != 1)
        return err_serialization("not a neumann_gpu IVF index file");
    st = check_entry_limit(e, h.rows);
    if (st != NMN_OK) return st;
    if (h.aux > (1ull << 40) || h.aux > nmn::persist_bytes_left(fp)) return err_serialization("file truncated (text block)");
    std::string text((size_t)h.aux, '\0');
    if (h.aux && fread(&text[0], 1, (size_t)h.aux, fp) != h.aux) return err_serialization("file truncated (text block)");
    JVal root;
    JParser jp(text.data(), text.size());
    if (!jp.value(&root, 0) || root.t != JVal::Obj) return err_serialization(jp.err.empty() ? "invalid text block" : jp.err);
    const JVal* jn = root.get("nprobe");
    const JVal* jd = root.get("dim");
    const JVal* jt = root.get("trained");
    const JVal* jk = root.get("keys");
    if (!jn || !jn->is_int || !jd || !jd->is_int || !jt || jt->t != JVal::Bool || !jk || jk->t != JVal::Arr || !jk->nums.empty())
        return err_serialization("invalid text block");
    auto res = std::make_unique<nmn_engine_ivf>();
    res->nprobe = (uint64_t)jn->i;
    res->dim = (uint64_t)jd->i;
    for (auto& k : jk->arr) {
        if (k.t != JVal::Str) return err_serialization("invalid key");
        res->keys.push_back(k.s);
    }
    if (res->keys.size() != h.rows) return err_serialization("key count differs from the header");
    if (jt->b) {
        nmn::PersistHeader hi{};
        if (nmn::persist_read_header(fp, path, &hi) != NMN_OK) return err_serialization(nmn_last_error());
        nmn_index_desc d{};
        d.device = e->cfg.device;
        d.cand_cap = e->cfg.cand_cap;
        if (nmn::persist_read_ivf(fp, path, hi, &d, &res->index) != NMN_OK) return err_gpu(NMN_ERR_STORAGE);
        if (nmn_ivf_len(res->index) != res->keys.size() || hi.dim != res->dim) return err_serialization("IVF section does not match the keys");
        res->n_clusters = nmn_ivf_clusters(res->index);
        res->centroids.resize((size_t)res->n_clusters * res->dim);
        if (nmn_ivf_centroids(res->index, res->centroids.data(), res->centroids.size()) != NMN_OK) return err_gpu(NMN_ERR_STORAGE);
    }
    *out = res.release();
    return NMN_OK;
}

}  // extern "C"
