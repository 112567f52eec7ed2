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
    nmn_index* idx = nullptr;
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

struct CollectionConfig {
    uint64_t dimension = 0;  // 0 = None
    int32_t metric = NMN_METRIC_COSINE;
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
    if (!it->second->idx) {  // built when no row of this dimension existed: nothing to patch, rebuild lazily
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
    if (m->dirty.empty() || !m->idx) {
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
        if (nmn_index_upload(m->idx, buf.data(), m->dirty[i], j - i) != NMN_OK) return false;
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
        nmn_status st = nmn_index_create(&d, &m->idx);
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
            nmn_status s2 = nmn_index_upload(m->idx, buf.data(), row0, cnt);
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

nmn_status gpu_topk(Collection* c, Mirror* m, const float* q, uint64_t top_k, int32_t metric,
                    const DeviceSelection* selected, nmn_results* res) {
    if (!m->idx) return NMN_OK;  // no rows of this dimension
    const uint64_t rows = nmn_index_rows(m->idx);
    const uint64_t taking_part = selected ? selected->count : rows - std::min<uint64_t>(m->n_dead, rows);
    uint64_t k = std::min<uint64_t>(top_k, taking_part);
    if (k == 0) return NMN_OK;
    std::vector<uint64_t> out_rows(k);
    std::vector<float> out_scores(k);
    uint32_t count = 0;
    nmn_status st;
    if (selected) {
        st = nmn_index_search_dmask_hint(m->idx, q, 1, (uint32_t)k, (nmn_metric)metric, selected->mask_dev, selected->count,
                                         out_rows.data(), out_scores.data(), &count, nullptr);
    } else {
        // deleted rows stay in the matrix until the next rebuild: the live bitmap keeps them out of every scan
        st = nmn_index_search(m->idx, q, 1, (uint32_t)k, (nmn_metric)metric, m->n_dead ? m->live.data() : nullptr,
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
        if (!m->dirty.empty() || (m->idx && !m->cols)) return kRetryExclusive;
    } else {
        nmn_status st = get_mirror(e, c, dim, &m);
        if (st != NMN_OK) return st;
    }
    if (!m->idx) return NMN_OK;
    nmn_status st = shared ? NMN_OK : columns_build(e, c, m);
    if (st != NMN_OK) return st;
    Program p;
    Code code = compile_filter(f, *m, p);
    p.ops = std::move(code.ops);
    // predicate and search in one call (nmn_index_search_pred): concurrent filtered searches then wait for ONE batch —
    // their predicates are evaluated together on the batch's stream, right before the sweep that serves them all
    if (dl.expired()) return err_timeout(op, dl.ms);  // lib.rs:2005-2010
    const uint64_t rows = nmn_index_rows(m->idx);
    const uint64_t k = std::min<uint64_t>(std::min<uint64_t>(top_k, rows), NMN_MAX_TOP_K);
    if (k == 0 || rows != m->row_to_slot.size()) {
        if (k == 0) return NMN_OK;
        return fail(NMN_ERR_STORAGE, "mirror and metadata columns disagree on the row count");
    }
    if (top_k > NMN_MAX_TOP_K) {
        // beyond the candidate pipeline: evaluate, then the large-k path over the bitmap (the two-step form)
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
    if (st != NMN_OK || !m->idx) return st;
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

nmn_strlist* nmn_engine_list_keys(nmn_engine* e) {
    nmn_strlist* l = new (std::nothrow) nmn_strlist();
    if (!e || !l) return l;
    WriteLock g(e);
    for (const auto& ent : e->dflt.slots)
        if (ent.live) l->items.push_back(ent.key);
    return l;
}

nmn_status nmn_engine_clear(nmn_engine* e, uint64_t* removed) {
    if (!e) return fail(NMN_ERR_INVALID_ARGUMENT, "null engine");
    WriteLock g(e);
    if (removed) *removed = e->dflt.live;
    e->dflt = Collection();
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
    for (const auto& ent : e->entities.slots)
        if (ent.live) l->items.push_back(ent.key);
    return l;
}

uint64_t nmn_engine_count_entities_with_embeddings(nmn_engine* e) {  // lib.rs:3235-3237
    if (!e) return 0;
    WriteLock g(e);
    return e->entities.live;
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
        st = locked_search(e, [&] { return &e->entities; }, q, dim, top_k, NMN_METRIC_COSINE, "search_entities", dl, res);
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
    e->configs[name] = CollectionConfig{dimension, metric};
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

int32_t nmn_engine_mirror_cached(nmn_engine* e, const char* coll) {
    if (!e) return 0;
    WriteLock g(e);
    Collection* c = e->storage(coll, false);
    return (c && !c->mirrors.empty()) ? 1 : 0;
}

}  // extern "C"
