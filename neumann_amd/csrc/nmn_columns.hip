// nmn_columns.hip — columnar metadata in HBM and the WHERE-predicate kernel.
//
// The reference evaluates a FilterCondition per stored key on the host: `store.get` (BTreeMap
// lookup + clone of the whole TensorData) then a recursive `evaluate_filter`
// (vector_engine/src/lib.rs:3582-3630) with the typed comparison of
// `compare_tensor_value_to_filter` (lib.rs:3648-3670).  Here each metadata field is one column of
// (kind u8, payload u64) cells aligned with the rows of the mirrored index, the condition is a
// postfix program, and ONE kernel evaluates it for every row: a wavefront owns 64 consecutive rows,
// each lane walks the (wave-uniform) program over its row's cells, and `__ballot` turns the 64
// verdicts into one word of the selection bitmap that the masked scan consumes unchanged.
//
// HBM-bound integer/byte work: per leaf the kernel reads 1 B (kind) [+ 8 B payload] per row, fully
// coalesced (64 B / 512 B per wave per column); the bitmap write is 1 bit per row.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <memory>
#include <condition_variable>
#include <new>
#include <vector>

#include "nmn_internal.h"

using namespace nmn;

namespace {

struct DevOp {  // one program step as the kernel reads it (column ids already resolved to pointers)
    uint32_t op, cmp, vkind, pad;
    const uint8_t* kinds;
    const uint64_t* payload;
    uint64_t a, b;
};

// compare_tensor_value_to_filter (lib.rs:3648-3670) followed by the operator's test of the
// ordering (lib.rs:3603-3620).  `ordering.is_some_and(cmp)`: an incomparable pair is false for
// every operator, Ne included.
__device__ __forceinline__ bool cmp_cell(uint32_t ck, uint64_t cp, uint32_t vk, uint64_t vp, uint32_t cmp) {
    int ord;
    if (ck == NMN_CELL_INT && vk == NMN_CELL_INT) {  // a.cmp(b)
        const long long a = (long long)cp, b = (long long)vp;
        ord = a < b ? -1 : (a > b ? 1 : 0);
    } else if ((ck == NMN_CELL_FLOAT || ck == NMN_CELL_INT) && (vk == NMN_CELL_FLOAT || vk == NMN_CELL_INT)) {
        // Float/Float, Float/Int (`*b as f64`), Int/Float (`*a as f64`): partial_cmp on f64
        const double a = ck == NMN_CELL_FLOAT ? __longlong_as_double((long long)cp) : (double)(long long)cp;
        const double b = vk == NMN_CELL_FLOAT ? __longlong_as_double((long long)vp) : (double)(long long)vp;
        if (a != a || b != b) return false;  // partial_cmp -> None
        ord = a < b ? -1 : (a > b ? 1 : 0);
    } else if (ck == NMN_CELL_BOOL && vk == NMN_CELL_BOOL) {
        const int a = (int)(cp & 1ull), b = (int)(vp & 1ull);
        ord = a - b;
    } else if (ck == NMN_CELL_NULL && vk == NMN_CELL_NULL) {
        ord = 0;
    } else {
        return false;  // incompatible types (strings arrive as NMN_PRED_STRSET, never here)
    }
    switch (cmp) {
        case NMN_CMP_EQ: return ord == 0;
        case NMN_CMP_NE: return ord != 0;
        case NMN_CMP_LT: return ord < 0;
        case NMN_CMP_LE: return ord <= 0;
        case NMN_CMP_GT: return ord > 0;
        default: return ord >= 0;
    }
}

// U consecutive bitmap words (U x 64 rows) per wave and trip: every leaf issues its U kind loads and U
// payload loads back to back and unconditionally (rows past the end are clamped, then masked), so a
// wave keeps 2U coalesced requests in flight instead of one dependent load after another — the kernel
// is a pure latency chain otherwise (program step -> kind -> payload -> validity word -> store).
constexpr int kPredUnroll = 4;
constexpr uint32_t kPredMaxBlocks = 4096;

__device__ __forceinline__ void pred_eval_body(const DevOp* __restrict__ ops, uint32_t n_ops,
                                               const uint64_t* __restrict__ consts,
                                               const uint64_t* __restrict__ valid, uint64_t n_rows,
                                               uint64_t* __restrict__ mask,
                                               unsigned long long* __restrict__ partial) {
    constexpr int U = kPredUnroll;
    __shared__ unsigned long long wave_sel[4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_words = (n_rows + 63) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long selected = 0;
    for (uint64_t w0 = wave * U; w0 < n_words; w0 += n_waves * U) {
        uint64_t row[U], vld[U], stack[U];  // stack[u]: bit i = slot i (depth <= 64, validated on the host)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = ((w0 + u) << 6) + lane;
            row[u] = r < n_rows ? r : n_rows - 1;
            vld[u] = (w0 + u) < n_words ? valid[w0 + u] : 0ull;
            stack[u] = 0;
        }
        uint32_t sp = 0;
        for (uint32_t i = 0; i < n_ops; i++) {
            const DevOp o = ops[i];  // wave-uniform: scalar loads, scalar branches
            bool r[U];
#pragma unroll
            for (int u = 0; u < U; u++) r[u] = false;
            switch (o.op) {
                case NMN_PRED_TRUE:
#pragma unroll
                    for (int u = 0; u < U; u++) r[u] = true;
                    break;
                case NMN_PRED_AND:
                case NMN_PRED_OR:
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const bool y = (stack[u] >> (sp - 1)) & 1ull, x = (stack[u] >> (sp - 2)) & 1ull;
                        r[u] = o.op == NMN_PRED_AND ? (x && y) : (x || y);
                    }
                    sp -= 2;
                    break;
                case NMN_PRED_EXISTS:
#pragma unroll
                    for (int u = 0; u < U; u++) r[u] = o.kinds[row[u]] != NMN_CELL_ABSENT;
                    break;
                case NMN_PRED_CMP:
                case NMN_PRED_IN:
                case NMN_PRED_STRSET: {
                    uint32_t ck[U];
                    uint64_t cp[U];
#pragma unroll
                    for (int u = 0; u < U; u++) ck[u] = o.kinds[row[u]];
#pragma unroll
                    for (int u = 0; u < U; u++) cp[u] = o.payload[row[u]];
                    if (o.op == NMN_PRED_CMP) {
#pragma unroll
                        for (int u = 0; u < U; u++) r[u] = ck[u] != NMN_CELL_ABSENT && cmp_cell(ck[u], cp[u], o.vkind, o.a, o.cmp);
                    } else if (o.op == NMN_PRED_IN) {
                        for (uint64_t j = 0; j < o.b; j++) {
                            const uint32_t vk = (uint32_t)consts[o.a + 2 * j];
                            const uint64_t vp = consts[o.a + 2 * j + 1];
#pragma unroll
                            for (int u = 0; u < U; u++)
                                r[u] = r[u] || (ck[u] != NMN_CELL_ABSENT && cmp_cell(ck[u], cp[u], vk, vp, NMN_CMP_EQ));
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; u++)
                            if (ck[u] == NMN_CELL_STRING && cp[u] < o.b)
                                r[u] = (consts[o.a + (cp[u] >> 6)] >> (cp[u] & 63ull)) & 1ull;
                    }
                    break;
                }
                default: break;  // NMN_PRED_FALSE
            }
#pragma unroll
            for (int u = 0; u < U; u++) stack[u] = (stack[u] & ~(1ull << sp)) | ((uint64_t)r[u] << sp);
            sp++;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool in_range = (((w0 + u) << 6) + lane) < n_rows;
            const uint64_t word = __ballot(in_range && (stack[u] & 1ull)) & vld[u];
            if (lane == 0 && (w0 + u) < n_words) {
                mask[w0 + u] = word;
                selected += (unsigned long long)__popcll(word);
            }
        }
    }
    // per-block partial counts, summed by count_reduce_kernel: one L2 atomic per wave on a single
    // address costs ~10 ns each and would dominate the kernel (16k waves -> ~170 us measured)
    if (lane == 0) wave_sel[threadIdx.x >> 6] = selected;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wave_sel[0] + wave_sel[1] + wave_sel[2] + wave_sel[3];
}

__global__ __launch_bounds__(256) void pred_eval_kernel(const DevOp* __restrict__ ops, uint32_t n_ops,
                                                        const uint64_t* __restrict__ consts,
                                                        const uint64_t* __restrict__ valid, uint64_t n_rows,
                                                        uint64_t* __restrict__ mask,
                                                        unsigned long long* __restrict__ partial) {
    pred_eval_body(ops, n_ops, consts, valid, n_rows, mask, partial);
}

// Several programs in one launch (blockIdx.y = program): the predicates of the filtered searches that ride one query
// batch (nmn_index_search_pred) are evaluated together, on the batch's stream, right before its sweep.
struct ProgDesc {
    uint32_t ops_off, n_ops;     // byte offset of the DevOps in the staged block
    uint32_t consts_off, pad;    // byte offset of the constants
    uint64_t* mask;              // result bitmap of this program
    unsigned long long* counts;  // [0] total, [1 ..] per-block partials
    unsigned long long* host_total;  // nullable: the total ALSO goes here — pinned host memory the caller reads without a copy (round 6)
    uint32_t* ticket;            // nullable (zero between launches): the LAST block of the program to finish sums the partials itself —
                                 // no count_reduce launch behind the evaluation (round 6)
};
__global__ __launch_bounds__(256) void pred_eval_batch_kernel(const uint8_t* __restrict__ base,
                                                              const uint64_t* __restrict__ valid, uint64_t n_rows) {
    const ProgDesc d = reinterpret_cast<const ProgDesc*>(base)[blockIdx.y];
    pred_eval_body(reinterpret_cast<const DevOp*>(base + d.ops_off), d.n_ops,
                   reinterpret_cast<const uint64_t*>(base + d.consts_off), valid, n_rows, d.mask, d.counts + 1);
    if (!d.ticket) return;  // (block-uniform)
    __shared__ uint32_t s_last;
    __shared__ unsigned long long acc[256];
    if (threadIdx.x == 0) {
        __threadfence();  // (this thread wrote the block's partial; a fence in all 256 threads of 2 400 blocks cost 0.3 ms)
        const uint32_t t = atomicAdd(d.ticket, 1u);
        s_last = t == gridDim.x - 1u ? 1u : 0u;
        if (s_last) *d.ticket = 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();  // acquire: the other blocks' partials
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256) s += __builtin_nontemporal_load(d.counts + 1 + i);
    acc[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {
        if (threadIdx.x < step) acc[threadIdx.x] += acc[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.counts[0] = acc[0];
        if (d.host_total) *d.host_total = acc[0];
    }
}
__global__ __launch_bounds__(256) void count_reduce_batch_kernel(const uint8_t* __restrict__ base, uint32_t n_blocks) {
    __shared__ unsigned long long acc[256];
    const ProgDesc d = reinterpret_cast<const ProgDesc*>(base)[blockIdx.x];
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < n_blocks; i += 256) s += d.counts[1 + i];
    acc[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {
        if (threadIdx.x < step) acc[threadIdx.x] += acc[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        d.counts[0] = acc[0];
        if (d.host_total) *d.host_total = acc[0];
    }
}

__global__ __launch_bounds__(256) void count_reduce_kernel(const unsigned long long* __restrict__ partial, uint32_t n,
                                                           unsigned long long* __restrict__ count) {
    __shared__ unsigned long long acc[256];
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) s += partial[i];
    acc[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t step = 128; step > 0; step >>= 1) {
        if (threadIdx.x < step) acc[threadIdx.x] += acc[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = acc[0];
}

__global__ void clear_row_kernel(uint8_t* const* kinds, uint32_t n_cols, uint64_t row) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cols) kinds[c][row] = NMN_CELL_ABSENT;
}

struct Column {
    uint8_t* kinds = nullptr;
    uint64_t* payload = nullptr;
};

}  // namespace

struct nmn_columns {
    int device = 0;
    uint64_t cap = 0, words = 0;
    std::vector<Column> cols;
    uint64_t* valid = nullptr;
    uint64_t* mask = nullptr;
    unsigned long long* count = nullptr;    // [0] total, [1 ..] per-block partial counts
    uint8_t* prog = nullptr;  // device staging: [DevOp x n_ops | consts]
    size_t prog_cap = 0;
    uint8_t** kinds_tab = nullptr;  // device table of the columns' kind arrays (clear_row)
    size_t kinds_tab_cap = 0;
    bool kinds_tab_dirty = true;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // Concurrent evaluations (nmn_columns_eval_acquire): each holds a slot — its own stream, program staging, counters
    // and RESULT BITMAP — until the search that consumes the bitmap is done.  Slots are created on demand.
    struct EvalSlot {
        hipStream_t stream = nullptr;
        uint8_t* prog = nullptr;
        size_t prog_cap = 0;
        unsigned long long* count = nullptr;
        uint64_t* mask = nullptr;
        uint8_t* pin = nullptr;   // pinned host staging: [count readback (8 B) | program + constants]
        size_t pin_cap = 0;
        bool busy = false;
    };
    std::vector<std::unique_ptr<EvalSlot>> slots;
    std::condition_variable slot_cv;
    static constexpr size_t kMaxSlots = 256;
};

#define COL_TRY(expr)                                              \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return nmn::set_error_hip(_e, #expr); \
    } while (0)

extern "C" nmn_status nmn_columns_create(int32_t device, uint64_t capacity_rows, nmn_columns** out) {
    if (!out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (capacity_rows == 0) return set_error(NMN_ERR_INVALID_ARGUMENT, "capacity_rows == 0");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        (void)hipGetLastError();
        return set_error(NMN_ERR_NO_DEVICE, "no HIP device");
    }
    int dev = device;
    if (dev < 0) COL_TRY(hipGetDevice(&dev));
    if (dev >= n_dev) return set_error(NMN_ERR_NO_DEVICE, "device ordinal out of range");
    COL_TRY(hipSetDevice(dev));
    nmn_columns* c = new (std::nothrow) nmn_columns();
    if (!c) return set_error(NMN_ERR_OUT_OF_MEMORY, "host alloc");
    c->device = dev;
    c->cap = capacity_rows;
    c->words = (capacity_rows + 63) / 64;
    auto bail = [&](hipError_t e, const char* what) {
        nmn_status st = set_error_hip(e, what);
        nmn_columns_destroy(c);
        return st;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->valid), c->words * 8)) != hipSuccess) return bail(e, "hipMalloc(valid)");
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->mask), c->words * 8)) != hipSuccess) return bail(e, "hipMalloc(mask)");
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->count), 8 * (1 + kPredMaxBlocks))) != hipSuccess) return bail(e, "hipMalloc(count)");
    if ((e = hipMemsetAsync(c->valid, 0, c->words * 8, c->stream)) != hipSuccess) return bail(e, "hipMemset(valid)");
    if ((e = hipMemsetAsync(c->mask, 0, c->words * 8, c->stream)) != hipSuccess) return bail(e, "hipMemset(mask)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = c;
    return NMN_OK;
}

extern "C" nmn_status nmn_columns_destroy(nmn_columns* c) {
    if (!c) return NMN_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& col : c->cols) {
        if (col.kinds) (void)hipFree(col.kinds);
        if (col.payload) (void)hipFree(col.payload);
    }
    for (void* p : {(void*)c->valid, (void*)c->mask, (void*)c->count, (void*)c->prog, (void*)c->kinds_tab})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (auto& sl : c->slots) {
        if (sl->stream) {
            (void)hipStreamSynchronize(sl->stream);
            (void)hipStreamDestroy(sl->stream);
        }
        for (void* p : {(void*)sl->prog, (void*)sl->count, (void*)sl->mask})
            if (p) (void)hipFree(p);
        if (sl->pin) (void)hipHostFree(sl->pin);
    }
    delete c;
    return NMN_OK;
}

extern "C" nmn_status nmn_columns_add(nmn_columns* c, uint32_t* column_out) {
    if (!c || !column_out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    COL_TRY(hipSetDevice(c->device));
    Column col;
    COL_TRY(hipMalloc(reinterpret_cast<void**>(&col.kinds), c->cap));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&col.payload), c->cap * 8);
    if (e == hipSuccess) e = hipMemsetAsync(col.kinds, 0, c->cap, c->stream);  // all cells ABSENT
    if (e == hipSuccess) e = hipMemsetAsync(col.payload, 0, c->cap * 8, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        (void)hipFree(col.kinds);
        if (col.payload) (void)hipFree(col.payload);
        return set_error_hip(e, "nmn_columns_add");
    }
    *column_out = (uint32_t)c->cols.size();
    c->cols.push_back(col);
    c->kinds_tab_dirty = true;
    return NMN_OK;
}

extern "C" uint32_t nmn_columns_count(const nmn_columns* c) { return c ? (uint32_t)c->cols.size() : 0; }

extern "C" nmn_status nmn_columns_write(nmn_columns* c, uint32_t column, uint64_t row0, uint64_t n,
                                        const uint8_t* kinds, const uint64_t* payloads) {
    if (!c || (n && (!kinds || !payloads))) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (column >= c->cols.size()) return set_error(NMN_ERR_INVALID_ARGUMENT, "no such column");
    if (row0 > c->cap || n > c->cap - row0) return set_error(NMN_ERR_CAPACITY, "cells beyond capacity_rows");
    if (n == 0) return NMN_OK;
    for (uint64_t i = 0; i < n; i++)
        if (kinds[i] > NMN_CELL_STRING) return set_error(NMN_ERR_INVALID_ARGUMENT, "bad cell kind");
    COL_TRY(hipSetDevice(c->device));
    COL_TRY(hipMemcpyAsync(c->cols[column].kinds + row0, kinds, n, hipMemcpyHostToDevice, c->stream));
    COL_TRY(hipMemcpyAsync(c->cols[column].payload + row0, payloads, n * 8, hipMemcpyHostToDevice, c->stream));
    COL_TRY(hipStreamSynchronize(c->stream));
    return NMN_OK;
}

static nmn_status sync_kinds_tab(nmn_columns* c) {
    if (!c->kinds_tab_dirty) return NMN_OK;
    const size_t n = c->cols.size();
    if (n > c->kinds_tab_cap) {
        if (c->kinds_tab) COL_TRY(hipFree(c->kinds_tab));
        c->kinds_tab = nullptr;
        c->kinds_tab_cap = 0;
        COL_TRY(hipMalloc(reinterpret_cast<void**>(&c->kinds_tab), (n + 16) * sizeof(uint8_t*)));
        c->kinds_tab_cap = n + 16;
    }
    std::vector<uint8_t*> tab(n);
    for (size_t i = 0; i < n; i++) tab[i] = c->cols[i].kinds;
    if (n) COL_TRY(hipMemcpyAsync(c->kinds_tab, tab.data(), n * sizeof(uint8_t*), hipMemcpyHostToDevice, c->stream));
    COL_TRY(hipStreamSynchronize(c->stream));  // `tab` is a stack temporary
    c->kinds_tab_dirty = false;
    return NMN_OK;
}

extern "C" nmn_status nmn_columns_clear_row(nmn_columns* c, uint64_t row) {
    if (!c) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (row >= c->cap) return set_error(NMN_ERR_CAPACITY, "row beyond capacity_rows");
    if (c->cols.empty()) return NMN_OK;
    COL_TRY(hipSetDevice(c->device));
    nmn_status st = sync_kinds_tab(c);
    if (st != NMN_OK) return st;
    const uint32_t n = (uint32_t)c->cols.size();
    hipLaunchKernelGGL(clear_row_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->kinds_tab, n, row);
    COL_TRY(hipGetLastError());
    COL_TRY(hipStreamSynchronize(c->stream));
    return NMN_OK;
}

extern "C" nmn_status nmn_columns_write_valid(nmn_columns* c, uint64_t word0, uint64_t n_words, const uint64_t* words) {
    if (!c || (n_words && !words)) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (word0 > c->words || n_words > c->words - word0) return set_error(NMN_ERR_CAPACITY, "words beyond capacity");
    if (n_words == 0) return NMN_OK;
    COL_TRY(hipSetDevice(c->device));
    COL_TRY(hipMemcpyAsync(c->valid + word0, words, n_words * 8, hipMemcpyHostToDevice, c->stream));
    COL_TRY(hipStreamSynchronize(c->stream));
    return NMN_OK;
}

// validate a program on the host (column ids, constant ranges, stack discipline) and resolve its column pointers
static nmn_status pred_compile(const nmn_columns* c, const nmn_pred_op* prog, uint32_t n_ops, uint64_t n_consts,
                               std::vector<DevOp>* out) {
    std::vector<DevOp>& dops = *out;
    dops.resize(n_ops);
    int64_t depth = 0;
    for (uint32_t i = 0; i < n_ops; i++) {
        const nmn_pred_op& o = prog[i];
        DevOp d{};
        d.op = o.op;
        d.cmp = o.cmp;
        d.vkind = o.vkind;
        d.a = o.a;
        d.b = o.b;
        const bool leaf_col = o.op == NMN_PRED_EXISTS || o.op == NMN_PRED_CMP || o.op == NMN_PRED_IN || o.op == NMN_PRED_STRSET;
        if (leaf_col) {
            if (o.column >= c->cols.size()) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: no such column");
            d.kinds = c->cols[o.column].kinds;
            d.payload = c->cols[o.column].payload;
        }
        switch (o.op) {
            case NMN_PRED_TRUE: case NMN_PRED_FALSE: case NMN_PRED_EXISTS: depth++; break;
            case NMN_PRED_CMP:
                if (o.cmp > NMN_CMP_GE || o.vkind > NMN_CELL_STRING) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: bad CMP");
                depth++;
                break;
            case NMN_PRED_IN:
                if (o.a > n_consts || o.b > (n_consts - o.a) / 2) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: IN list out of range");
                depth++;
                break;
            case NMN_PRED_STRSET:
                if (o.a > n_consts || (o.b + 63) / 64 > n_consts - o.a) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: bitset out of range");
                depth++;
                break;
            case NMN_PRED_AND: case NMN_PRED_OR:
                if (depth < 2) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: stack underflow");
                depth--;
                break;
            default: return set_error(NMN_ERR_INVALID_ARGUMENT, "program: unknown opcode");
        }
        if (depth > 64) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: stack deeper than 64");
        dops[i] = d;
    }
    if (depth != 1) return set_error(NMN_ERR_INVALID_ARGUMENT, "program: must leave exactly one value");
    return NMN_OK;
}

// upload the program, run the predicate and the count reduction on `s`, wait, return the number of selected rows.
// `*prog_buf` (device staging, grown as needed), `count` (1 + kPredMaxBlocks words) and `mask` belong to the caller.
static nmn_status pred_run(nmn_columns* c, const std::vector<DevOp>& dops, const uint64_t* consts, uint64_t n_consts,
                           uint64_t n_rows, hipStream_t s, uint8_t** prog_buf, size_t* prog_cap, unsigned long long* count,
                           uint64_t* mask, uint64_t* count_out, uint8_t** pin = nullptr, size_t* pin_cap = nullptr) {
    const uint32_t n_ops = (uint32_t)dops.size();
    const size_t ops_bytes = (size_t)n_ops * sizeof(DevOp), need = ops_bytes + (size_t)n_consts * 8 + 8;
    if (need > *prog_cap) {
        if (*prog_buf) COL_TRY(hipFree(*prog_buf));
        *prog_buf = nullptr;
        *prog_cap = 0;
        COL_TRY(hipMalloc(reinterpret_cast<void**>(prog_buf), need * 2));
        *prog_cap = need * 2;
    }
    // one H2D copy for the program and its constants — through pinned memory when the caller has some (concurrent
    // evaluations: pageable copies serialise in the runtime)
    const size_t stage_bytes = ops_bytes + (size_t)n_consts * 8;
    std::vector<uint8_t> staging;
    uint8_t* stage = nullptr;
    unsigned long long* cnt_host = nullptr;
    unsigned long long cnt_local = 0;
    if (pin) {
        if (8 + stage_bytes > *pin_cap) {
            if (*pin) (void)hipHostFree(*pin);
            *pin = nullptr;
            *pin_cap = 0;
            COL_TRY(hipHostMalloc(reinterpret_cast<void**>(pin), (8 + stage_bytes) * 2, hipHostMallocDefault));
            *pin_cap = (8 + stage_bytes) * 2;
        }
        cnt_host = reinterpret_cast<unsigned long long*>(*pin);
        stage = *pin + 8;
    } else {
        staging.resize(stage_bytes);
        stage = staging.data();
        cnt_host = &cnt_local;
    }
    memcpy(stage, dops.data(), ops_bytes);
    if (n_consts) memcpy(stage + ops_bytes, consts, (size_t)n_consts * 8);
    COL_TRY(hipMemcpyAsync(*prog_buf, stage, stage_bytes, hipMemcpyHostToDevice, s));
    const uint64_t n_words = (n_rows + 63) / 64;
    const uint64_t wave_trips = (n_words + kPredUnroll - 1) / kPredUnroll;  // one trip = kPredUnroll words of one wave
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((wave_trips + 3) / 4, kPredMaxBlocks);
    hipLaunchKernelGGL(pred_eval_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const DevOp*>(*prog_buf), n_ops,
                       reinterpret_cast<const uint64_t*>(*prog_buf + ops_bytes), c->valid, n_rows, mask, count + 1);
    COL_TRY(hipGetLastError());
    hipLaunchKernelGGL(count_reduce_kernel, dim3(1), dim3(256), 0, s, count + 1, blocks, count);
    COL_TRY(hipGetLastError());
    COL_TRY(hipMemcpyAsync(cnt_host, count, 8, hipMemcpyDeviceToHost, s));
    COL_TRY(hipStreamSynchronize(s));  // also keeps `staging` alive until the H2D copy has been consumed
    *count_out = *cnt_host;
    return NMN_OK;
}

// ---- what nmn_api.hip needs to evaluate the predicates of a query batch on the batch's own stream ------------
namespace nmn {

nmn_status columns_compile(const nmn_columns* c, const nmn_pred_op* prog, uint32_t n_ops, uint64_t n_consts,
                           uint64_t n_rows, std::vector<uint8_t>* ops_bytes) {
    if (!c || !prog || n_ops == 0 || !ops_bytes) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (n_rows > c->cap) return set_error(NMN_ERR_CAPACITY, "n_rows beyond capacity_rows");
    std::vector<DevOp> dops;
    nmn_status st = pred_compile(c, prog, n_ops, n_consts, &dops);  // reads c->cols: stable while no writer runs
    if (st != NMN_OK) return st;
    ops_bytes->resize(dops.size() * sizeof(DevOp));
    memcpy(ops_bytes->data(), dops.data(), ops_bytes->size());
    return NMN_OK;
}
size_t pred_desc_bytes() { return sizeof(ProgDesc); }
size_t pred_op_bytes() { return sizeof(DevOp); }
void pred_desc_write(uint8_t* dst, uint32_t ops_off, uint32_t n_ops, uint32_t consts_off, uint64_t* mask,
                     unsigned long long* counts, unsigned long long* host_total, uint32_t* ticket) {
    ProgDesc d;
    d.ops_off = ops_off;
    d.n_ops = n_ops;
    d.consts_off = consts_off;
    d.pad = 0;
    d.mask = mask;
    d.counts = counts;
    d.host_total = host_total;
    d.ticket = ticket;
    memcpy(dst, &d, sizeof d);
}
// blocks per program (the grid stays around kPredMaxBlocks); counts of a program: 1 + pred_batch_blocks words
uint32_t pred_batch_blocks(uint64_t n_rows, uint32_t n_prog) {
    const uint64_t n_words = (n_rows + 63) / 64;
    const uint64_t wave_trips = (n_words + kPredUnroll - 1) / kPredUnroll;
    return (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>((wave_trips + 3) / 4, std::max<uint64_t>(kPredMaxBlocks / std::max(n_prog, 1u), 64)));
}
// dev_block: [ProgDesc x n_prog | ops and constants], already on the device (same stream)
hipError_t launch_pred_batch(const nmn_columns* c, const uint8_t* dev_block, uint32_t n_prog, uint64_t n_rows,
                             hipStream_t s, bool counts_by_ticket) {
    if (n_prog == 0 || n_rows == 0) return hipSuccess;
    const uint32_t blocks = pred_batch_blocks(n_rows, n_prog);
    hipLaunchKernelGGL(pred_eval_batch_kernel, dim3(blocks, n_prog), dim3(256), 0, s, dev_block, c->valid, n_rows);
    if (!counts_by_ticket) hipLaunchKernelGGL(count_reduce_batch_kernel, dim3(n_prog), dim3(256), 0, s, dev_block, blocks);  // (else: every descriptor carries a ticket)
    return hipGetLastError();
}
uint64_t columns_words(const nmn_columns* c) { return c ? c->words : 0; }
int columns_device(const nmn_columns* c) { return c ? c->device : -1; }

}  // namespace nmn

extern "C" nmn_status nmn_columns_eval(nmn_columns* c, const nmn_pred_op* prog, uint32_t n_ops, const uint64_t* consts,
                                       uint64_t n_consts, uint64_t n_rows, uint64_t* count_out) {
    if (!c || !prog || n_ops == 0 || !count_out || (n_consts && !consts))
        return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (n_rows > c->cap) return set_error(NMN_ERR_CAPACITY, "n_rows beyond capacity_rows");
    std::vector<DevOp> dops;
    nmn_status st = pred_compile(c, prog, n_ops, n_consts, &dops);
    if (st != NMN_OK) return st;
    *count_out = 0;
    if (n_rows == 0) return NMN_OK;
    COL_TRY(hipSetDevice(c->device));
    return pred_run(c, dops, consts, n_consts, n_rows, c->stream, &c->prog, &c->prog_cap, c->count, c->mask, count_out);
}

extern "C" nmn_status nmn_columns_eval_acquire(nmn_columns* c, const nmn_pred_op* prog, uint32_t n_ops,
                                               const uint64_t* consts, uint64_t n_consts, uint64_t n_rows,
                                               uint64_t* count_out, uint32_t* slot_out, const uint64_t** mask_out) {
    if (!c || !prog || n_ops == 0 || !count_out || !slot_out || !mask_out || (n_consts && !consts))
        return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *count_out = 0;
    *mask_out = nullptr;
    if (n_rows > c->cap) return set_error(NMN_ERR_CAPACITY, "n_rows beyond capacity_rows");
    COL_TRY(hipSetDevice(c->device));
    std::vector<DevOp> dops;
    nmn_columns::EvalSlot* slot = nullptr;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        nmn_status st = pred_compile(c, prog, n_ops, n_consts, &dops);
        if (st != NMN_OK) return st;
        for (;;) {
            for (size_t i = 0; i < c->slots.size() && !slot; i++)
                if (!c->slots[i]->busy) {
                    slot = c->slots[i].get();
                    *slot_out = (uint32_t)i;
                }
            if (slot || c->slots.size() < nmn_columns::kMaxSlots) break;
            c->slot_cv.wait(lk);
        }
        if (!slot) {
            auto fresh = std::make_unique<nmn_columns::EvalSlot>();
            hipError_t e = hipStreamCreateWithFlags(&fresh->stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&fresh->count), (1 + (size_t)kPredMaxBlocks) * 8);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&fresh->mask), (c->words + 1) * 8);
            if (e != hipSuccess) {
                if (fresh->stream) (void)hipStreamDestroy(fresh->stream);
                if (fresh->count) (void)hipFree(fresh->count);
                if (fresh->mask) (void)hipFree(fresh->mask);
                return nmn::set_error_hip(e, "predicate slot");
            }
            *slot_out = (uint32_t)c->slots.size();
            c->slots.push_back(std::move(fresh));
            slot = c->slots.back().get();
        }
        slot->busy = true;
    }
    nmn_status st = NMN_OK;
    if (n_rows) st = pred_run(c, dops, consts, n_consts, n_rows, slot->stream, &slot->prog, &slot->prog_cap, slot->count,
                              slot->mask, count_out, &slot->pin, &slot->pin_cap);
    if (st != NMN_OK) {
        std::lock_guard<std::mutex> g(c->mu);
        slot->busy = false;
        c->slot_cv.notify_one();
        return st;
    }
    *mask_out = slot->mask;
    return NMN_OK;
}

extern "C" nmn_status nmn_columns_eval_release(nmn_columns* c, uint32_t slot) {
    if (!c) return set_error(NMN_ERR_INVALID_ARGUMENT, "null columns");
    std::lock_guard<std::mutex> g(c->mu);
    if (slot >= c->slots.size() || !c->slots[slot]->busy) return set_error(NMN_ERR_INVALID_ARGUMENT, "slot not held");
    c->slots[slot]->busy = false;
    c->slot_cv.notify_one();
    return NMN_OK;
}

extern "C" const uint64_t* nmn_columns_mask_device(const nmn_columns* c) { return c ? c->mask : nullptr; }
extern "C" const uint64_t* nmn_columns_valid_device(const nmn_columns* c) { return c ? c->valid : nullptr; }

extern "C" nmn_status nmn_columns_read_mask(nmn_columns* c, uint64_t* out_words, uint64_t n_words) {
    if (!c || (n_words && !out_words)) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (n_words > c->words) return set_error(NMN_ERR_CAPACITY, "words beyond capacity");
    if (n_words == 0) return NMN_OK;
    COL_TRY(hipSetDevice(c->device));
    COL_TRY(hipMemcpyAsync(out_words, c->mask, n_words * 8, hipMemcpyDeviceToHost, c->stream));
    COL_TRY(hipStreamSynchronize(c->stream));
    return NMN_OK;
}
