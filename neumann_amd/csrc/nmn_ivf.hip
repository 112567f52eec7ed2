// nmn_ivf.hip — IVF-Flat probe on the GPU: `IVFIndex::{add, search_with_nprobe}` of
// tensor_store/src/ivf.rs:276-406 for `IVFStorage::Flat`, on top of the flat-scan kernels.
//
// Layout: the vectors stay in ID (insertion) order in one nmn_index — row == the id `add` returns
// (ivf.rs:288) — plus `assign[row]`, the cluster of each row.  An inverted list is therefore a set of
// rows, not a contiguous range: a probe marks the nprobe nearest clusters, one kernel turns
// `assign` into the selection bitmap (1 bit per row), and the masked scan reads only the selected
// rows (3 KB contiguous each at d = 768).  `add` is an append, never a list reshuffle.
//
// Exactness: every distance the reference computes here is `squared_euclidean` (ivf.rs:500-508), the
// same strictly sequential f32 sum as the flat Euclidean metric, so the exact kernels restate it
// bit for bit: centroids are ranked by the squared distance, ascending, ties by centroid index
// (stable sort of an enumerate, ivf.rs:331-337); list members by `sqrt` of it, ascending; equal
// distances keep candidate order = probe order of the cluster, then list (= id) order (stable sort,
// ivf.rs:402).  `add` picks the first nearest centroid (`min_by`, ivf.rs:490-497).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <shared_mutex>
#include <new>
#include <vector>

#include "nmn_index.h"

using namespace nmn;

namespace {

constexpr uint32_t kNoRank = 0xFFFFFFFFu;
constexpr uint32_t kAssignChunk = 4096;  // rows assigned per exact sweep over the centroids

// first centroid whose squared distance is minimal, with `min_by`'s fold semantics on NaN
// (core::iter::Iterator::min_by keeps the earlier element unless the later compares strictly Less):
// scores are -d^2 in tile-major layout score_at(c, q, nql); one wave per new row.
__global__ __launch_bounds__(256) void ivf_assign_kernel(const uint32_t* __restrict__ scores, uint32_t n_clusters,
                                                         uint32_t nql, uint32_t nq, uint32_t* __restrict__ assign_out) {
    const uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (q >= nq) return;
    const float first = u2f(scores[score_at(0, q, nql)]);
    float best = -__builtin_inff();
    uint32_t best_c = kNoRank;
    for (uint32_t c = lane; c < n_clusters; c += 64) {
        const float s = u2f(scores[score_at(c, q, nql)]);
        if (s > best || (s == best && c < best_c)) {  // NaN never wins; ties go to the lower index
            best = s;
            best_c = c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off);
        const uint32_t oc = __shfl_down(best_c, off);
        if (ob > best || (ob == best && oc < best_c)) {
            best = ob;
            best_c = oc;
        }
    }
    if (lane == 0) {
        uint32_t c = best_c;
        if (first != first || c == kNoRank) c = 0;  // a NaN first element is never displaced; all-NaN keeps 0
        // -inf everywhere except NaNs: `best_c` may still be kNoRank only when every score is NaN or -inf;
        // with -inf scores the fold keeps the first element as well
        assign_out[q] = c;
    }
}

__global__ void ivf_probe_rank_kernel(const uint64_t* __restrict__ probe_rows, const uint32_t* __restrict__ probe_count,
                                      uint32_t* __restrict__ probe_rank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *probe_count) probe_rank[probe_rows[i]] = i;
}

// The whole centroid ranking of one query in one workgroup (n_clusters <= kRankMax): composite keys
// `score key << 32 | ~cluster` (score = -d^2, so descending composite = ascending squared distance, ties by cluster
// index — the reference's stable sort of an enumerate), bitonic sort in LDS, then probe_rows[0..np) = the clusters in
// probe order and probe_rank[cluster] = its rank or kNoRank.  Replaces five launches (keys, 4096-key tile sort, emit,
// memset, rank scatter: ~50 us of a 370 us probe) by one.
constexpr uint32_t kRankMax = 4096;
// Query blockIdx.x of a chunk of nql queries (their scores interleaved tile-major: score_at): outputs at probe_rows + q * n_clusters,
// probe_rank + q * n_clusters, probe_count + q.
__global__ __launch_bounds__(1024) void ivf_rank_kernel(const uint32_t* __restrict__ cscores, uint32_t nql, uint32_t n_clusters,
                                                        uint32_t np2, uint32_t np, uint64_t* __restrict__ probe_rows,
                                                        uint32_t* __restrict__ probe_count,
                                                        uint32_t* __restrict__ probe_rank) {
    __shared__ uint64_t t[kRankMax];
    const uint32_t tid = threadIdx.x, qi = blockIdx.x;
    probe_rows += (size_t)qi * n_clusters;
    probe_rank += (size_t)qi * n_clusters;
    probe_count += qi;
    for (uint32_t i = tid; i < np2; i += 1024) {
        uint64_t key = 0;
        if (i < n_clusters) {
            const uint32_t sk = bits_to_key(cscores[score_at(i, qi, nql)]);
            if (sk != kKeyMasked) key = ((uint64_t)sk << 32) | (uint32_t)~i;
        }
        t[i] = key;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t p = tid; p < (np2 >> 1); p += 1024) {
                const uint32_t lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = t[lo], b = t[hi];
                if ((a < b) == desc) {
                    t[lo] = b;
                    t[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t c = tid; c < n_clusters; c += 1024) probe_rank[c] = kNoRank;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < np; i += 1024) {
        const uint64_t key = t[i];
        if (key != 0) {
            const uint32_t c = ~(uint32_t)key;
            probe_rows[i] = c;
            probe_rank[c] = i;
            mine++;
        }
    }
    // participating clusters sort first: count = number of non-zero keys among the first np
    __shared__ uint32_t total;
    if (tid == 0) total = 0;
    __syncthreads();
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (tid == 0) *probe_count = total;
}

// rows [row_lo, n_rows) whose list is probed (the rows below row_lo are served by the list-major copy)
// (query blockIdx.y of a chunk: its ranks at probe_rank + y * n_clusters, its bitmap at mask + y * mask_stride)
__global__ __launch_bounds__(256) void ivf_mask_kernel(const uint32_t* __restrict__ assign,
                                                       const uint32_t* __restrict__ probe_rank, uint32_t n_clusters, uint64_t n_rows,
                                                       uint64_t row_lo, uint64_t* __restrict__ mask, uint64_t mask_stride) {
    probe_rank += (size_t)blockIdx.y * n_clusters;
    mask += (size_t)blockIdx.y * mask_stride;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_words = (n_rows + 63) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t w = wave; w < n_words; w += n_waves) {
        uint64_t word = 0ull;
        if (((w + 1) << 6) > row_lo) {  // (wave-uniform: words entirely below row_lo read nothing)
            const uint64_t row = (w << 6) + lane;
            const bool in = row >= row_lo && row < n_rows && probe_rank[assign[row]] != kNoRank;
            word = __ballot(in);
        }
        if (lane == 0) mask[w] = word;
    }
}

// The same selection over the LIST-MAJOR copy: its rows [list_off[c], list_off[c + 1]) are list c, so a probed list is a run of
// set bits and the list scan streams contiguous rows.  No per-row array is read: a lane finds its row's list by bisecting the
// (n_clusters + 1)-entry offset table.
__global__ __launch_bounds__(256) void ivf_range_mask_kernel(const uint32_t* __restrict__ list_off, uint32_t n_clusters,
                                                             const uint32_t* __restrict__ probe_rank, uint64_t n_rows,
                                                             uint64_t* __restrict__ mask, uint64_t mask_stride) {
    probe_rank += (size_t)blockIdx.y * n_clusters;
    mask += (size_t)blockIdx.y * mask_stride;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_words = (n_rows + 63) >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t w = wave; w < n_words; w += n_waves) {
        const uint64_t row = (w << 6) + lane;
        bool in = false;
        if (row < n_rows) {
            uint32_t lo = 0, hi = n_clusters;  // the list c with list_off[c] <= row < list_off[c + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (list_off[mid] <= (uint32_t)row) lo = mid;
                else hi = mid;
            }
            in = probe_rank[lo] != kNoRank;
        }
        const uint64_t word = __ballot(in);
        if (lane == 0) mask[w] = word;
    }
}

// dst[i][0 .. dim) = src[perm[pos0 + i]][0 .. dim): rows of the id-ordered corpus (stride ld) gathered into a tight block
__global__ __launch_bounds__(256) void ivf_gather_kernel(const float* __restrict__ src, uint32_t ld, uint32_t dim,
                                                         const uint32_t* __restrict__ perm, uint64_t pos0, uint32_t cnt,
                                                         float* __restrict__ dst) {
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
        const float* row = src + (uint64_t)perm[pos0 + i] * ld;
        float* out = dst + (uint64_t)i * dim;
        for (uint32_t c = threadIdx.x; c < dim; c += 256) out[c] = row[c];
    }
}

}  // namespace

struct nmn_ivf {
    nmn_index* vectors = nullptr;
    nmn_index* centroids = nullptr;
    uint32_t n_clusters = 0, dim = 0;
    int device = 0;
    uint64_t cap = 0;
    // The same vectors a second time in LIST-MAJOR order (the reference's IVFStorage::Flat keeps a Vec per list, ivf.rs:160-175):
    // the rows of list 0, then list 1, ..., ids ascending inside a list.  A probe over it reads contiguous row ranges instead of
    // a bitmap's worth of scattered rows, and no per-row side array at all.  Rebuilt (ivf_relayout) after a build / load and
    // when the vectors added since make up an eighth of it; those younger vectors are scanned in `vectors` through the bitmap.
    nmn_index* cvec = nullptr;
    uint64_t c_rows = 0;                 // ids [0, c_rows) are in cvec
    std::vector<uint32_t> perm_host;     // cvec row -> id
    std::vector<uint32_t> list_off_host; // [n_clusters + 1] first cvec row of every list
    uint32_t* list_off = nullptr;        // device copy
    bool cvec_failed = false;            // no HBM for the copy: probes stay on the bitmap over `vectors`
    uint32_t flags = 0, cand_cap = 0;    // of the nmn_index_desc the index was created with
    uint32_t* assign = nullptr;          // device [cap]
    std::vector<uint32_t> assign_host;   // same, for cluster_sizes and the tie order of equal distances
    std::vector<float> centroids_host;   // trained centroids (nmn_ivf_build), row-major n_clusters x dim
    hipStream_t stream = nullptr;
    uint32_t* cscores = nullptr;         // exact -d^2 of a chunk of queries vs every centroid (tile-major)
    size_t cscores_cap = 0;
    uint64_t* ckeys = nullptr;           // sort buffer of the centroid ranking
    uint64_t* probe_rows = nullptr;      // [n_clusters] clusters in probe order
    float* probe_scores = nullptr;
    uint32_t* probe_count = nullptr;
    uint32_t* probe_rank = nullptr;      // [n_clusters] probe rank or kNoRank
    uint64_t* mask = nullptr;            // [ceil(cap/64)]
    float* qraw = nullptr;               // one query, dim floats
    float* qpad = nullptr;               // padded to ld
    QInfo* qinfo = nullptr;              // [kAssignChunk] (qmag is unused by the L2 metrics)
    QState* qstate = nullptr;
    uint32_t* assign_tmp = nullptr;      // [kAssignChunk]
    uint32_t assign_chunk = kAssignChunk;
    // Searches hold `rw` shared (add / build / the accessors exclusive) and take a PROBE SLOT each: stream and scratch of
    // one centroid ranking + selection bitmap, created on demand.  The list scans themselves go through the flat index's
    // host API, whose coalescer lets several selective probes run side by side.
    std::shared_mutex rw;
    struct ProbeSlot {
        hipStream_t stream = nullptr;
        uint32_t* cscores = nullptr;     // [centroids padded]
        uint64_t* ckeys = nullptr;       // large-k sort buffer (more than kRankMax centroids only)
        uint64_t* probe_rows = nullptr;  // [n_clusters]
        float* probe_scores = nullptr;
        uint32_t* probe_count = nullptr;
        uint32_t* probe_rank = nullptr;
        uint64_t* mask = nullptr;
        uint64_t* mask_c = nullptr;      // selection over the list-major copy
        float* qraw = nullptr;
        float* qpad = nullptr;
        QInfo* qinfo = nullptr;
        QState* qstate = nullptr;
        uint8_t* pin = nullptr;          // pinned: [queries nb x dim x 4 | probe rows nb x n_clusters x 8]
        uint32_t nb = 1;                 // queries of one chunk the buffers above are sized for (probe_slot_grow)
        // a lone query's list scans enqueued right behind its bitmaps (one round trip per call): their results, device and pinned,
        // two parts (list-major copy, younger vectors) of res_k entries each: [rows u64 | scores f32 | count u32]
        uint8_t* res_dev = nullptr;
        uint8_t* res_pin = nullptr;
        uint64_t res_k = 0;
        bool busy = false;
    };
    std::vector<std::unique_ptr<ProbeSlot>> slots;
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::vector<uint64_t> list_sizes;    // rows per cluster (host), for the selectivity hint of a probe
    uint64_t list_sizes_rows = 0;        // rows accounted for in list_sizes
    static constexpr size_t kMaxSlots = 64;
};

#define IVF_TRY(expr)                                         \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return set_error_hip(_e, #expr); \
    } while (0)

extern "C" nmn_status nmn_ivf_destroy(nmn_ivf* ivf) {
    if (!ivf) return NMN_OK;
    (void)hipSetDevice(ivf->device);
    if (ivf->stream) (void)hipStreamSynchronize(ivf->stream);
    for (void* p : {(void*)ivf->assign, (void*)ivf->cscores, (void*)ivf->ckeys, (void*)ivf->probe_rows,
                    (void*)ivf->probe_scores, (void*)ivf->probe_count, (void*)ivf->probe_rank, (void*)ivf->mask,
                    (void*)ivf->qraw, (void*)ivf->qpad, (void*)ivf->qinfo, (void*)ivf->qstate, (void*)ivf->assign_tmp,
                    (void*)ivf->list_off})
        if (p) (void)hipFree(p);
    for (auto& sl : ivf->slots) {
        if (sl->stream) {
            (void)hipStreamSynchronize(sl->stream);
            (void)hipStreamDestroy(sl->stream);
        }
        for (void* p : {(void*)sl->cscores, (void*)sl->ckeys, (void*)sl->probe_rows, (void*)sl->probe_scores,
                        (void*)sl->probe_count, (void*)sl->probe_rank, (void*)sl->mask, (void*)sl->mask_c, (void*)sl->qraw, (void*)sl->qpad,
                        (void*)sl->qinfo, (void*)sl->qstate})
            if (p) (void)hipFree(p);
        if (sl->pin) (void)hipHostFree(sl->pin);
        if (sl->res_dev) (void)hipFree(sl->res_dev);
        if (sl->res_pin) (void)hipHostFree(sl->res_pin);
    }
    if (ivf->stream) (void)hipStreamDestroy(ivf->stream);
    if (ivf->cvec) nmn_index_destroy(ivf->cvec);
    if (ivf->vectors) nmn_index_destroy(ivf->vectors);
    if (ivf->centroids) nmn_index_destroy(ivf->centroids);
    delete ivf;
    return NMN_OK;
}

// allocate the index; `centroids` may be null (nmn_ivf_build trains them afterwards)
static nmn_status ivf_new(const nmn_index_desc* desc, const float* centroids, uint32_t n_clusters, nmn_ivf** out) {
    *out = nullptr;
    if (n_clusters == 0) return set_error(NMN_ERR_INVALID_ARGUMENT, "an IVF index needs at least one centroid");
    nmn_ivf* ivf = new (std::nothrow) nmn_ivf();
    if (!ivf) return set_error(NMN_ERR_OUT_OF_MEMORY, "host alloc");
    auto bail = [&](nmn_status st) {
        nmn_ivf_destroy(ivf);
        return st;
    };
    nmn_status st = nmn_index_create(desc, &ivf->vectors);
    if (st != NMN_OK) return bail(st);
    nmn_index_desc cd = *desc;
    cd.capacity_rows = n_clusters;
    cd.row_base = 0;
    cd.device = ivf->vectors->device;
    st = nmn_index_create(&cd, &ivf->centroids);
    if (st != NMN_OK) return bail(st);
    if (centroids) {
        st = nmn_index_upload(ivf->centroids, centroids, 0, n_clusters);
        if (st != NMN_OK) return bail(st);
    }
    ivf->n_clusters = n_clusters;
    ivf->flags = desc->flags;
    ivf->cand_cap = desc->cand_cap;
    ivf->dim = desc->dim;
    ivf->device = ivf->vectors->device;
    ivf->cap = ivf->vectors->cap;
    const uint32_t ld = ivf->vectors->ld;
    const size_t c_pad = ivf->centroids->cap_pad;
    hipError_t e = hipSetDevice(ivf->device);
    auto alloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, std::max<size_t>(bytes, 64));
    };
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ivf->stream, hipStreamNonBlocking);
    alloc(reinterpret_cast<void**>(&ivf->assign), std::max<uint64_t>(ivf->cap, 1) * 4);
    // rows assigned per sweep: bound the score matrix to 64 MiB whatever the number of clusters
    ivf->assign_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(kAssignChunk, (16ull << 20) / c_pad));
    ivf->cscores_cap = c_pad * ivf->assign_chunk;
    alloc(reinterpret_cast<void**>(&ivf->cscores), ivf->cscores_cap * 4);
    alloc(reinterpret_cast<void**>(&ivf->ckeys), largek_sort_len(n_clusters) * 8);
    alloc(reinterpret_cast<void**>(&ivf->probe_rows), (size_t)n_clusters * 8);
    alloc(reinterpret_cast<void**>(&ivf->probe_scores), (size_t)n_clusters * 4);
    alloc(reinterpret_cast<void**>(&ivf->probe_count), 8);  // [0] clusters probed, [1] results of the list scan
    alloc(reinterpret_cast<void**>(&ivf->probe_rank), (size_t)n_clusters * 4);
    alloc(reinterpret_cast<void**>(&ivf->mask), ((ivf->cap + 63) / 64 + 1) * 8);
    alloc(reinterpret_cast<void**>(&ivf->qraw), (size_t)desc->dim * 4);
    alloc(reinterpret_cast<void**>(&ivf->qpad), (size_t)ld * 4);
    alloc(reinterpret_cast<void**>(&ivf->qinfo), sizeof(QInfo) * kAssignChunk);
    alloc(reinterpret_cast<void**>(&ivf->qstate), sizeof(QState) * kAssignChunk);
    alloc(reinterpret_cast<void**>(&ivf->assign_tmp), 4 * kAssignChunk);
    if (e == hipSuccess) e = hipMemsetAsync(ivf->qinfo, 0, sizeof(QInfo) * kAssignChunk, ivf->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ivf->stream);
    if (e != hipSuccess) return bail(set_error_hip(e, "nmn_ivf_create"));
    *out = ivf;
    return NMN_OK;
}

extern "C" nmn_status nmn_ivf_create(const nmn_index_desc* desc, const float* centroids, uint32_t n_clusters,
                                     nmn_ivf** out) {
    if (!desc || !centroids || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    return ivf_new(desc, centroids, n_clusters, out);
}

extern "C" uint64_t nmn_ivf_len(const nmn_ivf* ivf) { return ivf ? ivf->vectors->rows : 0; }
extern "C" uint32_t nmn_ivf_clusters(const nmn_ivf* ivf) { return ivf ? ivf->n_clusters : 0; }
extern "C" nmn_index* nmn_ivf_vectors(nmn_ivf* ivf) { return ivf ? ivf->vectors : nullptr; }

extern "C" nmn_status nmn_ivf_cluster_sizes(nmn_ivf* ivf, uint64_t* out_sizes) {
    if (!ivf || !out_sizes) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::unique_lock<std::shared_mutex> g(ivf->rw);
    std::fill(out_sizes, out_sizes + ivf->n_clusters, 0ull);
    for (uint32_t c : ivf->assign_host) out_sizes[c]++;
    return NMN_OK;
}

// exact -d^2 of `nq` queries (already padded to ld in device memory) against every centroid -> cscores
static nmn_status centroid_scores(nmn_ivf* ivf, const float* qpad_dev, uint32_t nq) {
    ExactScanParams ep{};
    ep.corpus = ivf->centroids->corpus;
    ep.norms = ivf->centroids->norms;
    ep.qpad = qpad_dev;
    ep.qinfo = ivf->qinfo;
    ep.qstate = nullptr;
    ep.mask = nullptr;
    ep.scores = ivf->cscores;
    ep.n_rows = ivf->n_clusters;
    ep.nql = nq;
    ep.ld = ivf->centroids->ld;
    ep.dim = ivf->dim;
    ep.nq = nq;
    ep.metric = kMetricNegL2Sq;
    IVF_TRY(launch_exact_scan(ep, ivf->stream));
    return NMN_OK;
}

// nearest centroid (first minimum of the exact squared distances) of rows [row0, row0 + n) -> assign / assign_host
static nmn_status assign_rows(nmn_ivf* ivf, uint64_t row0, uint64_t n) {
    const uint32_t ld = ivf->vectors->ld;
    if (ivf->assign_host.size() < row0 + n) ivf->assign_host.resize(row0 + n);
    for (uint64_t off = 0; off < n; off += ivf->assign_chunk) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(ivf->assign_chunk, n - off);
        // the rows are laid out exactly like padded queries: score them against the centroids in place
        nmn_status st = centroid_scores(ivf, ivf->vectors->corpus + (row0 + off) * (uint64_t)ld, cnt);
        if (st != NMN_OK) return st;
        hipLaunchKernelGGL(ivf_assign_kernel, dim3((cnt * 64 + 255) / 256), dim3(256), 0, ivf->stream, ivf->cscores,
                           ivf->n_clusters, cnt, cnt, ivf->assign + row0 + off);
        IVF_TRY(hipGetLastError());
        IVF_TRY(hipMemcpyAsync(ivf->assign_host.data() + row0 + off, ivf->assign + row0 + off, (size_t)cnt * 4,
                               hipMemcpyDeviceToHost, ivf->stream));
    }
    IVF_TRY(hipStreamSynchronize(ivf->stream));
    if (row0 == 0 || ivf->list_sizes.size() != ivf->n_clusters) {
        ivf->list_sizes.assign(ivf->n_clusters, 0);
        for (uint64_t r = 0; r < row0; r++) ivf->list_sizes[ivf->assign_host[r]]++;
    }
    for (uint64_t r = row0; r < row0 + n; r++) ivf->list_sizes[ivf->assign_host[r]]++;
    ivf->list_sizes_rows = row0 + n;
    return NMN_OK;
}

// (Re)build the list-major copy from the id-ordered vectors and their list assignments.  Caller holds ivf->rw exclusively.
// A stable counting sort of the assignments gives the permutation; the rows are gathered on the device 128 Ki rows at a time
// and go through the ordinary device upload (magnitudes, mirrors).  Not enough HBM for the copy is not an error.
static nmn_status ivf_relayout(nmn_ivf* ivf) {
    static const bool disabled = getenv("NMN_IVF_NO_LIST_MAJOR") != nullptr;  // A/B: every probe through the bitmap over `vectors`
    const uint64_t n = ivf->vectors->rows;
    if (disabled || ivf->cvec_failed || n < 4096 || ivf->assign_host.size() < n) return NMN_OK;
    IVF_TRY(hipSetDevice(ivf->device));
    if (!ivf->cvec) {
        nmn_index_desc d{};
        d.dim = ivf->dim;
        d.flags = ivf->flags;
        d.capacity_rows = ivf->cap;
        d.row_base = 0;
        d.device = ivf->device;
        d.cand_cap = ivf->cand_cap;
        if (nmn_index_create(&d, &ivf->cvec) != NMN_OK || ivf->cvec->ld != ivf->vectors->ld) {
            if (ivf->cvec) nmn_index_destroy(ivf->cvec);
            ivf->cvec = nullptr;
            ivf->cvec_failed = true;
            ivf->c_rows = 0;
            return NMN_OK;
        }
    }
    const uint32_t C = ivf->n_clusters;
    ivf->list_off_host.assign((size_t)C + 1, 0u);
    for (uint64_t r = 0; r < n; r++) ivf->list_off_host[ivf->assign_host[r] + 1]++;
    for (uint32_t c = 0; c < C; c++) ivf->list_off_host[c + 1] += ivf->list_off_host[c];
    ivf->perm_host.resize(n);
    {
        std::vector<uint32_t> cur(ivf->list_off_host.begin(), ivf->list_off_host.end() - 1);
        for (uint64_t r = 0; r < n; r++) ivf->perm_host[cur[ivf->assign_host[r]]++] = (uint32_t)r;  // ids ascending inside a list
    }
    constexpr uint64_t kChunk = 128 * 1024;
    uint32_t* perm_dev = nullptr;
    float* tmp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&perm_dev), n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tmp), std::min(kChunk, n) * (uint64_t)ivf->dim * 4);
    if (e == hipSuccess && !ivf->list_off) e = hipMalloc(reinterpret_cast<void**>(&ivf->list_off), ((size_t)C + 1) * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(perm_dev, ivf->perm_host.data(), n * 4, hipMemcpyHostToDevice, ivf->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ivf->list_off, ivf->list_off_host.data(), ((size_t)C + 1) * 4, hipMemcpyHostToDevice, ivf->stream);
    nmn_status st = NMN_OK;
    for (uint64_t pos0 = 0; pos0 < n && e == hipSuccess && st == NMN_OK; pos0 += kChunk) {
        const uint32_t cnt = (uint32_t)std::min(kChunk, n - pos0);
        hipLaunchKernelGGL(ivf_gather_kernel, dim3(std::min<uint32_t>(cnt, 4096)), dim3(256), 0, ivf->stream, ivf->vectors->corpus,
                           ivf->vectors->ld, ivf->dim, perm_dev, pos0, cnt, tmp);
        e = hipGetLastError();
        if (e == hipSuccess) st = nmn_index_upload_device(ivf->cvec, tmp, pos0, cnt, ivf->stream);
        if (e == hipSuccess && st == NMN_OK) e = hipStreamSynchronize(ivf->stream);  // (`tmp` is reused by the next chunk)
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ivf->stream);
    if (perm_dev) (void)hipFree(perm_dev);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess || st != NMN_OK) {
        // (out of memory half way, ...): drop the copy, keep serving from the bitmap over `vectors`
        (void)hipGetLastError();
        nmn_index_destroy(ivf->cvec);
        ivf->cvec = nullptr;
        ivf->cvec_failed = true;
        ivf->c_rows = 0;
        return NMN_OK;
    }
    ivf->c_rows = n;
    return NMN_OK;
}

extern "C" uint64_t nmn_ivf_list_major_rows(const nmn_ivf* ivf) { return ivf ? ivf->c_rows : 0; }

extern "C" nmn_status nmn_ivf_add(nmn_ivf* ivf, const float* rows_host, uint64_t n, uint32_t* clusters_out) {
    if (!ivf || (n && !rows_host)) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return NMN_OK;
    std::unique_lock<std::shared_mutex> g(ivf->rw);
    const uint64_t row0 = ivf->vectors->rows;
    nmn_status st = nmn_index_upload(ivf->vectors, rows_host, row0, n);  // ids = insertion order (ivf.rs:287-289)
    if (st != NMN_OK) return st;
    IVF_TRY(hipSetDevice(ivf->device));
    IVF_TRY(hipStreamSynchronize(ivf->vectors->host_stream));  // rows are in place before another stream reads them
    st = assign_rows(ivf, row0, n);
    if (st != NMN_OK) return st;
    if (clusters_out) memcpy(clusters_out, ivf->assign_host.data() + row0, n * 4);
    // the vectors added since the list-major copy was laid out are scanned through the bitmap; once they are an eighth of it
    // (or there is no copy yet and the index is worth one) it is laid out afresh
    const uint64_t rows = row0 + n, young = rows - ivf->c_rows;
    if (young >= std::max<uint64_t>(4096, ivf->c_rows / 8)) return ivf_relayout(ivf);
    return NMN_OK;
}

// ---- IVFIndex::train + add on the GPU (ivf.rs:222-316; KMeans::fit, delta_vector.rs:737-901) ----------------------
namespace {

inline uint64_t lcg(uint64_t s) { return s * 6364136223846793005ull + 1ull; }  // wrapping_mul / wrapping_add

// euclidean_distance_sq (delta_vector.rs:896-901) on the host: k * dim per iteration (centroid movement only)
float host_dist_sq(const float* a, const float* b, uint64_t dim) {
    float s = -0.0f;
    for (uint64_t i = 0; i < dim; i++) {
        const float d = a[i] - b[i];
        const float p = d * d;
        s = s + p;
    }
    return s;
}

}  // namespace

extern "C" nmn_status nmn_ivf_build(const nmn_index_desc* desc, const float* rows_host, uint64_t n, uint32_t num_clusters,
                                    const nmn_kmeans_options* opt, nmn_ivf** out) {
    if (!desc || !rows_host || !opt || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (n == 0 || num_clusters == 0) return set_error(NMN_ERR_INVALID_ARGUMENT, "nothing to train on");
    if (desc->capacity_rows < n) return set_error(NMN_ERR_CAPACITY, "capacity_rows < n");
    const uint32_t k = (uint32_t)std::min<uint64_t>(num_clusters, n);  // `k.min(vectors.len())`
    const uint64_t dim = desc->dim;
    nmn_ivf* ivf = nullptr;
    nmn_status st = ivf_new(desc, nullptr, k, &ivf);
    if (st != NMN_OK) return st;
    auto bail = [&](nmn_status code) {
        nmn_ivf_destroy(ivf);
        return code;
    };
    std::unique_lock<std::shared_mutex> g(ivf->rw);
    st = nmn_index_upload(ivf->vectors, rows_host, 0, n);  // ids = order of the input (ivf.rs:287-289)
    if (st != NMN_OK) return bail(st);
    hipError_t he = hipSetDevice(ivf->device);
    if (he == hipSuccess) he = hipStreamSynchronize(ivf->vectors->host_stream);
    if (he != hipSuccess) return bail(set_error_hip(he, "nmn_ivf_build"));
    const uint32_t ld = ivf->vectors->ld;
    hipStream_t s = ivf->stream;
    std::vector<float> cents((size_t)k * dim);  // current centroids, host copy (row-major k x dim)

    // ---- initialisation -----------------------------------------------------------------------------------
    if (opt->init_method == 0) {  // KMeansInit::Random: Fisher-Yates with the LCG, first k indices (delta_vector.rs:781-800)
        std::vector<uint64_t> idx(n);
        for (uint64_t i = 0; i < n; i++) idx[i] = i;
        uint64_t state = opt->seed;
        for (uint64_t i = n - 1; i >= 1; i--) {
            state = lcg(state);
            std::swap(idx[i], idx[state % (i + 1)]);
        }
        for (uint32_t j = 0; j < k; j++) memcpy(cents.data() + (size_t)j * dim, rows_host + idx[j] * dim, dim * sizeof(float));
    } else {  // KMeansPlusPlus (delta_vector.rs:805-853): distances on the GPU, the f32 running sums on the host
        uint32_t* sweep = nullptr;
        float* dist_dev = nullptr;
        const uint64_t n_pad = (n + 63) & ~63ull;
        he = hipMalloc(reinterpret_cast<void**>(&sweep), n_pad * 4);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&dist_dev), n * 4);
        std::vector<float> dist(n, 3.402823466e+38f);  // f32::MAX
        if (he == hipSuccess) he = hipMemcpyAsync(dist_dev, dist.data(), n * 4, hipMemcpyHostToDevice, s);
        uint64_t state = lcg(opt->seed);
        uint64_t pick = state % n;
        memcpy(cents.data(), rows_host + pick * dim, dim * sizeof(float));
        for (uint32_t j = 1; j < k && he == hipSuccess; j++) {
            // dist[i] = min(dist[i], |v_i - last centroid|^2): the last centroid is row `pick`, already a padded query
            ExactScanParams ep{};
            ep.corpus = ivf->vectors->corpus;
            ep.norms = ivf->vectors->norms;
            ep.qpad = ivf->vectors->corpus + pick * (uint64_t)ld;
            ep.qinfo = ivf->qinfo;
            ep.scores = sweep;
            ep.n_rows = n;
            ep.nql = 1;
            ep.ld = ld;
            ep.dim = (uint32_t)dim;
            ep.nq = 1;
            ep.metric = kMetricNegL2Sq;
            he = launch_exact_scan(ep, s);
            if (he == hipSuccess) he = launch_kmeans_min_update(dist_dev, sweep, n, s);
            if (he == hipSuccess) he = hipMemcpyAsync(dist.data(), dist_dev, n * 4, hipMemcpyDeviceToHost, s);
            if (he == hipSuccess) he = hipStreamSynchronize(s);
            if (he != hipSuccess) break;
            float total = -0.0f;  // `distances.iter().sum()`
            for (uint64_t i = 0; i < n; i++) total = total + dist[i];
            state = lcg(state);
            if (total == 0.0f) {
                pick = state % n;
            } else {
                const float frac = (float)state / (float)UINT64_MAX;  // `rng_state as f32 / u64::MAX as f32`
                const float threshold = frac * total;
                float cumulative = 0.0f;
                pick = 0;
                for (uint64_t i = 0; i < n; i++) {
                    cumulative = cumulative + dist[i];
                    if (cumulative >= threshold) {
                        pick = i;
                        break;
                    }
                }
            }
            memcpy(cents.data() + (size_t)j * dim, rows_host + pick * dim, dim * sizeof(float));
        }
        if (sweep) (void)hipFree(sweep);
        if (dist_dev) (void)hipFree(dist_dev);
        if (he != hipSuccess) return bail(set_error_hip(he, "k-means++ initialisation"));
    }

    // ---- Lloyd iterations ---------------------------------------------------------------------------------
    uint32_t* members_dev = nullptr;
    uint64_t* offsets_dev = nullptr;
    float* new_dev = nullptr;
    he = hipMalloc(reinterpret_cast<void**>(&members_dev), n * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&offsets_dev), ((size_t)k + 1) * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&new_dev), (size_t)k * ld * 4);
    auto free_tmp = [&] {
        for (void* p : {(void*)members_dev, (void*)offsets_dev, (void*)new_dev})
            if (p) (void)hipFree(p);
    };
    if (he != hipSuccess) {
        free_tmp();
        return bail(set_error_hip(he, "k-means buffers"));
    }
    std::vector<uint32_t> members(n);
    std::vector<uint64_t> offsets((size_t)k + 1);
    std::vector<float> new_pad((size_t)k * ld), new_cents((size_t)k * dim);
    st = nmn_index_upload(ivf->centroids, cents.data(), 0, k);
    for (uint64_t it = 0; it < opt->max_iterations && st == NMN_OK; it++) {
        he = hipStreamSynchronize(ivf->centroids->host_stream);
        if (he != hipSuccess) break;
        st = assign_rows(ivf, 0, n);  // nearest_centroid for every vector (delta_vector.rs:755-757)
        if (st != NMN_OK) break;
        // update_centroids (867-893): members of each cluster in vector order (stable counting sort)
        std::fill(offsets.begin(), offsets.end(), 0ull);
        for (uint64_t i = 0; i < n; i++) offsets[ivf->assign_host[i] + 1]++;
        for (uint32_t c = 0; c < k; c++) offsets[c + 1] += offsets[c];
        {
            std::vector<uint64_t> cur(offsets.begin(), offsets.end() - 1);
            for (uint64_t i = 0; i < n; i++) members[cur[ivf->assign_host[i]]++] = (uint32_t)i;
        }
        he = hipMemcpyAsync(members_dev, members.data(), n * 4, hipMemcpyHostToDevice, s);
        if (he == hipSuccess) he = hipMemcpyAsync(offsets_dev, offsets.data(), ((size_t)k + 1) * 8, hipMemcpyHostToDevice, s);
        if (he == hipSuccess) he = launch_kmeans_update(ivf->vectors->corpus, ld, (uint32_t)dim, members_dev, offsets_dev, k, new_dev, s);
        if (he == hipSuccess) he = hipMemcpyAsync(new_pad.data(), new_dev, (size_t)k * ld * 4, hipMemcpyDeviceToHost, s);
        if (he == hipSuccess) he = hipStreamSynchronize(s);
        if (he != hipSuccess) break;
        float movement = 0.0f;  // `.fold(0.0f32, f32::max)` of the centroid displacements (763-767)
        for (uint32_t c = 0; c < k; c++) {
            memcpy(new_cents.data() + (size_t)c * dim, new_pad.data() + (size_t)c * ld, dim * sizeof(float));
            movement = std::fmax(movement, std::sqrt(host_dist_sq(cents.data() + (size_t)c * dim, new_cents.data() + (size_t)c * dim, dim)));
        }
        cents.swap(new_cents);
        st = nmn_index_upload(ivf->centroids, cents.data(), 0, k);
        if (movement < opt->convergence_threshold) break;
    }
    free_tmp();
    if (he != hipSuccess) return bail(set_error_hip(he, "k-means iteration"));
    if (st != NMN_OK) return bail(st);
    // ---- `for vector in &vectors { index.add(vector) }` under the trained centroids --------------------------
    he = hipStreamSynchronize(ivf->centroids->host_stream);
    if (he != hipSuccess) return bail(set_error_hip(he, "nmn_ivf_build"));
    st = assign_rows(ivf, 0, n);
    if (st != NMN_OK) return bail(st);
    ivf->centroids_host = cents;
    st = ivf_relayout(ivf);
    if (st != NMN_OK) return bail(st);
    *out = ivf;
    return NMN_OK;
}

extern "C" nmn_status nmn_ivf_centroids(nmn_ivf* ivf, float* out, uint64_t cap_floats) {
    if (!ivf || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    std::unique_lock<std::shared_mutex> g(ivf->rw);
    const uint64_t need = (uint64_t)ivf->n_clusters * ivf->dim;
    if (cap_floats < need) return set_error(NMN_ERR_BUFFER_TOO_SMALL, "centroid buffer too small");
    if (ivf->centroids_host.size() == need) {
        memcpy(out, ivf->centroids_host.data(), need * sizeof(float));
        return NMN_OK;
    }
    IVF_TRY(hipSetDevice(ivf->device));
    IVF_TRY(hipMemcpy2D(out, (size_t)ivf->dim * 4, ivf->centroids->corpus, (size_t)ivf->centroids->ld * 4, (size_t)ivf->dim * 4,
                        ivf->n_clusters, hipMemcpyDeviceToHost));
    return NMN_OK;
}

// a free probe slot (created on demand; waits when kMaxSlots are all busy)
static nmn_status probe_slot_acquire(nmn_ivf* ivf, nmn_ivf::ProbeSlot** out, uint32_t* busy_now = nullptr) {
    std::unique_lock<std::mutex> lk(ivf->slot_mu);
    if (busy_now) {
        *busy_now = 1;
        for (auto& sl : ivf->slots) *busy_now += sl->busy ? 1u : 0u;
    }
    for (;;) {
        for (auto& sl : ivf->slots)
            if (!sl->busy) {
                sl->busy = true;
                *out = sl.get();
                return NMN_OK;
            }
        if (ivf->slots.size() < nmn_ivf::kMaxSlots) break;
        ivf->slot_cv.wait(lk);
    }
    auto sl = std::make_unique<nmn_ivf::ProbeSlot>();
    const size_t c_pad = ivf->centroids->cap_pad;
    const uint32_t ld = ivf->vectors->ld;
    hipError_t e = hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking);
    auto alloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, std::max<size_t>(bytes, 64));
    };
    alloc(reinterpret_cast<void**>(&sl->cscores), c_pad * 4);
    if (ivf->n_clusters > kRankMax) alloc(reinterpret_cast<void**>(&sl->ckeys), largek_sort_len(ivf->n_clusters) * 8);
    alloc(reinterpret_cast<void**>(&sl->probe_rows), (size_t)ivf->n_clusters * 8);
    alloc(reinterpret_cast<void**>(&sl->probe_scores), (size_t)ivf->n_clusters * 4);
    alloc(reinterpret_cast<void**>(&sl->probe_count), 8);
    alloc(reinterpret_cast<void**>(&sl->probe_rank), (size_t)ivf->n_clusters * 4);
    alloc(reinterpret_cast<void**>(&sl->mask), ((ivf->cap + 63) / 64 + 1) * 8);
    alloc(reinterpret_cast<void**>(&sl->mask_c), ((ivf->cap + 63) / 64 + 1) * 8);
    alloc(reinterpret_cast<void**>(&sl->qraw), (size_t)ivf->dim * 4);
    alloc(reinterpret_cast<void**>(&sl->qpad), (size_t)ld * 4);
    alloc(reinterpret_cast<void**>(&sl->qinfo), sizeof(QInfo));
    alloc(reinterpret_cast<void**>(&sl->qstate), sizeof(QState));
    if (e == hipSuccess) e = hipMemsetAsync(sl->qinfo, 0, sizeof(QInfo), sl->stream);
    if (e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void**>(&sl->pin), (size_t)ivf->dim * 4 + (size_t)ivf->n_clusters * 8 + 16,
                          hipHostMallocDefault);
    if (e != hipSuccess) {
        if (sl->stream) (void)hipStreamDestroy(sl->stream);
        for (void* p : {(void*)sl->cscores, (void*)sl->ckeys, (void*)sl->probe_rows, (void*)sl->probe_scores,
                        (void*)sl->probe_count, (void*)sl->probe_rank, (void*)sl->mask, (void*)sl->mask_c, (void*)sl->qraw, (void*)sl->qpad,
                        (void*)sl->qinfo, (void*)sl->qstate})
            if (p) (void)hipFree(p);
        if (sl->pin) (void)hipHostFree(sl->pin);
        return set_error_hip(e, "IVF probe slot");
    }
    sl->busy = true;
    *out = sl.get();
    ivf->slots.push_back(std::move(sl));
    return NMN_OK;
}
// Size a slot's per-query buffers for chunks of `nb` queries (many queries of one nmn_ivf_search call share the centroid
// sweep, the ranking launch, the bitmap launches and ONE round trip to the host).  Only grows.
static nmn_status probe_slot_grow(nmn_ivf* ivf, nmn_ivf::ProbeSlot* sl, uint32_t nb) {
    if (nb <= sl->nb) return NMN_OK;
    IVF_TRY(hipStreamSynchronize(sl->stream));
    const size_t c_pad = ivf->centroids->cap_pad, C = ivf->n_clusters, words = (ivf->cap + 63) / 64 + 1;
    const uint32_t ld = ivf->vectors->ld;
    hipError_t e = hipSuccess;
    auto regrow = [&](void** p, size_t bytes) {
        if (e != hipSuccess) return;
        if (*p) (void)hipFree(*p);
        *p = nullptr;
        e = hipMalloc(p, std::max<size_t>(bytes, 64));
    };
    regrow(reinterpret_cast<void**>(&sl->cscores), c_pad * nb * 4);
    regrow(reinterpret_cast<void**>(&sl->probe_rows), C * nb * 8);
    regrow(reinterpret_cast<void**>(&sl->probe_count), (size_t)nb * 8);
    regrow(reinterpret_cast<void**>(&sl->probe_rank), C * nb * 4);
    regrow(reinterpret_cast<void**>(&sl->mask), words * nb * 8);
    regrow(reinterpret_cast<void**>(&sl->mask_c), words * nb * 8);
    regrow(reinterpret_cast<void**>(&sl->qraw), (size_t)ivf->dim * nb * 4);
    regrow(reinterpret_cast<void**>(&sl->qpad), (size_t)ld * nb * 4);
    regrow(reinterpret_cast<void**>(&sl->qinfo), sizeof(QInfo) * nb);
    regrow(reinterpret_cast<void**>(&sl->qstate), sizeof(QState) * nb);
    if (e == hipSuccess) e = hipMemsetAsync(sl->qinfo, 0, sizeof(QInfo) * nb, sl->stream);
    if (e == hipSuccess) {
        if (sl->pin) (void)hipHostFree(sl->pin);
        sl->pin = nullptr;
        e = hipHostMalloc(reinterpret_cast<void**>(&sl->pin), ((size_t)ivf->dim * 4 * nb + 15 & ~(size_t)15) + C * 8 * nb + 16, hipHostMallocDefault);
    }
    if (e != hipSuccess) {
        sl->nb = 0;  // (buffers in an unknown state: the next call grows them again from scratch)
        return set_error_hip(e, "IVF probe slot (chunk buffers)");
    }
    sl->nb = nb;
    return NMN_OK;
}

static void probe_slot_release(nmn_ivf* ivf, nmn_ivf::ProbeSlot* sl) {
    {
        std::lock_guard<std::mutex> g(ivf->slot_mu);
        sl->busy = false;
    }
    ivf->slot_cv.notify_one();
}

extern "C" nmn_status nmn_ivf_search(nmn_ivf* ivf, const float* queries, uint32_t nq, uint32_t k, uint32_t nprobe,
                                     uint64_t* out_ids, float* out_distances, uint32_t* out_counts,
                                     nmn_search_stats* stats) {
    if (!ivf || !queries || !out_ids || !out_distances || !out_counts)
        return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    if (k == 0) return set_error(NMN_ERR_INVALID_TOP_K, "k == 0");
    if (nq == 0) return NMN_OK;
    std::shared_lock<std::shared_mutex> g(ivf->rw);  // concurrent with other searches, not with add / build
    IVF_TRY(hipSetDevice(ivf->device));
    const uint64_t n_rows = ivf->vectors->rows;
    const uint32_t np = std::min<uint32_t>(nprobe, ivf->n_clusters);  // ivf.rs:339
    std::vector<uint32_t> rank_host;
    std::vector<uint64_t> tmp_ids;
    std::vector<float> tmp_dist;
    nmn_ivf::ProbeSlot* sl = nullptr;
    uint32_t searches_now = 1;  // searches of this index in flight, this one included
    nmn_status st = probe_slot_acquire(ivf, &sl, &searches_now);
    if (st != NMN_OK) return st;
    struct Release {
        nmn_ivf* ivf;
        nmn_ivf::ProbeSlot* sl;
        ~Release() { probe_slot_release(ivf, sl); }
    } release{ivf, sl};
    hipStream_t s = sl->stream;
    // Queries are served in chunks of up to kProbeChunk: ONE copy of the chunk's queries, one exact sweep of all of them
    // over the centroids, one ranking launch (a workgroup per query), one launch per bitmap kind (a grid row per query) and
    // one copy of the probe orders back — a single round trip to the host for the whole chunk where every query used to pay
    // its own; the list scans then follow query by query (each its own bitmap and, with k + 1, its own tie handling).
    // (chunk: up to 64 queries, fewer when their bitmaps — two per query — would take more than 256 MiB)
    const size_t mask_words = (ivf->cap + 63) / 64 + 1;
    const uint32_t kProbeChunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(64, ((size_t)256 << 20) / (mask_words * 16)));
    const uint32_t chunk = (ivf->n_clusters <= kRankMax) ? std::min<uint32_t>(nq, kProbeChunk) : 1u;  // (> 4096 lists: the large-k sort, one query at a time)
    st = probe_slot_grow(ivf, sl, chunk);
    if (st != NMN_OK) return st;
    float* pin_q = reinterpret_cast<float*>(sl->pin);
    uint64_t* probe_host_all = reinterpret_cast<uint64_t*>(sl->pin + (((size_t)ivf->dim * 4 * sl->nb + 15) & ~(size_t)15));
    const uint64_t c_rows = ivf->cvec ? std::min(ivf->c_rows, n_rows) : 0;  // ids the list-major copy covers
    // The list scans of a chunk as ONE batch per part (list-major copy / younger vectors): the flat index's batched sweep reads a
    // bitmap per query, so sixteen probes cost one pass over the vectors instead of sixteen launch-bound searches — whenever that
    // pass is the cheaper way (the coalescer's own estimate: a search alone costs max(100 us, its share of a sweep)).
    const uint64_t kk1 = std::min<uint64_t>((uint64_t)k + 1, std::max<uint64_t>(n_rows, 1));
    std::vector<uint64_t> bc_ids, bt_ids, hint_c, hint_t;
    std::vector<float> bc_dist, bt_dist;
    std::vector<uint32_t> bc_cnt, bt_cnt;
    bool batched_c = false, batched_t = false;
    auto probed_of = [&](const uint64_t* probe_host, uint64_t* pc, uint64_t* pt) {
        uint64_t probed_rows = 0, probed_c = 0;
        for (uint32_t i = 0; i < np; i++)
            if (probe_host[i] < ivf->list_sizes.size()) {
                probed_rows += ivf->list_sizes[probe_host[i]];
                if (c_rows) probed_c += ivf->list_off_host[probe_host[i] + 1] - ivf->list_off_host[probe_host[i]];
            }
        if (ivf->list_sizes_rows != n_rows) probed_rows = UINT64_MAX;  // sizes not current (never after add / build)
        *pc = probed_c;
        *pt = (probed_rows == UINT64_MAX) ? UINT64_MAX : probed_rows - std::min(probed_rows, probed_c);
    };
    auto scan_many = [&](nmn_index* ix, const uint64_t* masks, const std::vector<uint64_t>& hints, const float* q0, uint32_t nb,
                         std::vector<uint64_t>& ids, std::vector<float>& dist, std::vector<uint32_t>& cnt, bool* done,
                         nmn_search_stats* stats_out) -> nmn_status {
        *done = false;
        static const bool no_many = getenv("NMN_IVF_NO_BATCHED_SCANS") != nullptr;  // (A/B switch)
        if (no_many || nb < 2 || kk1 > NMN_MAX_TOP_K || nb > index_hostio_many_capacity(ix, kMetricNegL2, (uint32_t)kk1)) return NMN_OK;
        const double sweep_us = (double)ix->rows * ix->ld * 2.0 / 5.5e6, fixed_us = 100.0;
        double separately = 0.0;
        for (uint32_t i = 0; i < nb; i++) {
            const double sel = hints[i] == UINT64_MAX ? 1.0 : (double)hints[i] / (double)std::max<uint64_t>(ix->rows, 1);
            separately += std::max(fixed_us, sel * sweep_us);
        }
        if (separately < 1.1 * sweep_us + fixed_us) return NMN_OK;
        ids.assign((size_t)nb * kk1, UINT64_MAX);
        dist.assign((size_t)nb * kk1, 0.f);
        cnt.assign(nb, 0);
        std::vector<HostSearchSpec> specs(nb);
        for (uint32_t i = 0; i < nb; i++) {
            specs[i].query = q0 + (size_t)i * ivf->dim;
            specs[i].mask = masks + (size_t)i * mask_words;
            specs[i].mask_rows = hints[i];
            specs[i].out_rows = ids.data() + (size_t)i * kk1;
            specs[i].out_scores = dist.data() + (size_t)i * kk1;
            specs[i].out_count = cnt.data() + i;
        }
        nmn_status st2 = index_search_hostio_many(ix, specs.data(), nb, (uint32_t)kk1, kMetricNegL2, stats_out);
        if (st2 == NMN_OK) *done = true;
        return st2;
    };
    for (uint32_t q = 0; q < nq; q++) {
        uint64_t* o_ids = out_ids + (size_t)q * k;
        float* o_dist = out_distances + (size_t)q * k;
        std::fill(o_ids, o_ids + k, UINT64_MAX);
        std::fill(o_dist, o_dist + k, __builtin_inff());
        out_counts[q] = 0;
        if (n_rows == 0 || np == 0) continue;
        const float* qh = queries + (size_t)q * ivf->dim;
        const uint32_t qc = q % chunk;  // position in its chunk
        if (qc == 0) {
            // 1. (whole chunk) rank the centroids by squared distance (ascending; ties by index), keep the first nprobe, turn the
            //    list assignments into selection bitmaps — all on this slot's stream, one wait
            const uint32_t nb = std::min<uint32_t>(chunk, nq - q);
            memcpy(pin_q, qh, (size_t)ivf->dim * 4 * nb);
            IVF_TRY(hipMemcpyAsync(sl->qraw, pin_q, (size_t)ivf->dim * 4 * nb, hipMemcpyHostToDevice, s));
            // The centroid ranking is a squared-distance scan: it reads the first `dim` elements of each query and nothing qprep
            // derives (no magnitude, no margins).  Where the rows are not padded (stride == dim) and the eight-lanes-per-row scan
            // serves (fewer than 2^16 centroids) the raw queries ARE the padded ones: no qprep launch in front of it (5.6 us +
            // a 4-us gap of a lone probe's ~150, profiles/r05q_*).
            const bool raw_q = ivf->centroids->ld == ivf->dim && ivf->vectors->ld == ivf->dim && ivf->n_clusters < (1u << 16);
            if (!raw_q)
                IVF_TRY(launch_qprep(sl->qraw, nb, ivf->dim, ivf->vectors->ld, kMetricNegL2Sq, ivf->centroids->max_norm_bits, sl->qpad,
                                     sl->qinfo, sl->qstate, 0, s));
            {
                ExactScanParams ep{};
                ep.corpus = ivf->centroids->corpus;
                ep.norms = ivf->centroids->norms;
                ep.qpad = raw_q ? sl->qraw : sl->qpad;
                ep.qinfo = sl->qinfo;
                ep.scores = sl->cscores;
                ep.n_rows = ivf->n_clusters;
                ep.nql = nb;
                ep.ld = ivf->centroids->ld;
                ep.dim = ivf->dim;
                ep.nq = nb;
                ep.metric = kMetricNegL2Sq;
                IVF_TRY(launch_exact_scan(ep, s));
            }
            if (ivf->n_clusters <= kRankMax) {
                uint32_t np2 = 2;
                while (np2 < ivf->n_clusters) np2 <<= 1;
                hipLaunchKernelGGL(ivf_rank_kernel, dim3(nb), dim3(1024), 0, s, sl->cscores, nb, ivf->n_clusters, np2, np, sl->probe_rows,
                                   sl->probe_count, sl->probe_rank);
            } else {
                IVF_TRY(launch_largek(sl->cscores, ivf->n_clusters, sl->ckeys, np, 0, sl->probe_rows, sl->probe_scores,
                                      sl->probe_count, s));
                IVF_TRY(hipMemsetAsync(sl->probe_rank, 0xFF, (size_t)ivf->n_clusters * 4, s));
                hipLaunchKernelGGL(ivf_probe_rank_kernel, dim3((np + 255) / 256), dim3(256), 0, s, sl->probe_rows, sl->probe_count,
                                   sl->probe_rank);
            }
            if (c_rows) {
                const uint64_t cw = (c_rows + 63) / 64;
                hipLaunchKernelGGL(ivf_range_mask_kernel, dim3((uint32_t)std::min<uint64_t>((cw + 3) / 4, 4096), nb), dim3(256), 0, s,
                                   ivf->list_off, ivf->n_clusters, sl->probe_rank, c_rows, sl->mask_c, (uint64_t)mask_words);
            }
            if (c_rows < n_rows) {
                const uint64_t n_words = (n_rows + 63) / 64;
                const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_words + 3) / 4, 4096);
                hipLaunchKernelGGL(ivf_mask_kernel, dim3(blocks, nb), dim3(256), 0, s, ivf->assign, sl->probe_rank, ivf->n_clusters, n_rows,
                                   c_rows, sl->mask, (uint64_t)mask_words);
            }
            IVF_TRY(hipGetLastError());
            // A lone query, nobody else searching: its list scans are enqueued HERE, on this stream, behind the bitmaps they read —
            // the probe order, both result lists and the counts come back in one copy each behind ONE wait (the call used to be two
            // round trips: 0.235 ms at 2M x 768 of which the scans' kernels are a third).  With other searches in flight the scans
            // go through the flat index's host path below instead, where the coalescer merges concurrent probes into batched sweeps.
            static const bool no_direct = getenv("NMN_IVF_NO_DIRECT") != nullptr;  // (A/B switch)
            const bool direct = nb == 1 && nq == 1 && searches_now <= 1 && !no_direct && kk1 <= NMN_MAX_TOP_K;
            const size_t part_bytes = ((size_t)kk1 * 12 + 4 + 15) & ~(size_t)15;
            if (direct) {
                if (sl->res_k < kk1) {
                    IVF_TRY(hipStreamSynchronize(s));
                    if (sl->res_dev) (void)hipFree(sl->res_dev);
                    if (sl->res_pin) (void)hipHostFree(sl->res_pin);
                    sl->res_dev = sl->res_pin = nullptr;
                    sl->res_k = 0;
                    IVF_TRY(hipMalloc(reinterpret_cast<void**>(&sl->res_dev), 2 * part_bytes));
                    IVF_TRY(hipHostMalloc(reinterpret_cast<void**>(&sl->res_pin), 2 * part_bytes, hipHostMallocDefault));
                    sl->res_k = kk1;
                }
                const size_t pb = (((size_t)sl->res_k * 12 + 4 + 15) & ~(size_t)15);
                auto part = [&](uint8_t* base, int i, uint64_t** r, float** sc, uint32_t** c) {
                    uint8_t* b = base + (size_t)i * pb;
                    *r = reinterpret_cast<uint64_t*>(b);
                    *sc = reinterpret_cast<float*>(b + (size_t)sl->res_k * 8);
                    *c = reinterpret_cast<uint32_t*>(b + (size_t)sl->res_k * 12);
                };
                // (this call waits for its answer: the list scans take the SHORT launch chain, nmn_index.h — a scan whose candidate
                //  list overflowed reports it in its count and is repeated below with the whole chain)
                auto enqueue_scans = [&](bool short_chain, bool want_c, bool want_t) -> nmn_status {
                    uint64_t* r;
                    float* sc;
                    uint32_t* c;
                    nmn_status e = NMN_OK;
                    if (c_rows && want_c) {
                        part(sl->res_dev, 0, &r, &sc, &c);
                        e = index_search_device(ivf->cvec, sl->qraw, 1, (uint32_t)kk1, kMetricNegL2, sl->mask_c, r, sc, c, s, short_chain);
                        if (e != NMN_OK) return e;
                    }
                    if (c_rows < n_rows && want_t) {
                        part(sl->res_dev, 1, &r, &sc, &c);
                        e = index_search_device(ivf->vectors, sl->qraw, 1, (uint32_t)kk1, kMetricNegL2, sl->mask, r, sc, c, s, short_chain);
                        if (e != NMN_OK) return e;
                    }
                    if (hipMemcpyAsync(sl->res_pin, sl->res_dev, 2 * pb, hipMemcpyDeviceToHost, s) != hipSuccess)
                        return set_error(NMN_ERR_STORAGE, "ivf: result copy");
                    return NMN_OK;
                };
                st = enqueue_scans(true, true, true);
                if (st != NMN_OK) return st;
                IVF_TRY(hipMemcpyAsync(probe_host_all, sl->probe_rows, (size_t)nb * ivf->n_clusters * 8, hipMemcpyDeviceToHost, s));
                IVF_TRY(hipStreamSynchronize(s));
                auto flagged = [&](int i) { return *reinterpret_cast<const uint32_t*>(sl->res_pin + (size_t)i * pb + (size_t)sl->res_k * 12) == 0xFFFFFFFFu; };
                const bool again_c = c_rows && flagged(0), again_t = c_rows < n_rows && flagged(1);
                if (again_c || again_t) {
                    if (again_c) index_short_chain_flagged(ivf->cvec);
                    if (again_t) index_short_chain_flagged(ivf->vectors);
                    st = enqueue_scans(false, again_c, again_t);
                    if (st != NMN_OK) return st;
                    IVF_TRY(hipStreamSynchronize(s));
                }
            } else {
                IVF_TRY(hipMemcpyAsync(probe_host_all, sl->probe_rows, (size_t)nb * ivf->n_clusters * 8, hipMemcpyDeviceToHost, s));
                IVF_TRY(hipStreamSynchronize(s));
            }
            if (direct) {
                const size_t pb = (((size_t)sl->res_k * 12 + 4 + 15) & ~(size_t)15);
                auto take = [&](int i, std::vector<uint64_t>& ids, std::vector<float>& dist, std::vector<uint32_t>& cnt) {
                    const uint8_t* b = sl->res_pin + (size_t)i * pb;
                    const uint64_t* r = reinterpret_cast<const uint64_t*>(b);
                    const float* sc = reinterpret_cast<const float*>(b + (size_t)sl->res_k * 8);
                    ids.assign(r, r + kk1);
                    dist.assign(sc, sc + kk1);
                    cnt.assign(1, std::min<uint32_t>(*reinterpret_cast<const uint32_t*>(b + (size_t)sl->res_k * 12), (uint32_t)kk1));
                };
                batched_c = batched_t = false;
                if (c_rows) {
                    take(0, bc_ids, bc_dist, bc_cnt);
                    batched_c = true;
                }
                if (c_rows < n_rows) {
                    take(1, bt_ids, bt_dist, bt_cnt);
                    batched_t = true;
                }
                if (stats) {  // (what the host path reports for the last list scan of a call)
                    st = nmn_index_last_stats(c_rows ? ivf->cvec : ivf->vectors, s, stats);
                    if (st != NMN_OK) return st;
                }
            }
            // 2a. the chunk's list scans, batched where that pays (results at k + 1, picked up query by query below)
            hint_c.assign(nb, 0);
            hint_t.assign(nb, 0);
            for (uint32_t i = 0; i < nb; i++) probed_of(probe_host_all + (size_t)i * ivf->n_clusters, &hint_c[i], &hint_t[i]);
            if (!direct) batched_c = batched_t = false;
            if (c_rows && !direct) {
                st = scan_many(ivf->cvec, sl->mask_c, hint_c, qh, nb, bc_ids, bc_dist, bc_cnt, &batched_c, (stats && q + nb == nq) ? stats : nullptr);
                if (st != NMN_OK) return st;
            }
            if (c_rows < n_rows && !direct) {
                st = scan_many(ivf->vectors, sl->mask, hint_t, qh, nb, bt_ids, bt_dist, bt_cnt, &batched_t,
                               (stats && q + nb == nq && !c_rows) ? stats : nullptr);
                if (st != NMN_OK) return st;
            }
        }
        const uint64_t* probe_host = probe_host_all + (size_t)qc * ivf->n_clusters;
        const uint64_t* mask_q = sl->mask + (size_t)qc * mask_words;
        const uint64_t* mask_cq = sl->mask_c + (size_t)qc * mask_words;
        // rows in the probed lists: the selectivity hint of the list scan (how the flat index's coalescer decides what may
        // run side by side) and the most the scan can return
        uint64_t probed_c = 0, probed_t = 0;
        probed_of(probe_host, &probed_c, &probed_t);
        // 2. masked scan ranking by distance (negated so that nearest = largest).  One result more than asked for: the scan
        //    breaks equal distances by id, the reference by candidate order, so a run of equal distances that straddles
        //    the cut must be seen whole before it is reordered and cut.
        uint64_t kk = kk1;
        uint32_t cnt = 0;
        std::vector<uint64_t> part_ids;
        std::vector<float> part_dist;
        for (;;) {
            const bool first = kk == kk1;  // (the batched scans of 2a answer the first round; an extension searches alone)
            tmp_ids.assign(kk, UINT64_MAX);
            tmp_dist.assign(kk, 0.f);
            if (c_rows) {  // the list-major copy: results are ITS rows, mapped back to ids
                if (first && batched_c) {
                    cnt = bc_cnt[qc];
                    std::copy(bc_ids.begin() + (size_t)qc * kk1, bc_ids.begin() + (size_t)(qc + 1) * kk1, tmp_ids.begin());
                    std::copy(bc_dist.begin() + (size_t)qc * kk1, bc_dist.begin() + (size_t)(qc + 1) * kk1, tmp_dist.begin());
                } else {
                    st = index_search_hostio(ivf->cvec, qh, 1, (uint32_t)kk, kMetricNegL2, mask_cq, true, tmp_ids.data(), tmp_dist.data(),
                                             &cnt, (stats && q + 1 == nq) ? stats : nullptr, probed_c);
                    if (st != NMN_OK) return st;
                }
                for (uint32_t i = 0; i < cnt; i++) tmp_ids[i] = ivf->vectors->row_base + ivf->perm_host[tmp_ids[i]];
            }
            if (c_rows < n_rows) {  // the younger vectors, in id order
                uint32_t cnt_t = 0;
                part_ids.assign(kk, UINT64_MAX);
                part_dist.assign(kk, 0.f);
                if (first && batched_t) {
                    cnt_t = bt_cnt[qc];
                    std::copy(bt_ids.begin() + (size_t)qc * kk1, bt_ids.begin() + (size_t)(qc + 1) * kk1, part_ids.begin());
                    std::copy(bt_dist.begin() + (size_t)qc * kk1, bt_dist.begin() + (size_t)(qc + 1) * kk1, part_dist.begin());
                } else {
                    st = index_search_hostio(ivf->vectors, qh, 1, (uint32_t)kk, kMetricNegL2, mask_q, true, part_ids.data(), part_dist.data(),
                                             &cnt_t, (stats && q + 1 == nq && !c_rows) ? stats : nullptr, probed_t);
                    if (st != NMN_OK) return st;
                }
                if (!c_rows) {
                    tmp_ids.swap(part_ids);
                    tmp_dist.swap(part_dist);
                    cnt = cnt_t;
                } else if (cnt_t) {
                    // merge the two lists (each best first: score = -distance descending, ties by id) and keep kk
                    std::vector<uint64_t> mi(kk, UINT64_MAX);
                    std::vector<float> md(kk, 0.f);
                    uint32_t a = 0, b = 0, o = 0;
                    while (o < kk && (a < cnt || b < cnt_t)) {
                        const bool take_a = b >= cnt_t || (a < cnt && (tmp_dist[a] > part_dist[b] || (tmp_dist[a] == part_dist[b] && tmp_ids[a] < part_ids[b])));
                        if (take_a) { mi[o] = tmp_ids[a]; md[o] = tmp_dist[a]; a++; }
                        else { mi[o] = part_ids[b]; md[o] = part_dist[b]; b++; }
                        o++;
                    }
                    tmp_ids.swap(mi);
                    tmp_dist.swap(md);
                    cnt = o;
                }
            }
            const bool cut_inside_run = cnt > k && tmp_dist[k] == tmp_dist[k - 1];
            // the run is seen whole as soon as it ENDS inside the fetched list (its last fetched entry differs from the
            // k-th); only a run that reaches the end of a full list may continue beyond it
            const bool run_ends_inside = cut_inside_run && tmp_dist[cnt - 1] != tmp_dist[k - 1];
            if (!cut_inside_run || run_ends_inside || cnt < kk || kk >= n_rows) break;  // nothing more to learn / fetch
            kk = std::min<uint64_t>(kk * 2, n_rows);
        }
        for (uint32_t i = 0; i < cnt; i++) tmp_dist[i] = -tmp_dist[i];
        // 3. equal distances keep candidate order: probe order of the cluster, then id (stable sort, ivf.rs:402)
        bool any_tie = false;
        for (uint32_t i = 1; i < cnt && !any_tie; i++) any_tie = tmp_dist[i] == tmp_dist[i - 1];
        if (any_tie) {
            rank_host.assign(ivf->n_clusters, kNoRank);
            for (uint32_t i = 0; i < np; i++)
                if (probe_host[i] < ivf->n_clusters) rank_host[probe_host[i]] = i;
            const uint64_t base = ivf->vectors->row_base;
            for (uint32_t a = 0; a < cnt;) {
                uint32_t b = a + 1;
                while (b < cnt && tmp_dist[b] == tmp_dist[a]) b++;
                if (b - a > 1)
                    std::sort(tmp_ids.begin() + a, tmp_ids.begin() + b, [&](uint64_t x, uint64_t y) {
                        const uint32_t rx = rank_host[ivf->assign_host[x - base]], ry = rank_host[ivf->assign_host[y - base]];
                        return rx != ry ? rx < ry : x < y;
                    });
                a = b;
            }
        }
        const uint32_t keep = std::min<uint32_t>(cnt, k);
        std::copy(tmp_ids.begin(), tmp_ids.begin() + keep, o_ids);
        std::copy(tmp_dist.begin(), tmp_dist.begin() + keep, o_dist);
        out_counts[q] = keep;
    }
    return NMN_OK;
}

// ---- persistence of the device layout ------------------------------------------------------------------
// File = Header{kind = ivf, dim, rows = vectors, aux = clusters} | centroids clusters x dim f32 | assign[] rows x u32 | a flat
// shard section (nmn_persist.hip) holding the vectors in id order.  A load re-creates the index WITHOUT re-running k-means or
// the list assignment: centroids and lists come back exactly as saved (the reference rebuilds its IVF index from the
// vectors after a restart; the trained centroids are the expensive part — tensor_store/src/ivf.rs:222-233).
#include "nmn_persist.h"

namespace nmn {
nmn_status persist_write_ivf(nmn_ivf* ivf, FILE* fp, const char* path) {
    std::unique_lock<std::shared_mutex> g(ivf->rw);
    const uint64_t rows = ivf->vectors->rows;
    std::vector<float> cents((size_t)ivf->n_clusters * ivf->dim);
    IVF_TRY(hipSetDevice(ivf->device));
    if (ivf->centroids_host.size() == cents.size()) cents = ivf->centroids_host;
    else
        IVF_TRY(hipMemcpy2D(cents.data(), (size_t)ivf->dim * 4, ivf->centroids->corpus, (size_t)ivf->centroids->ld * 4,
                            (size_t)ivf->dim * 4, ivf->n_clusters, hipMemcpyDeviceToHost));
    if (ivf->assign_host.size() < rows) return set_error(NMN_ERR_STORAGE, "IVF lists are not current");
    PersistHeader h{};
    memcpy(h.magic, "NMNIDX\0\1", 8);
    h.version = 1;
    h.kind = kPersistIvf;
    h.dim = ivf->dim;
    h.rows = rows;
    h.aux = ivf->n_clusters;
    if (fwrite(&h, sizeof h, 1, fp) != 1 || fwrite(cents.data(), 4, cents.size(), fp) != cents.size() ||
        (rows && fwrite(ivf->assign_host.data(), 4, rows, fp) != rows))
        return persist_io_error("cannot write", path);
    return persist_write_shard(ivf->vectors, fp, path);
}

// the section persist_write_ivf wrote, header already read into h
nmn_status persist_read_ivf(FILE* fp, const char* path, const PersistHeader& h, const nmn_index_desc* overrides, nmn_ivf** out) {
    *out = nullptr;
    if (h.kind != kPersistIvf || h.dim == 0 || h.aux == 0 || h.aux > 0xFFFFFFFFull)
        return set_error(NMN_ERR_SERIALIZATION, "not an IVF index file");
    const uint32_t n_clusters = (uint32_t)h.aux;
    {   // what the header announces must be in the file before anything is sized by it
        const long here = ftell(fp);
        long end = -1;
        if (here >= 0 && fseek(fp, 0, SEEK_END) == 0) end = ftell(fp);
        if (here < 0 || end < here || fseek(fp, here, SEEK_SET) != 0) return set_error(NMN_ERR_IO, "IO error: cannot seek in the index file");
        const uint64_t left = (uint64_t)(end - here);
        if ((uint64_t)n_clusters * h.dim > left / 4ull || h.rows > left / 4ull)
            return set_error(NMN_ERR_SERIALIZATION, "index file truncated (centroids / lists)");
    }
    std::vector<float> cents((size_t)n_clusters * h.dim);
    std::vector<uint32_t> assign((size_t)h.rows);
    if (fread(cents.data(), 4, cents.size(), fp) != cents.size() || (h.rows && fread(assign.data(), 4, h.rows, fp) != h.rows))
        return set_error(NMN_ERR_SERIALIZATION, "index file truncated (centroids / lists)");
    for (uint32_t a : assign)
        if (a >= n_clusters) return set_error(NMN_ERR_SERIALIZATION, "index file corrupt: a list id is out of range");
    PersistHeader hv{};
    nmn_status st = persist_read_header(fp, path, &hv);
    if (st != NMN_OK) return st;
    // (the section must announce exactly the payload its shape implies — persist_read_header has checked that many bytes are
    //  in the file — before anything is allocated by that shape: payload_bytes = 0 would pass the size check with any rows)
    if (hv.kind != kPersistFlat || hv.dim != h.dim || hv.rows != h.rows ||
        hv.payload_bytes != hv.rows * (uint64_t)hv.dim * 4ull + hv.rows * 4ull)
        return set_error(NMN_ERR_SERIALIZATION, "index file header is inconsistent");
    nmn_index_desc d{};
    d.dim = h.dim;
    d.flags = overrides ? overrides->flags : 0;
    d.capacity_rows = std::max<uint64_t>(overrides ? overrides->capacity_rows : 0, std::max<uint64_t>(h.rows, 1));
    d.device = overrides ? overrides->device : -1;
    d.cand_cap = overrides ? overrides->cand_cap : 0;
    nmn_ivf* ivf = nullptr;
    st = ivf_new(&d, cents.data(), n_clusters, &ivf);
    if (st != NMN_OK) return st;
    // the vectors: stream the shard section into ivf->vectors (checksum + integrity check of the magnitudes)
    st = persist_read_rows_into(fp, hv, ivf->vectors);
    if (st == NMN_OK && h.rows) {
        hipError_t e = hipSetDevice(ivf->device);
        if (e == hipSuccess) e = hipMemcpy(ivf->assign, assign.data(), (size_t)h.rows * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = set_error_hip(e, "uploading the lists");
    }
    if (st != NMN_OK) {
        const std::string keep = nmn_last_error();
        nmn_ivf_destroy(ivf);
        return set_error(st, keep.c_str());
    }
    ivf->assign_host = std::move(assign);
    ivf->centroids_host = std::move(cents);
    ivf->list_sizes.assign(n_clusters, 0);
    for (uint32_t a : ivf->assign_host) ivf->list_sizes[a]++;
    ivf->list_sizes_rows = h.rows;
    st = ivf_relayout(ivf);  // (the file holds the vectors in id order; the list-major copy is derived like after a build)
    if (st != NMN_OK) {
        nmn_ivf_destroy(ivf);
        return st;
    }
    *out = ivf;
    return NMN_OK;
}
}  // namespace nmn

extern "C" nmn_status nmn_ivf_save(nmn_ivf* ivf, const char* path) {
    if (!ivf || !path) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    FILE* fp = fopen(path, "wb");
    if (!fp) return persist_io_error("cannot create", path);
    nmn_status st = persist_write_ivf(ivf, fp, path);
    if (fclose(fp) != 0 && st == NMN_OK) st = persist_io_error("cannot close", path);
    return st;
}

extern "C" nmn_status nmn_ivf_load(const char* path, const nmn_index_desc* overrides, uint64_t max_file_bytes,
                                   uint64_t max_entries, nmn_ivf** out) {
    if (!path || !out) return set_error(NMN_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    nmn_status st = persist_check_file_size(path, max_file_bytes, nullptr);
    if (st != NMN_OK) return st;
    FILE* fp = fopen(path, "rb");
    if (!fp) return persist_io_error("cannot open", path);
    PersistHeader h{};
    st = persist_read_header(fp, path, &h);
    if (st == NMN_OK) st = persist_check_entries(h.rows, max_entries);
    if (st == NMN_OK) st = persist_read_ivf(fp, path, h, overrides, out);
    fclose(fp);
    return st;
}
