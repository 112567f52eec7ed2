// Concurrent callers of the C ABI on one resident shard (no engine, no Python):
//   g++ -O2 -std=c++17 -I include -o tools/micro/index_mt tools/micro/index_mt.cpp -L neumann_amd/lib -lneumann_gpu -lpthread -Wl,-rpath,$PWD/neumann_amd/lib
//   ./index_mt rows dim k seconds_per_point metric threads...
// Every thread calls nmn_index_search(nq = 1) in a loop with its own query; prints queries/s per thread count and the
// coalescing counters.  One query of every thread is checked against a single-threaded call (bit-equal).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "neumann_gpu.h"

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? atoll(argv[1]) : 10000000;
    const uint32_t dim = argc > 2 ? atoi(argv[2]) : 768;
    const uint32_t k = argc > 3 ? atoi(argv[3]) : 100;
    const double secs = argc > 4 ? atof(argv[4]) : 3.0;
    const nmn_metric metric = (nmn_metric)(argc > 5 ? atoi(argv[5]) : 0);
    nmn_index_desc d;
    memset(&d, 0, sizeof d);
    d.dim = dim;
    d.capacity_rows = rows;
    d.device = 0;
    nmn_index* idx = nullptr;
    if (nmn_index_create(&d, &idx) != 0) { printf("create failed: %s\n", nmn_last_error()); return 1; }
    if (nmn_index_fill_synthetic(idx, 20240601, 0, rows) != 0) { printf("fill failed: %s\n", nmn_last_error()); return 1; }
    const int max_threads_q = 512;
    std::vector<float> Q((size_t)max_threads_q * dim);
    nmn_synth_fill_host(Q.data(), 8, 0, max_threads_q, dim);
    // INDEX_MT_FILTER=B: an int column bucket = row % B; thread t searches WHERE bucket = t % B through
    // nmn_index_search_pred (predicate + search in one call)
    const int filt = getenv("INDEX_MT_FILTER") ? atoi(getenv("INDEX_MT_FILTER")) : 0;
    nmn_columns* cols = nullptr;
    uint32_t col = 0;
    if (filt) {
        if (nmn_columns_create(0, rows, &cols) != 0 || nmn_columns_add(cols, &col) != 0) { printf("columns failed: %s\n", nmn_last_error()); return 1; }
        const uint64_t chunk = 1 << 20;
        std::vector<uint8_t> kinds(chunk, NMN_CELL_INT);
        std::vector<uint64_t> pay(chunk);
        for (uint64_t r0 = 0; r0 < rows; r0 += chunk) {
            const uint64_t n = std::min(chunk, rows - r0);
            for (uint64_t i = 0; i < n; i++) pay[i] = (r0 + i) % (uint64_t)filt;
            if (nmn_columns_write(cols, col, r0, n, kinds.data(), pay.data()) != 0) { printf("columns write failed\n"); return 1; }
        }
        std::vector<uint64_t> ones((rows + 63) / 64, ~0ull);
        if (rows & 63) ones.back() = (1ull << (rows & 63)) - 1ull;
        if (nmn_columns_write_valid(cols, 0, ones.size(), ones.data()) != 0) { printf("valid failed\n"); return 1; }
    }
    auto one_search = [&](int t, uint64_t* rws, float* sc, uint32_t* c) -> int {
        if (!filt)
            return nmn_index_search(idx, Q.data() + (size_t)t * dim, 1, k, metric, nullptr, rws, sc, c, nullptr);
        nmn_pred_op op;
        memset(&op, 0, sizeof op);
        op.op = NMN_PRED_CMP;
        op.cmp = NMN_CMP_EQ;
        op.vkind = NMN_CELL_INT;
        op.column = col;
        op.a = (uint64_t)(t % filt);
        uint64_t selected = 0;
        return nmn_index_search_pred(idx, cols, &op, 1, nullptr, 0, Q.data() + (size_t)t * dim, 1, k, metric, rws, sc, c,
                                     &selected, nullptr);
    };
    const int max_threads = 512;
    std::vector<uint64_t> ref_rows((size_t)max_threads * k);
    std::vector<float> ref_scores((size_t)max_threads * k);
    uint32_t cnt = 0;
    for (int t = 0; t < max_threads; t += 37)  // reference answers, one caller at a time
        if (one_search(t, ref_rows.data() + (size_t)t * k, ref_scores.data() + (size_t)t * k, &cnt) != 0) { printf("search failed: %s\n", nmn_last_error()); return 1; }
    for (int a = 6; a < argc; a++) {
        const int nt = std::min(atoi(argv[a]), max_threads);
        std::atomic<long> done{0};
        std::atomic<int> bad{0};
        std::atomic<bool> stop{false};
        uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
        nmn_index_coalesce_stats(idx, &b0, &r0);
        std::vector<std::thread> th;
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                std::vector<uint64_t> rws(k);
                std::vector<float> sc(k);
                uint32_t c = 0;
                while (!stop) {
                    if (one_search(t, rws.data(), sc.data(), &c) != 0) { bad++; break; }
                    if (t % 37 == 0 && (memcmp(rws.data(), ref_rows.data() + (size_t)t * k, (size_t)k * 8) != 0 ||
                                        memcmp(sc.data(), ref_scores.data() + (size_t)t * k, (size_t)k * 4) != 0)) bad++;
                    done++;
                }
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        nmn_index_coalesce_stats(idx, &b1, &r1);
        printf("rows=%llu dim=%u k=%u metric=%d threads=%d: %.0f queries/s (%.3f ms per call), %llu merged batches carrying %llu calls, mismatches=%d\n",
               (unsigned long long)rows, dim, k, (int)metric, nt, done / dt, 1e3 * dt * nt / std::max<long>(done, 1),
               (unsigned long long)(b1 - b0), (unsigned long long)(r1 - r0), bad.load());
    }
    nmn_index_destroy(idx);
    return 0;
}
