// Concurrent callers of the C ABI's multi-GPU handle (nmn_sharded_*; no engine, no Python):
//   g++ -O2 -std=c++17 -I include -o tools/micro/sharded_mt tools/micro/sharded_mt.cpp -L neumann_amd/lib -lneumann_gpu -lpthread -Wl,-rpath,$PWD/neumann_amd/lib
//   ./sharded_mt rows dim k seconds_per_point n_shards threads...
// The shards go round-robin over the node's GPUs (SHARDED_MT_DEVICES="0,0,0,0": an explicit list, e.g. logical shards on one
// GPU).  Every thread calls nmn_sharded_search(nq = 1) in a loop with its own query; prints queries/s per thread count and
// the handle's coalescing counters.  Every 37th thread's answers are compared with a single-threaded call (bit-equal).
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
    const uint32_t n_shards = argc > 5 ? atoi(argv[5]) : 1;
    std::vector<int> devices;
    if (const char* e = getenv("SHARDED_MT_DEVICES"))
        for (const char* p = e; *p;) {
            devices.push_back(atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    nmn_sharded_desc d;
    memset(&d, 0, sizeof d);
    d.dim = dim;
    d.capacity_rows = rows;
    d.n_shards = n_shards;
    d.gather = NMN_GATHER_AUTO;
    d.devices = devices.size() == n_shards ? devices.data() : nullptr;
    nmn_sharded* s = nullptr;
    if (nmn_sharded_create(&d, &s) != 0) { printf("create failed: %s\n", nmn_last_error()); return 1; }
    if (nmn_sharded_fill_synthetic(s, 20240601, 0, rows) != 0) { printf("fill failed: %s\n", nmn_last_error()); return 1; }
    const int max_threads = 512;
    std::vector<float> Q((size_t)max_threads * dim);
    nmn_synth_fill_host(Q.data(), 8, 0, max_threads, dim);
    std::vector<uint64_t> ref_rows((size_t)max_threads * k);
    std::vector<float> ref_scores((size_t)max_threads * k);
    uint32_t cnt = 0;
    for (int t = 0; t < max_threads; t += 37)  // reference answers, one caller at a time
        if (nmn_sharded_search(s, Q.data() + (size_t)t * dim, 1, k, NMN_METRIC_COSINE, nullptr, ref_rows.data() + (size_t)t * k,
                               ref_scores.data() + (size_t)t * k, &cnt, nullptr) != 0) { printf("search failed: %s\n", nmn_last_error()); return 1; }
    printf("# %u shard(s), gather %s, devices:", n_shards, nmn_sharded_gather_mode(s) == NMN_GATHER_RCCL ? "RCCL all-gather" : "peer copies");
    for (uint32_t g = 0; g < n_shards; g++) printf(" %d", nmn_sharded_device(s, g));
    printf("\n");
    for (int a = 6; a < argc; a++) {
        const int nt = std::min(atoi(argv[a]), max_threads);
        std::atomic<long> done{0};
        std::atomic<int> bad{0};
        std::atomic<bool> stop{false};
        uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
        nmn_sharded_coalesce_stats(s, &b0, &r0);
        std::vector<std::thread> th;
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                std::vector<uint64_t> rws(k);
                std::vector<float> sc(k);
                uint32_t c = 0;
                while (!stop) {
                    if (nmn_sharded_search(s, Q.data() + (size_t)t * dim, 1, k, NMN_METRIC_COSINE, nullptr, rws.data(), sc.data(), &c, nullptr) != 0) { bad++; break; }
                    if (t % 37 == 0 && (memcmp(rws.data(), ref_rows.data() + (size_t)t * k, (size_t)k * 8) != 0 ||
                                        memcmp(sc.data(), ref_scores.data() + (size_t)t * k, (size_t)k * 4) != 0)) bad++;
                    done++;
                }
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        nmn_sharded_coalesce_stats(s, &b1, &r1);
        printf("rows=%llu dim=%u k=%u shards=%u threads=%d: %.0f queries/s (%.3f ms per call), %llu merged batches carrying %llu calls, mismatches=%d\n",
               (unsigned long long)rows, dim, k, n_shards, nt, done / dt, 1e3 * dt * nt / std::max<long>(done, 1),
               (unsigned long long)(b1 - b0), (unsigned long long)(r1 - r0), bad.load());
    }
    nmn_sharded_destroy(s);
    return 0;
}
