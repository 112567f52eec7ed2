// Multi-threaded throughput of the C++ VectorEngine mirror (no Python, no GIL):
//   g++ -O2 -std=c++17 -I include -o engine_mt tools/micro/engine_mt.cpp -L neumann_amd/lib -lneumann_gpu -lpthread -Wl,-rpath,$PWD/neumann_amd/lib
//   ./engine_mt rows dim k queries_per_thread threads...
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "neumann_engine.h"
#include "neumann_gpu.h"

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? atoll(argv[1]) : 1000000;
    const uint32_t dim = argc > 2 ? atoi(argv[2]) : 768;
    const uint64_t k = argc > 3 ? atoll(argv[3]) : 100;
    const int per = argc > 4 ? atoi(argv[4]) : 200;
    nmn_engine_config cfg;
    nmn_engine_config_default(&cfg);
    // ENGINE_MT_DEVICES="0,1,2,3": every collection mirror is one nmn_sharded index over these GPUs (an ordinal may repeat)
    if (const char* e = getenv("ENGINE_MT_DEVICES"))
        for (const char* p = e; *p && cfg.n_devices < NMN_ENGINE_MAX_DEVICES;) {
            cfg.devices[cfg.n_devices++] = atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    const int filtered = getenv("ENGINE_MT_FILTER") ? atoi(getenv("ENGINE_MT_FILTER")) : 0;
    nmn_engine* e = nullptr;
    if (nmn_engine_create(&cfg, &e) != 0) { printf("create failed: %s\n", nmn_engine_last_error()); return 1; }
    const uint64_t chunk = 50000;
    std::vector<float> buf((size_t)chunk * dim);
    std::vector<std::string> names(chunk);
    std::vector<const char*> keys(chunk);
    for (uint64_t r0 = 0; r0 < rows; r0 += chunk) {
        const uint64_t n = std::min(chunk, rows - r0);
        nmn_synth_fill_host(buf.data(), 7, r0, n, dim);
        for (uint64_t i = 0; i < n; i++) { names[i] = "k" + std::to_string(r0 + i); keys[i] = names[i].c_str(); }
        if (filtered) {
            // ENGINE_MT_FILTER=B: every row carries bucket = row % B; thread t searches WHERE bucket = t % B (pre-filter)
            for (uint64_t i = 0; i < n; i++) {
                nmn_meta_field mf;
                mf.name = "bucket";
                mf.value.kind = NMN_VAL_INT;
                mf.value.b = 0;
                mf.value.i = (int64_t)((r0 + i) % (uint64_t)filtered);
                mf.value.f = 0.0;
                mf.value.s = nullptr;
                if (nmn_engine_store_embedding_with_metadata(e, keys[i], buf.data() + i * dim, dim, &mf, 1) != 0) { printf("store failed\n"); return 1; }
            }
        } else if (nmn_engine_batch_store(e, keys.data(), buf.data(), n, dim) != 0) { printf("store failed\n"); return 1; }
    }
    std::vector<float> Q((size_t)64 * dim);
    nmn_synth_fill_host(Q.data(), 8, 0, 64, dim);
    nmn_results* r = nullptr;
    if (nmn_engine_search_similar(e, Q.data(), dim, k, &r) != 0) { printf("search failed: %s\n", nmn_engine_last_error()); return 1; }
    nmn_results_free(r);
    if (filtered) {  // first filtered search builds the metadata columns
        nmn_value v;
        v.kind = NMN_VAL_INT; v.b = 0; v.i = 0; v.f = 0.0; v.s = nullptr;
        nmn_filter* f0 = nmn_filter_cmp(NMN_OP_EQ, "bucket", &v);
        nmn_filtered_config fc;
        nmn_filtered_config_default(&fc);
        fc.strategy = NMN_FILTER_PRE;
        if (nmn_engine_search_similar_filtered(e, Q.data(), dim, k, f0, &fc, &r) != 0) { printf("filtered search failed: %s\n", nmn_engine_last_error()); return 1; }
        nmn_results_free(r);
        nmn_filter_free(f0);
    }
    for (int a = 5; a < argc; a++) {
        const int nt = atoi(argv[a]);
        std::atomic<int> bad{0};
        std::atomic<bool> stop{false};
        std::atomic<long> writes{0};
        // ENGINE_MT_WRITERS=n: n threads keep overwriting existing keys and storing / deleting extra ones meanwhile
        const int writers = getenv("ENGINE_MT_WRITERS") ? atoi(getenv("ENGINE_MT_WRITERS")) : 0;
        std::vector<std::thread> wr;
        for (int w = 0; w < writers; w++)
            wr.emplace_back([&, w] {
                std::vector<float> v(dim, 0.5f);
                for (long i = 0; !stop; i++) {
                    const std::string extra = "extra" + std::to_string(w) + "_" + std::to_string(i % 64);
                    const std::string existing = "k" + std::to_string((i * 7919 + w) % rows);
                    v[i % dim] = (float)(i % 13) - 6.0f;
                    if (nmn_engine_store_embedding(e, extra.c_str(), v.data(), dim) != 0) bad++;
                    if (nmn_engine_store_embedding(e, existing.c_str(), v.data(), dim) != 0) bad++;
                    if (i % 3 == 2 && nmn_engine_delete_embedding(e, extra.c_str()) != 0) bad++;
                    writes += 2;
                }
            });
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                nmn_filter* flt = nullptr;
                nmn_filtered_config fc;
                nmn_filtered_config_default(&fc);
                fc.strategy = NMN_FILTER_PRE;
                if (filtered) {
                    nmn_value v;
                    v.kind = NMN_VAL_INT;
                    v.b = 0;
                    v.i = t % filtered;
                    v.f = 0.0;
                    v.s = nullptr;
                    flt = nmn_filter_cmp(NMN_OP_EQ, "bucket", &v);
                }
                for (int i = 0; i < per; i++) {
                    nmn_results* rr = nullptr;
                    const float* q = Q.data() + (size_t)((t * per + i) % 64) * dim;
                    const nmn_status st = filtered ? nmn_engine_search_similar_filtered(e, q, dim, k, flt, &fc, &rr)
                                                   : nmn_engine_search_similar(e, q, dim, k, &rr);
                    if (st != 0 || nmn_results_len(rr) < std::min<uint64_t>(k, filtered ? rows / filtered : rows)) bad++;
                    nmn_results_free(rr);
                }
                if (flt) nmn_filter_free(flt);
            });
        for (auto& x : th) x.join();
        stop = true;
        for (auto& x : wr) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("rows=%lu dim=%u k=%lu threads=%d writers=%d (%ld writes): %.0f queries/s (%.3f ms per query per thread)%s\n",
               (unsigned long)rows, dim, (unsigned long)k, nt, writers, writes.load(), nt * per / dt, dt / per * 1e3,
               bad ? "  ERRORS" : "");
    }
    nmn_engine_destroy(e);
    return 0;
}
