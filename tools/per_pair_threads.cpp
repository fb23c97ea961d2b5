// per_pair_threads.cpp -- the reference's parallel mode over its per-pair ABI (align_benchmark.c:246-284: T OpenMP threads,
// benchmark_edit.c:45-87: quicked_new / quicked_align / quicked_free per pair) against libquicked_hip.so, without a Python
// interpreter in the way:  per_pair_threads LENGTH ALGO(banded|quicked) CALLS_PER_THREAD [THREADS]
//   g++ -O2 -std=c++17 -pthread tools/per_pair_threads.cpp -Iinclude -Lquicked_amd -lquicked_hip -Wl,-rpath,$PWD/quicked_amd -o tools/bin/per_pair_threads
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include "quicked.h"

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "20", 0);
    const int len = argc > 1 ? atoi(argv[1]) : 1000;
    const bool quick = argc > 2 && strcmp(argv[2], "quicked") == 0;
    const int calls = argc > 3 ? atoi(argv[3]) : 400;
    // pairs: random text, pattern = text with 5 % substitutions
    std::vector<std::string> pats, txts;
    unsigned x = 12345;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int i = 0; i < 64; ++i) {
        std::string t(len, 'A'), p;
        for (auto& c : t) c = "ACGT"[rnd() & 3];
        p = t;
        for (int e = 0; e < len / 20; ++e) p[rnd() % len] = "ACGT"[rnd() & 3];
        pats.push_back(p); txts.push_back(t);
    }
    quicked_params_t params = quicked_default_params();
    if (!quick) { params.algo = BANDED; params.only_score = true; }
    auto once = [&](int i) {
        quicked_aligner_t a;
        quicked_new(&a, &params);
        quicked_align(&a, pats[i % 64].data(), len, txts[i % 64].data(), len);
        const int s = a.score;
        quicked_free(&a);
        return s;
    };
    std::vector<int> want;
    for (int i = 0; i < 64; ++i) want.push_back(once(i));
    std::vector<int> threads = {1, 2, 4, 8, 16, 32};
    if (argc > 4) threads = {atoi(argv[4])};                        // one thread count only (profiling runs)
    for (int T : threads) {
        std::atomic<int> ready{0}, bad{0};
        std::atomic<bool> go{false};
        std::vector<std::thread> th;
        for (int k = 0; k < T; ++k) th.emplace_back([&, k] {
            for (int i = 0; i < 8; ++i) once(i);                  // this thread's context
            ++ready;
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < calls; ++i) if (once(i + k) != want[(i + k) % 64]) ++bad;
        });
        while (ready.load() < T) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true);
        for (auto& t : th) t.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d bp %s, %2d host threads x %d new+align+free each: %9.0f calls/s (%.3f ms per call and thread)%s\n", len,
               quick ? "QuickEd + CIGAR" : "BandEd score-only", T, calls, T * calls / dt, dt / calls * 1e3, bad.load() ? "  MISMATCH" : "");
        fflush(stdout);
    }
    return 0;
}
