// Scenarios for the HOST layer of libquicked_hip.so under sanitizers, on a machine without a GPU: the library's own
// qe_driver.hip (host half) / qe_stages.hip / qe_pool.h / qe_batch.h / qe_capi.cpp built with g++ against the fake HIP runtime
// of tests/native/hip_stub (kernels replaced by host stand-ins), driven through the C-ABI exactly like
// tests/test_gpu_pools.py drives the real library: rotation of queued runs and fetches, early finish and merged flows,
// thread churn on leased contexts, per-pair calls from many threads, reclaim under a device-memory budget, destroy with
// runs queued.  Built and run by tests/test_host_sanitizers.py with -fsanitize=thread and -fsanitize=address,undefined.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "quicked.h"
#include "quicked_batch.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "host_scenarios: %s failed at line %d\n", #cond, __LINE__); exit(1); } } while (0)

struct Pairs {
    std::string pp, tp;
    std::vector<int64_t> po, to;
    std::vector<int32_t> pl, tl;
    int64_t n = 0;
};
static Pairs make_pairs(int n, int len, unsigned seed) {
    Pairs P;
    P.n = n;
    unsigned x = seed * 2654435761u + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    for (int i = 0; i < n; ++i) {
        const int L = len - (int)(rnd() % 7);
        P.po.push_back((int64_t)P.pp.size()); P.to.push_back((int64_t)P.tp.size());
        for (int k = 0; k < L; ++k) { const char c = "ACGT"[rnd() & 3]; P.pp.push_back(c); P.tp.push_back((rnd() % 20) ? c : "ACGT"[rnd() & 3]); }
        P.pl.push_back(L); P.tl.push_back(L);
    }
    return P;
}
static quicked_batch_t* create(const Pairs& P) {
    return quicked_batch_create(P.n, P.pp.data(), P.po.data(), P.pl.data(), P.tp.data(), P.to.data(), P.tl.data());
}
static quicked_params_t params(quicked_algo_t algo, bool only_score) {
    quicked_params_t p = quicked_default_params();
    p.algo = algo; p.only_score = only_score;
    return p;
}
static void fetch_all(quicked_batch_t* b, int64_t n) {
    CHECK(quicked_batch_fetch(b) >= 0);
    std::vector<int32_t> sc(n), st(n);
    CHECK(quicked_batch_scores(b, sc.data(), st.data()) >= 0);
    const int64_t bytes = quicked_batch_cigar_bytes(b);
    if (bytes > 0) {
        std::vector<char> pool((size_t)bytes);
        std::vector<int64_t> off(n);
        CHECK(quicked_batch_cigars(b, pool.data(), off.data()) >= 0);
    }
}

// 1. one thread, several batch objects, queued runs of every kind in rotation, fetched out of order
static void scenario_rotation() {
    const Pairs P = make_pairs(300, 400, 1);
    std::vector<quicked_batch_t*> bs;
    for (int k = 0; k < 5; ++k) { bs.push_back(create(P)); CHECK(bs.back()); }
    const quicked_params_t pb = params(BANDED, true), pq = params(QUICKED, false), pw = params(WINDOWED, true), ph = params(HIRSCHBERG, false);
    const quicked_params_t pqs = params(QUICKED, true);                 // only_score: the score pass (one score per task + the fast flow's lists)
    for (int round = 0; round < 6; ++round) {
        for (size_t k = 0; k < bs.size(); ++k) {
            const quicked_params_t* p = (round + k) % 4 == 0 ? &pb : ((round + k) % 4 == 1 ? ((round & 1) ? &pqs : &pq) : ((round + k) % 4 == 2 ? &pw : &ph));
            CHECK(quicked_batch_run(bs[k], p, (round == 0) ? 1 : 0) >= 0);
        }
        for (size_t k = bs.size(); k-- > 0;) fetch_all(bs[k], P.n);
    }
    CHECK(quicked_batch_reload(bs[0], P.n, P.pp.data(), P.po.data(), P.pl.data(), P.tp.data(), P.to.data(), P.tl.data()) >= 0);
    CHECK(quicked_batch_run(bs[0], &pq, 0) >= 0);
    CHECK(quicked_batch_sync(bs[0]) >= 0);
    for (quicked_batch_t* b : bs) quicked_batch_destroy(b);             // one of them with a run queued and never fetched
}

// 2. QuickEd's fast flow with pairs that leave it: the early-finish threads race the caller's fetches; merged flows
static void scenario_early_finish(int skip_every, int objects, int rounds) {
    setenv("QE_STUB_SKIP_EVERY", std::to_string(skip_every).c_str(), 1);
    const Pairs P = make_pairs(256, 300, 2);
    std::vector<quicked_batch_t*> bs;
    // every third object runs with only_score: its queued runs are score passes inside the fast flow, whose left pairs go
    // through the same early-finish threads (a flow of their own: merged flows take runs of equal parameters)
    const quicked_params_t pq_cigar = params(QUICKED, false), pq_score = params(QUICKED, true);
    auto pq_of = [&](int k) -> const quicked_params_t* { return (k % 3 == 2) ? &pq_score : &pq_cigar; };
    for (int k = 0; k < objects; ++k) { bs.push_back(create(P)); CHECK(bs.back()); CHECK(quicked_batch_run(bs[k], pq_of(k), 1) >= 0); }
    std::atomic<int> next{0};
    std::thread fetcher([&] {                                          // fetches from another thread than the one that queues
        for (int i = 0; i < rounds * objects; ++i) {
            while (next.load() <= i) std::this_thread::yield();
            fetch_all(bs[i % objects], P.n);
        }
    });
    for (int r = 0; r < rounds; ++r)
        for (int k = 0; k < objects; ++k) {
            while (next.load() - (r * objects + k) < -objects + 1) std::this_thread::yield();
            // a batch object is queued again only after its last run was fetched
            while (r > 0 && next.load() < (r - 1) * objects + k + 1) std::this_thread::yield();
            CHECK(quicked_batch_run(bs[k], pq_of(k), 0) >= 0);
            next.store(r * objects + k + 1);
        }
    fetcher.join();
    int64_t st[4];
    CHECK(quicked_early_finish_stats(st) >= 0);
    printf("early finish (every %d-th task leaves the fast flow, %d objects): flows %lld over %lld batches, merged flows %lld over %lld batches\n",
           skip_every, objects, (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3]);
    for (quicked_batch_t* b : bs) quicked_batch_destroy(b);
    unsetenv("QE_STUB_SKIP_EVERY");
}

// 2b. quicked_pool_trim() from another thread while runs with deferred pairs are queued and fetched: the trim joins the
// library's idle early-finish threads (finisher_retire) -- a finisher that has just finished a job takes its batches' fin_mu
// for the bookkeeping, a caller that holds one of those may be waiting for the pool's mutex inside finisher_submit, and the
// trim holds that mutex while it joins: the three-way cycle of round 5's review (ADVICE medium).  A finisher is busy until
// its bookkeeping is done and the trim backs off while anything is busy.  The window of the old ordering was a few
// instructions wide (this scenario did not hit it in five runs of the old code): it is here as the stress of that path --
// hundreds of thousands of trims against finishers at work -- under ThreadSanitizer's lock-order checks, not as a detector.
static void scenario_trim_race(int rounds) {
    setenv("QE_STUB_SKIP_EVERY", "4", 1);
    const Pairs P = make_pairs(192, 300, 5);
    const int objects = 4;
    std::vector<quicked_batch_t*> bs;
    const quicked_params_t pq = params(QUICKED, false);
    for (int k = 0; k < objects; ++k) { bs.push_back(create(P)); CHECK(bs.back()); CHECK(quicked_batch_run(bs[k], &pq, 1) >= 0); }
    std::atomic<bool> stop{false};
    std::atomic<long> trims{0};
    std::thread trimmer([&] {
        while (!stop.load()) { CHECK(quicked_pool_trim() >= 0); ++trims; std::this_thread::yield(); }
    });
    for (int r = 0; r < rounds; ++r) {
        for (int k = 0; k < objects; ++k) CHECK(quicked_batch_run(bs[k], &pq, 0) >= 0);
        for (int k = 0; k < objects; ++k) fetch_all(bs[k], P.n);
    }
    stop.store(true);
    trimmer.join();
    printf("trim race: %d rounds of %d queued runs with deferred pairs, %ld trims from another thread\n", rounds, objects, trims.load());
    for (quicked_batch_t* b : bs) quicked_batch_destroy(b);
    unsetenv("QE_STUB_SKIP_EVERY");
}

// 2c. reload right behind a fetch: the caller's own fetch has done a queued run's deferred pairs, the early-finish thread that
// was handed the same run gets to its job late -- it looks the job over (which device, which flow) before it takes the batch's
// fin_mu and finds the run superseded -- while the caller is already inside quicked_batch_reload / quicked_batch_configure,
// which write the batch.  (Round 6's loaded ThreadSanitizer rounds found the finisher reading B.device there, one run in ~100
// of scenario 1; this scenario is that window, over and over.)
static void scenario_reload_race(int rounds) {
    setenv("QE_STUB_SKIP_EVERY", "3", 1);
    const Pairs P = make_pairs(128, 200, 7), Q = make_pairs(128, 200, 8);
    const quicked_params_t pq = params(QUICKED, false);
    std::vector<quicked_batch_t*> bs;
    for (int k = 0; k < 3; ++k) { bs.push_back(create(P)); CHECK(bs.back()); CHECK(quicked_batch_run(bs[k], &pq, 1) >= 0); }
    for (int r = 0; r < rounds; ++r)
        for (size_t k = 0; k < bs.size(); ++k) {
            CHECK(quicked_batch_run(bs[k], &pq, 0) >= 0);
            fetch_all(bs[k], P.n);
            const Pairs& N = ((r + (int)k) & 1) ? Q : P;
            CHECK(quicked_batch_reload(bs[k], N.n, N.pp.data(), N.po.data(), N.pl.data(), N.tp.data(), N.to.data(), N.tl.data()) >= 0);
            CHECK(quicked_batch_configure(bs[k], (r + (int)k) % 2, 0) >= 0);
        }
    for (quicked_batch_t* b : bs) quicked_batch_destroy(b);
    unsetenv("QE_STUB_SKIP_EVERY");
}

// 3. thread churn: short-lived threads with a batch each take over the contexts the ones before them left
static void scenario_churn(int threads, int alive) {
    std::atomic<int> done{0};
    for (int base = 0; base < threads; base += alive) {
        std::vector<std::thread> ts;
        for (int k = 0; k < alive && base + k < threads; ++k)
            ts.emplace_back([&, k] {
                const Pairs P = make_pairs(128 + 16 * k, 250, 10 + k);
                quicked_batch_t* b = create(P);
                CHECK(b);
                const quicked_params_t pq = params(QUICKED, false), pb = params(BANDED, true);
                CHECK(quicked_batch_run(b, &pb, 1) >= 0);
                CHECK(quicked_batch_run(b, &pq, 1) >= 0);
                CHECK(quicked_batch_run(b, &pq, 0) >= 0);
                fetch_all(b, P.n);
                quicked_batch_destroy(b);
                ++done;
            });
        for (std::thread& t : ts) t.join();
    }
    CHECK(done.load() == threads);
    int64_t ps[8];
    CHECK(quicked_pool_stats(ps) >= 0);
    printf("churn: %d threads, pool stats: held %lld B, reclaims %lld, sets %lld, sub-batches %lld, contexts %lld / leased %lld\n", threads,
           (long long)ps[0], (long long)ps[1], (long long)ps[2], (long long)ps[3], (long long)ps[6], (long long)ps[7]);
    CHECK(quicked_pool_trim() >= 0);
}

// 4. the per-pair drop-in ABI from several threads at once (one aligner per thread, as align_benchmark.c:246-284)
static void scenario_per_pair(int threads, int calls) {
    std::vector<std::thread> ts;
    for (int k = 0; k < threads; ++k)
        ts.emplace_back([&, k] {
            const Pairs P = make_pairs(calls, 200, 100 + k);
            for (int i = 0; i < calls; ++i) {
                quicked_params_t p = quicked_default_params();
                if (i % 3 == 1) { p.algo = BANDED; p.only_score = true; }
                quicked_aligner_t a;
                CHECK(quicked_new(&a, &p) >= 0);
                CHECK(quicked_align(&a, P.pp.data() + P.po[i], P.pl[i], P.tp.data() + P.to[i], P.tl[i]) >= 0);
                if (i % 5 == 0) CHECK(quicked_align(&a, "", 0, "ACGT", 4) == QUICKED_EMPTY_SEQUENCE);
                CHECK(quicked_free(&a) >= 0);
            }
        });
    for (std::thread& t : ts) t.join();
}

// 5. two threads whose pools do not fit the device together: the out-of-memory path reclaims, nobody fails
static void scenario_budget() {
    std::vector<std::thread> ts;
    for (int k = 0; k < 2; ++k)
        ts.emplace_back([&, k] {
            const Pairs P = make_pairs(2000, 900, 200 + k);
            quicked_batch_t* b = create(P);
            CHECK(b);
            const quicked_params_t pq = params(QUICKED, false);
            for (int r = 0; r < 4; ++r) { CHECK(quicked_batch_run(b, &pq, r == 0 ? 1 : 0) >= 0); fetch_all(b, P.n); }
            quicked_batch_destroy(b);
        });
    for (std::thread& t : ts) t.join();
    int64_t ps[8];
    CHECK(quicked_pool_stats(ps) >= 0);
    printf("budget: pools hold %lld B, allocations that needed a reclaim of level >= 2: %lld\n", (long long)ps[0], (long long)ps[1]);
    if (getenv("QE_STUB_EXPECT_RECLAIM")) CHECK(ps[1] > 0);
    CHECK(quicked_pool_trim() >= 0);
}

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "all";
    if (which == "all" || which == "rotation") scenario_rotation();
    if (which == "all" || which == "early") { scenario_early_finish(5, 3, 6); scenario_early_finish(16, 8, 4); }
    if (which == "all" || which == "trim") scenario_trim_race(12);
    if (which == "all" || which == "reload") scenario_reload_race(40);
    if (which == "all" || which == "churn") scenario_churn(18, 3);
    if (which == "all" || which == "perpair") scenario_per_pair(4, 40);
    if (which == "all" || which == "budget") scenario_budget();
    CHECK(quicked_pool_trim() >= 0);
    printf("host_scenarios ok (%s)\n", which.c_str());
    return 0;
}
