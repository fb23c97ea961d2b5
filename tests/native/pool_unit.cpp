// Host-side unit test of qe_pool.h (no GPU, no HIP call): the lease of contexts, the per-device book's arithmetic (ledger_plan),
// the API scope's lock discipline.  Compiled and run by tests/test_pool_cpu.py.
#include "qe_pool.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <thread>

using namespace qe;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "pool_unit: %s failed at line %d\n", #cond, __LINE__); return 1; } } while (0)
static const double GB = 1e9;

static void set_held(Context* c, size_t bytes) {      // what a pool's account() does, without a pool
    g_book[c->device].held -= c->held.load();
    c->held = bytes;
    g_book[c->device].held += bytes;
}

int main() {
    // ---- 1. leases: a context is never destroyed; a thread that ends leaves it to the next one
    Context* first = nullptr;
    std::thread([&] { first = lease_context(0); tl_leases.v.push_back(first); }).join();
    CHECK(first != nullptr);
    CHECK(!first->leased.load());                                 // the thread-local destructor ended the lease
    Context* again = nullptr;
    std::thread([&] { again = lease_context(0); tl_leases.v.push_back(again); }).join();
    CHECK(again == first);                                        // ... and the next thread took the SAME context over
    { std::lock_guard<std::mutex> lk(g_ctx_mu); CHECK(g_ctx_all.size() == 1); }
    // two threads alive at once need two contexts; a third, later, reuses one of them; another device gets its own
    Context *a = nullptr, *b = nullptr, *c = nullptr, *d1 = nullptr;
    {
        std::atomic<Context*> ha{nullptr}, hb{nullptr};
        std::thread ta([&] { Context* x = lease_context(0); tl_leases.v.push_back(x); ha = x; while (!hb.load()) std::this_thread::yield(); });
        std::thread tb([&] { while (!ha.load()) std::this_thread::yield(); Context* x = lease_context(0); tl_leases.v.push_back(x); hb = x; });
        ta.join(); tb.join();
        a = ha; b = hb;
    }
    CHECK(a != b && (a == first || b == first));
    std::thread([&] { c = lease_context(0); d1 = lease_context(1); tl_leases.v.push_back(c); tl_leases.v.push_back(d1); }).join();
    CHECK(c == a || c == b);
    CHECK(d1 != a && d1 != b && d1->device == 1);
    { std::lock_guard<std::mutex> lk(g_ctx_mu); CHECK(g_ctx_all.size() == 3); }
    // the context that holds more memory is preferred (its pools are reused instead of stranded)
    set_held(a, 10 * (size_t)GB); set_held(b, 40 * (size_t)GB);
    Context* pick = nullptr;
    std::thread([&] { pick = lease_context(0); tl_leases.v.push_back(pick); }).join();
    CHECK(pick == b);

    // ---- 2. the book: what a planning context may hold
    // device 0: a holds 10 GB, b holds 40 GB, 200 GB free.  me = a.  b idle: it keeps what it holds, no more.
    a->leased = true; b->leased = true;
    size_t owed = 1;
    size_t mine = ledger_plan(a, 200 * (size_t)GB, 300 * (size_t)GB, &owed);
    CHECK(std::abs((double)mine - (0.92 * 250 * GB - 40 * GB)) < 1e6);      // 0.92 x (free + all pools) - what b holds
    CHECK(a->planned == mine && a->wanted == 300 * (size_t)GB && owed == 0);
    // b at work (a call in progress) and still growing towards a plan of 150 GB: it counts with that plan, up to an equal share
    b->planned = 150 * (size_t)GB; b->in_call = true;
    mine = ledger_plan(a, 200 * (size_t)GB, 300 * (size_t)GB, &owed);
    CHECK(std::abs((double)mine - (0.92 * 250 * GB - 0.92 * 250 * GB / 2)) < 1e6);                  // its claim: min(150, 230 / 2) = 115 GB
    CHECK(owed == 0);                                                   // b's growth (75 GB) fits what is free
    // the same with little free memory: b is owed room that is not there -- a has to hand some back
    mine = ledger_plan(a, 20 * (size_t)GB, 300 * (size_t)GB, &owed);
    const double space = 0.92 * 70 * GB, claim_b = std::max(40 * GB, std::min(150 * GB, space / 2));
    CHECK(std::abs((double)mine - std::max(space - claim_b, 0.0)) < 1e6);
    CHECK(std::abs((double)owed - std::max((claim_b - 40 * GB) - 0.92 * 20 * GB, 0.0)) < 1e6);
    // b idle again (no call, no event recorded: no run on the device): only what it holds counts, whatever it once planned
    b->in_call = false;
    mine = ledger_plan(a, 200 * (size_t)GB, 30 * (size_t)GB, &owed);
    CHECK(std::abs((double)mine - (0.92 * 250 * GB - 40 * GB)) < 1e6 && a->planned == 30 * (size_t)GB);
    // contexts of another device do not enter
    set_held(d1, 100 * (size_t)GB);
    CHECK(ledger_plan(a, 200 * (size_t)GB, 30 * (size_t)GB) == mine);
    CHECK(unleased_held(1) == 100 * (size_t)GB && unleased_held(0) == 0);

    // ---- 3. the API scope: `busy` is held from the first use of a context to the end of the OUTERMOST call
    {
        ApiScope outer;
        a->busy.lock(); a->in_call = true; tl_locked.push_back(a);      // what ctx() does
        {
            ApiScope inner;                                              // a nested entry point (quicked_align -> quicked_batch_run)
        }
        CHECK(!a->busy.try_lock());                                      // still held after the inner call returned
        CHECK(a->in_call.load());
    }
    CHECK(a->busy.try_lock());                                           // released by the outermost scope
    a->busy.unlock();
    CHECK(!a->in_call.load());
    // a reclaiming thread only ever try_locks: a context in a call is skipped, an idle one is taken
    {
        ApiScope call;
        b->busy.lock(); b->in_call = true; tl_locked.push_back(b);
        bool got = true;
        std::thread([&] { std::unique_lock<std::mutex> lk(b->busy, std::try_to_lock); got = lk.owns_lock(); }).join();
        CHECK(!got);
    }
    {
        bool got = false;
        std::thread([&] { std::unique_lock<std::mutex> lk(b->busy, std::try_to_lock); got = lk.owns_lock(); }).join();
        CHECK(got);
    }
    printf("pool_unit ok\n");
    return 0;
}
