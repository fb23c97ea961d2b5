// qe_pool.h -- device memory of libquicked_hip.so: bump pools, per-thread contexts on lease, the process-wide book.
// Header-only (inline functions and variables): included by the device translation unit (qe_driver.hip) and by the C-ABI
// translation unit (qe_capi.cpp), one instance of every variable in the library.
//
// Replaces mm_allocator (quicked_utils/src/mm_allocator.c:141-426) on this path.  The reference runs one aligner per host
// thread (tools/align_benchmark/align_benchmark.c:246-249), each with an arena that grows on demand and "cannot fail"
// (mm_allocator.c:251-334); quicked_free gives it back (quicked.c:371-375).  Here:
//
//   DevicePool   a bump allocator over a few hipMalloc'ed chunks; sizes are closed-form in plen / tlen / cutoff, so a
//                run repeats the request sequence of the run before and steady-state runs never allocate.
//   Context      what one host thread uses on one device: streams, pools, pinned stages.  Contexts are never destroyed;
//                a thread holds one ON LEASE.  When the thread ends, its lease ends (a host-side flag: a thread-local
//                destructor must not call into HIP) and the next thread that needs a context takes it over, pools and all.
//   the book     per device: the bytes every context's pools hold (exact: pools count their own chunks), what each has
//                planned to grow to, and a pressure counter.  A thread plans its pools before a run against the others'
//                entries; a context counts with its plan while it has a call in progress or runs on the device, with
//                what it holds otherwise.  No wall-clock windows.
//   out of memory  is a path, not an error: (0) this pool's untouched chunks, (1) pools of contexts without a lease,
//                (2) this thread's other pools, (3) pools of contexts whose threads have no call in progress (their
//                streams are drained first; the owner finds them empty and allocates again), (4) pressure: every context
//                shrinks to one pool set at its next run, and the allocation is retried for QE_OOM_WAIT_MS (10 s).
//                Every context has a `busy` mutex: its thread holds it for the length of an API call, a reclaiming thread
//                only ever try_locks it.
//   streams      are a budget too (the runtime's hardware queues): a context that is doing nothing gives its set / side
//                streams back before another thread creates one (retire_idle_streams); events recorded on a stream that
//                is gone are never touched again (StreamTag).
#pragma once
#include <hip/hip_runtime.h>

#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

namespace qe {

// ---------------------------------------------------------------------------
// errors: the C-ABI has no exception channel; a HIP failure is fatal for the
// call and reported as QUICKED_ERROR with the reason on stderr.
// ---------------------------------------------------------------------------
struct HipError { hipError_t e; const char* what; int line; };
#define HIP_CHECK(expr)                                                             \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) throw qe::HipError{e__, #expr, __LINE__};            \
    } while (0)

// ---------------------------------------------------------------------------
// The chip, as the launch-shape rules see it: compute units and SIMDs of a device, read from the runtime once per device.
// Every "how many waves fill the chip" threshold of qe_stages.hip / qe_driver.hip is a multiple of these -- a launch is "one
// round of waves" while it has at most two waves per SIMD (the occupancy of the 158-246-VGPR alignment kernels) -- instead of
// a wave count that happens to fit the 256 CUs of one MI355X in SPX mode (a CPX partition has 32, other parts other counts).
// ---------------------------------------------------------------------------
struct Chip {
    int cus = 256, simds = 1024;
    size_t slots2() const { return (size_t)simds * 2; }       // wave slots at two waves per SIMD
    size_t frac2(double f) const { return (size_t)((double)simds * 2.0 * f); }
};
inline const Chip& chip(int device) {
    static Chip table[16];
    static std::once_flag once[16];
    const int d = (device >= 0 && device < 16) ? device : 0;
    std::call_once(once[d], [d] {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && cus > 0) { table[d].cus = cus; table[d].simds = cus * 4; }
        else (void)hipGetLastError();
    });
    return table[d];
}

// ---------------------------------------------------------------------------
// The library's switches.  Every QE_* environment variable it knows is read ONCE, the first time any of them is asked
// for, into one immutable table; the launch path only ever looks names up in that table (no getenv per stage call,
// nothing that races an embedding application's setenv).  Production needs none of them: they force kernel forms for the
// parity tests, lower thresholds so that small inputs reach deep paths, and switch traces on.  A name that is not in
// the list is a programming error.  quicked_debug_reload_env() (tests: the in-process suites change switches between
// runs) parses the environment again; it must not run concurrently with other calls into the library.
// ---------------------------------------------------------------------------
struct SwitchTable {
    static constexpr int N = 29;
    static constexpr const char* names[N] = {
        "QE_QUICKED_FAST", "QE_QUICKED_EST", "QE_FINISH_MERGE", "QE_FINISH_MERGE_PAIRS", "QE_FINISHERS", "QE_WAVE_PRIO", "QE_LANE_REL",
        "QE_COOP_G", "QE_COOP_FILL_G", "QE_COOP_LDS", "QE_WAVE", "QE_SCORE_SYS", "QE_STAGE3_DEVICE", "QE_FORMAT_WAVE", "QE_WINDOWED_CP",
        "QE_WINDOWED_QUAD", "QE_WINDOWED_SYS", "QE_SPLIT_BYTES", "QE_FILL_SYS", "QE_COOP_TALL_FILL", "QE_FILL_MULTI", "QE_TRACE_SYS",
        "QE_TRACE", "QE_TRACE_POOL", "QE_OOM_WAIT_MS", "QE_SCORE_WAVES", "QE_QUICKED_SCORE_PASS", "QE_QUICKED_SCORE_PASS_FAST", "QE_SCORE_PASS_COOP_G"};
    bool set[N];
    long long value[N];
    SwitchTable() {
        for (int i = 0; i < N; ++i) {
            const char* e = getenv(names[i]);
            set[i] = e != nullptr;
            value[i] = e ? strtoll(e, nullptr, 10) : 0;
        }
    }
};
inline std::atomic<const SwitchTable*> g_switches{nullptr};
inline const SwitchTable& switches() {
    const SwitchTable* t = g_switches.load(std::memory_order_acquire);
    if (!t) {
        const SwitchTable* fresh = new SwitchTable();
        if (g_switches.compare_exchange_strong(t, fresh, std::memory_order_acq_rel)) t = fresh;
        else delete fresh;
    }
    return *t;
}
inline void switches_reload() { g_switches.store(new SwitchTable(), std::memory_order_release); }     // the old table stays (a reader may hold it): a few hundred bytes per reload, tests only
inline int switch_index(const char* name) {
    for (int i = 0; i < SwitchTable::N; ++i) if (strcmp(SwitchTable::names[i], name) == 0) return i;
    fprintf(stderr, "[quicked_hip] internal error: unknown switch %s\n", name);
    abort();
}
inline bool env_set(const char* name) { return switches().set[switch_index(name)]; }
inline long long env_ll(const char* name, long long dflt) { const SwitchTable& t = switches(); const int i = switch_index(name); return t.set[i] ? t.value[i] : dflt; }
inline int env_int(const char* name, int dflt) { return (int)env_ll(name, dflt); }
inline bool pool_trace() { return env_set("QE_TRACE_POOL"); }
inline double mono_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

// ---------------------------------------------------------------------------
// the book, per device
// ---------------------------------------------------------------------------
inline constexpr int QE_MAX_DEVICES = 16;
struct DeviceBook {
    std::atomic<size_t> held{0};            // bytes in pool chunks of all contexts of this device
    std::atomic<uint64_t> epoch{0};         // bumped whenever the library allocates or frees device memory: a cached hipMemGetInfo reading is good for one epoch
    std::atomic<uint64_t> pressure{0};      // bumped by a thread that is out of memory after every reclaim
    std::atomic<int64_t> oom_events{0};     // allocations that needed level >= 2 (quicked_pool_stats()[1]: the planner is there to keep this 0)
};
inline DeviceBook g_book[QE_MAX_DEVICES];

struct DevicePool;
struct Context;
// -> true if device memory went back.  level: see the header comment.  `keep`: a pool of the caller that must survive
inline bool reclaim(int device, int level, DevicePool* keep);
inline void oom_report(int device, size_t bytes, const DevicePool* pool);

// hipMalloc with the out-of-memory path.  `keep`: the pool that is being grown (its memory is in use by the current run)
inline void device_malloc(void** p, size_t bytes, int device, DevicePool* keep, const char* what, int line) {
    DeviceBook& bk = g_book[device];
    hipError_t e = hipMalloc(p, bytes);
    for (int level = 0; e == hipErrorOutOfMemory && level <= 3; ++level) {
        (void)hipGetLastError();
        if (level >= 2) ++bk.oom_events;
        if (reclaim(device, level, keep)) e = hipMalloc(p, bytes);
    }
    if (e == hipErrorOutOfMemory) {
        // other threads are in the middle of runs: ask them to shrink, and take what comes free
        (void)hipGetLastError();
        ++bk.pressure;
        const int wait_ms = env_int("QE_OOM_WAIT_MS", 10000);
        const double t_end = mono_ms() + wait_ms;
        if (pool_trace()) fprintf(stderr, "[qe-pool] out of memory for %.2f GB: waiting up to %d ms for other threads\n", bytes / 1e9, wait_ms);
        while (e == hipErrorOutOfMemory && mono_ms() < t_end) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            (void)reclaim(device, 1, keep);
            (void)reclaim(device, 3, keep);
            e = hipMalloc(p, bytes);
            if (e == hipErrorOutOfMemory) (void)hipGetLastError();
        }
    }
    if (e != hipSuccess) {
        if (e == hipErrorOutOfMemory) oom_report(device, bytes, keep);
        throw HipError{e, what, line};
    }
    ++bk.epoch;
}
inline void device_free(void* p, int device) {
    if (!p) return;
    (void)hipFree(p);
    ++g_book[device].epoch;
}

// ---------------------------------------------------------------------------
// Device pool: chunks carved by a bump pointer, reset per run.  No device-side malloc, no per-pair hipMalloc.
// ---------------------------------------------------------------------------
struct DevicePool {
    // Chunks keep every pointer handed out during a run valid: when a run needs more than the arena holds, another chunk
    // is allocated.  The next run repeats the same request sequence and fits the same chunks.
    struct Chunk { uint8_t* base; size_t cap; bool used; int idle = 0; };     // used: something was carved from it since the last reset; idle: resets in a row it was not
    std::vector<Chunk> chunks;
    size_t cur = 0, top = 0, cap = 0;            // cap = total bytes over all chunks
    int device = 0;
    std::atomic<size_t>* owner_held = nullptr;   // the context's entry in the book
    void account(size_t add, size_t sub) {
        cap += add; cap -= sub;
        if (owner_held) { *owner_held += add; *owner_held -= sub; }
        g_book[device].held += add; g_book[device].held -= sub;
    }
    void free_chunk(Chunk& c) { device_free(c.base, device); account(0, c.cap); c.base = nullptr; c.cap = 0; }
    // A request has to fit ONE chunk.  A pool whose chunks date from runs with smaller requests (another budget, other
    // pairs) skips them and asks for a new one; when the device has no room for that next to them, the chunks this run has
    // not touched go back first -- their slots stay in the list (empty), so marks taken earlier in the run stay valid.
    bool drop_unused_chunks() {
        bool any = false;
        for (auto& c : chunks)
            if (!c.used && c.base) {
                if (pool_trace()) fprintf(stderr, "[qe-pool %p] drop unused chunk %.2f GB\n", (void*)this, c.cap / 1e9);
                free_chunk(c); any = true;
            }
        return any;
    }
    void add_chunk(size_t bytes) {
        Chunk c; c.cap = bytes; c.base = nullptr; c.used = false; c.idle = 0;
        device_malloc((void**)&c.base, bytes, device, this, "hipMalloc(pool chunk)", __LINE__);
        if (pool_trace()) fprintf(stderr, "[qe-pool %p] + chunk %.2f GB (pool %.2f GB, %zu chunks)\n", (void*)this, bytes / 1e9, (cap + bytes) / 1e9, chunks.size() + 1);
        chunks.push_back(c);
        account(bytes, 0);
    }
    void reset() {
        if (chunks.size() > 24) {     // a run repeats its request sequence, so the same chunks fit again: fold only on runaway growth
            HIP_CHECK(hipDeviceSynchronize());
            const size_t total = cap;
            for (auto& c : chunks) if (c.base) free_chunk(c);
            chunks.clear();
            add_chunk(total);
        }
        // Chunks that date from another workload (their sizes fit none of this run sequence's requests) go back to the
        // device after 8 runs in a row without use when memory is short (less than a quarter free; hipFree waits for
        // everything in flight, which a session with room to spare should not pay for)
        const int idle_max = 8;
        for (auto& c : chunks) {
            if (c.used || !c.base) { c.idle = 0; continue; }
            if (++c.idle >= idle_max && chunks.size() > 1) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= total_b / 4) { c.idle = idle_max / 2; continue; }
                if (pool_trace()) fprintf(stderr, "[qe-pool %p] stale chunk %.2f GB goes back\n", (void*)this, c.cap / 1e9);
                free_chunk(c);
            }
        }
        chunks.erase(std::remove_if(chunks.begin(), chunks.end(), [](const Chunk& c) { return c.cap == 0; }), chunks.end());
        for (auto& c : chunks) c.used = false;
        cur = 0; top = 0;
    }
    struct Mark { size_t cur, top; };
    Mark mark() const { return Mark{cur, top}; }
    void release(Mark m) { cur = m.cur; top = m.top; }
    template <typename T> T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (chunks.empty()) add_chunk(std::max(bytes, (size_t)1 << 26));
        while (chunks[cur].cap - top < bytes) {
            if (cur + 1 == chunks.size())
                add_chunk(std::max(bytes + ((size_t)1 << 20), std::min(std::max(cap / 4, (size_t)1 << 26), (size_t)1 << 32)));
            ++cur; top = 0;
        }
        chunks[cur].used = true;
        T* p = (T*)(chunks[cur].base + top);
        top += bytes;
        return p;
    }
    // One allocation up front for a run whose needs are known (the planner's figures): bump requests then never grow the
    // pool 4 GB at a time (a 94 GB fill matrix used to cost ~25 hipMallocs and seconds on a batch's first runs)
    void reserve(size_t bytes) {
        size_t room = 0;
        for (size_t i = cur; i < chunks.size(); ++i) room += (i == cur) ? chunks[i].cap - top : chunks[i].cap;
        if (room >= bytes) return;
        add_chunk(bytes - room + ((size_t)64 << 20));
    }
    // Grow to the chunk list of a pool that serves the same request sequence (another set of the rotation), so that the
    // run which first uses this one does not pay for its allocations; skipped when memory is short.
    void mirror(const DevicePool& o) {
        for (size_t i = chunks.size(); i < o.chunks.size(); ++i) {
            size_t free_b = 0, total_b = 0;
            if (o.chunks[i].cap == 0) continue;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * o.chunks[i].cap) return;
            add_chunk(o.chunks[i].cap);
        }
    }
    // the caller has made sure nothing on the device still uses this pool's memory
    void release_all() {
        if (pool_trace() && cap) fprintf(stderr, "[qe-pool %p] release_all %.2f GB\n", (void*)this, cap / 1e9);
        for (auto& c : chunks) if (c.base) free_chunk(c);
        chunks.clear(); cur = 0; top = 0;
    }
    size_t largest_chunk() const { size_t m = 0; for (const auto& c : chunks) m = std::max(m, c.cap); return m; }
};

// carves a fixed arena (a batch's persistent buffers)
struct ArenaCarver {
    uint8_t* base; size_t top;
    template <typename T> T* take(size_t count) {
        T* p = (T*)(base + top);
        top += (count * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

// Pinned host staging for the small per-run uploads (task lists, layouts).  A hipMemcpyAsync from pageable memory
// returns only when the copy has been performed, i.e. when the stream has reached it -- which puts the host to sleep
// behind run k-2 every time it queues run k.  Staged through pinned memory the copy is truly asynchronous; a stage is
// reused only after the run that filled it is over (there are 2 x NA of them, so that run is NA runs back).
struct PinnedStage {
    struct Chunk { uint8_t* base; size_t cap; };
    std::vector<Chunk> chunks;
    size_t cur = 0, top = 0;
    hipEvent_t done = nullptr;          // end of the run that used this stage last
    bool pending = false;
    void reset() {
        if (pending) { HIP_CHECK(hipEventSynchronize(done)); pending = false; }
        cur = 0; top = 0;
    }
    uint8_t* take(size_t bytes) {
        bytes = (bytes + 63) & ~(size_t)63;
        if (chunks.empty() || chunks[cur].cap - top < bytes) {
            while (cur + 1 < chunks.size() && chunks[cur + 1].cap < bytes) ++cur;
            if (!chunks.empty() && cur + 1 < chunks.size()) { ++cur; top = 0; }
            else {
                Chunk c; c.cap = std::max(bytes, (size_t)8 << 20); c.base = nullptr;
                HIP_CHECK(hipHostMalloc((void**)&c.base, c.cap, hipHostMallocDefault));
                chunks.push_back(c); cur = chunks.size() - 1; top = 0;
            }
        }
        uint8_t* p = chunks[cur].base + top;
        top += bytes;
        return p;
    }
};

// ---------------------------------------------------------------------------
// Streams that go back to the runtime (Context::retire_streams) and the events recorded on them.  A batch object's
// ev_done / ev_unpacked and a context's ev_last are waited for or queried by OTHER threads, at any later time; the HIP
// runtime faults on an event whose stream has been destroyed (hipEventSynchronize from quicked_batch_destroy: one GPU-suite
// run in five).  Every stream has a tag that says whether it is alive, an event keeps the tag of the stream it was last
// recorded on, and whoever touches such an event from outside its context holds g_stream_life shared and skips the call
// when the stream is gone -- its work was drained before it was destroyed, so the event is complete.  Destroying streams
// takes the lock exclusively (try_lock: a thread that sits in a long wait only postpones the clean-up).
// ---------------------------------------------------------------------------
struct StreamTag { std::atomic<bool> alive{true}; };
using StreamTagRef = std::shared_ptr<StreamTag>;
inline std::shared_mutex g_stream_life;
inline bool stream_gone(const StreamTagRef& t) { return t && !t->alive.load(); }

// ---------------------------------------------------------------------------
// Context: what one host thread uses on one device.  Never destroyed; held on lease (see the header comment).
// ---------------------------------------------------------------------------
struct Context {
    int device = 0;
    // ---- the book's entry of this context (planned / wanted under g_ctx_mu; held is exact and atomic)
    std::atomic<size_t> held{0};
    size_t planned = 0, wanted = 0;
    // ---- lease and call state
    std::atomic<bool> leased{false};
    std::atomic<int> lessee_tid{0};
    std::mutex busy;                         // held by the lessee during an API call; try_locked by a reclaiming thread
    std::atomic<bool> in_call{false};
    uint64_t pressure_seen = 0;
    void* small_batch = nullptr;             // the batch object single quicked_align calls reuse (qe_driver.hip: align_pairs)
    // pinned memory the loads of small batches go through (batch_load: a single pair's strings and tables in one copy launch
    // instead of two synchronous pageable copies and six asynchronous ones, ~0.1 ms of a 0.26 ms call); free again when
    // batch_load returns (it ends with a stream synchronisation)
    uint8_t* small_pin = nullptr;
    size_t small_pin_cap = 0;
    std::atomic<bool> small_pin_set{false};      // readable without `busy`: a reclaim pass asks whether there is anything to free
    uint8_t* small_pinned(size_t bytes) {
        if (small_pin_cap < bytes) {
            if (small_pin) { (void)hipHostFree(small_pin); small_pin = nullptr; small_pin_cap = 0; }
            const size_t cap = std::max(bytes, (size_t)1 << 20);
            HIP_CHECK(hipHostMalloc((void**)&small_pin, cap, hipHostMallocDefault));
            small_pin_cap = cap;
            small_pin_set = true;
        }
        return small_pin;
    }
    void release_small_pinned() {          // (its users end with a stream synchronisation: nothing of it is in flight between calls)
        if (small_pin) { (void)hipHostFree(small_pin); small_pin = nullptr; small_pin_cap = 0; }
        small_pin_set = false;
    }
    void* merge_batch = nullptr;             // the stand-in object of merged early finishes (qe_driver.hip: merged_finish)
    bool util_pinned = false;                // the utility pool holds live data of the call in progress (merged_finish): not to be reclaimed

    // Two phases of a run use two pools so that consecutive runs pipeline:
    //   W: pack + bound stages (WindowEd, band doubling)      A: the BandEd kernels (score / fill / traceback / format)
    // Up to NA sets of {stream, W pool, A pool, pinned stages} rotate between the consecutive runs of a thread; a batch
    // object has as many plane sets.  Large batches use three (a 100 k-pair kernel nearly fills the chip: more in flight
    // only queue); small ones as many as it takes to keep ~2 waves on every SIMD (plan in run_batch): a 12.5 k-pair run
    // is 196 waves of ~11 ms each, the chip holds 2048.
    static constexpr int NA = 12;
    hipStream_t stream_w = nullptr;          // utility stream: loads, fetches, the validator -- everything outside a run
    hipStream_t stream_w2[NA] = {}, stream_a2[NA] = {};
    StreamTagRef tag_a2[NA], tag_x, ev_last_tag;      // see StreamTag
    hipStream_t stream = nullptr;            // where the current phase launches
    // fork-join helper: the reverse half passes of a Hirschberg level run next to the forward ones (two 5000-wave launches
    // on one stream each end in a tail of their own; side by side the chip stays full until both are nearly done)
    hipStream_t stream_x = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t side_stream() {
        if (!stream_x) {
            HIP_CHECK(hipStreamCreateWithFlags(&stream_x, hipStreamNonBlocking));
            tag_x = std::make_shared<StreamTag>();
            if (!ev_fork) HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            if (!ev_join) HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        return stream_x;
    }
    DevicePool pool_w, pool_w2[NA], pool_a2[NA];
    int ai = 0;                              // which set the current run uses
    PinnedStage stage[2 * NA];               // see PinnedStage
    int si = 0;
    bool staging = false;                    // uploads on the current A stream go through stage[si]
    // a pool had to take the others' memory: no rotation and no fast flow on this thread -- for a while.  A workload change in
    // the middle of a session (other pairs, another estimate: the pools' chunks stop fitting) ends in the same reclaim as a
    // device that is really too small; only the second keeps coming back.  tight_left runs with one set, then the plan is
    // tried again; every further event doubles the spell (16 .. 1024 runs).
    bool memory_tight = false;
    int tight_left = 0, tight_spell = 16;
    int last_na = 0, last_sub_batches = 0;   // what the planner chose for the last run (quicked_pool_stats)
    int in_flight = 1;                       // runs of this thread that may be on the device at once while the current one executes
    size_t pool_budget = 0;                  // bytes one A pool may hold in this run (plan_pools)
    size_t seen_free = 0, seen_total = 0;    // last hipMemGetInfo reading of this device ...
    uint64_t seen_epoch = ~(uint64_t)0;      // ... and the book's epoch it was taken in
    hipStream_t& sa() { return stream_a2[ai]; }
    DevicePool& pa() { return pool_a2[ai]; }
    hipStream_t& sw() { return stream_w2[ai]; }
    DevicePool& pw() { return pool_w2[ai]; }
    DevicePool* scratch_p = nullptr;         // the current phase's pool
    hipEvent_t ev_pack = nullptr, ev_stage = nullptr, ev_decided[NA] = {};
    bool decided_set[NA] = {};               // the set's last run's k_stage1_decide (stream A) still reads its W pool: the set's next W phase waits for it
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_last = nullptr;            // end of the last run this context queued (the book: "runs on the device")
    // HIP-event pairs around the BandEd launches of every run since the last collection (bench.py's roofline legs), by kind:
    // 0 score-only pass of a BANDED run or of QuickEd's stage 3, 1 fill, 2 Hirschberg half pass
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
    std::vector<int> kev_kind;
    size_t kev_used = 0;
    std::pair<hipEvent_t, hipEvent_t>* kernel_events(int kind) {
        if (kev_used >= 4096) return nullptr;
        if (kev_used == kev.size()) {
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
            kev.emplace_back(a, b);
            kev_kind.push_back(0);
        }
        kev_kind[kev_used] = kind;
        return &kev[kev_used++];
    }
    void phase_w() { ensure_set(ai); stream = sw(); scratch_p = &pw(); }
    void phase_a() { ensure_set(ai); stream = sa(); scratch_p = &pa(); }
    void phase_u() { stream = stream_w; scratch_p = &pool_w; }
    void sync_all() {
        HIP_CHECK(hipStreamSynchronize(stream_w));
        if (stream_x) HIP_CHECK(hipStreamSynchronize(stream_x));
        for (auto q : stream_a2) if (q) HIP_CHECK(hipStreamSynchronize(q));
    }
    // A set's stream is created when the rotation first reaches it: a thread that only loads and fetches (an uploader, a
    // fetcher) has the utility stream and nothing else, a thread that runs large batches four streams -- the runtime maps
    // all streams of the process onto GPU_MAX_HW_QUEUES hardware queues, and streams that share one serialise.
    // A run's W phase and A phase depend on each other (pack -> bound stages -> align step): one stream serves both;
    // consecutive runs are on different sets, that is where the overlap comes from.
    void ensure_set(int q);
    explicit Context(int dev) : device(dev) {
        pool_w.device = dev; pool_w.owner_held = &held;
        for (auto& q : pool_w2) { q.device = dev; q.owner_held = &held; }
        for (auto& q : pool_a2) { q.device = dev; q.owner_held = &held; }
    }
    void init() {
        if (stream_w) { if (!stream) phase_u(); return; }
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreateWithFlags(&stream_w, hipStreamNonBlocking));
        if (!ev_pack) {            // the events are made once; the utility stream again after retire_streams(true)
            HIP_CHECK(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ev_stage, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ev_last, hipEventDisableTiming));
            for (auto& e : ev_decided) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_CHECK(hipEventCreate(&ev0));
            HIP_CHECK(hipEventCreate(&ev1));
        }
        phase_u();
    }
    // has work on the device (queried by other threads: hipEventQuery on an event the owner may be re-recording is safe,
    // the runtime serialises event operations)
    bool runs_on_device() const {
        std::shared_lock<std::shared_mutex> life(g_stream_life);
        return runs_on_device_locked();
    }
    bool runs_on_device_locked() const {       // the caller holds g_stream_life (shared or exclusive)
        if (!ev_last || stream_gone(std::atomic_load(&ev_last_tag))) return false;
        const hipError_t e = hipEventQuery(ev_last);
        if (e == hipErrorNotReady) return true;
        if (e != hipSuccess) (void)hipGetLastError();
        return false;
    }
    // every stream drained, every pool but `keep` back to the device.  The caller holds `busy` (the owner inside a call, or a
    // reclaiming thread that got it with try_lock) and is bound to this context's device.
    bool release_pools(DevicePool* keep, bool also_current_w) {
        if (held.load() == 0) return false;
        bool freed = false;
        bool drained = false;
        auto drain = [&]() {
            if (drained) return true;
            if (stream_w && hipStreamSynchronize(stream_w) != hipSuccess) return false;      // (no streams at all: retired, drained then)
            if (stream_x && hipStreamSynchronize(stream_x) != hipSuccess) return false;
            for (auto q : stream_a2) if (q && hipStreamSynchronize(q) != hipSuccess) return false;
            drained = true;
            return true;
        };
        for (int q = 0; q < NA; ++q) {
            if (&pool_a2[q] != keep && pool_a2[q].cap != 0) { if (!drain()) return freed; pool_a2[q].release_all(); freed = true; }
            if (&pool_w2[q] != keep && pool_w2[q].cap != 0 && (also_current_w || q != ai)) { if (!drain()) return freed; pool_w2[q].release_all(); freed = true; }
        }
        if (&pool_w != keep && pool_w.cap != 0 && !util_pinned && (also_current_w || scratch_p != &pool_w)) { if (!drain()) return freed; pool_w.release_all(); freed = true; }
        return freed;
    }
    // The set streams and the side stream go back to the runtime (the utility stream and the events stay).  The HIP runtime
    // maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues and streams that share a queue serialise: the streams
    // of contexts whose threads have ended must not keep queues that a live thread's rotation needs (a bench line whose
    // uploader / fetcher threads had left 20 streams behind ran its 11-deep stream of small batches at 3.5 instead of
    // 6.3 M alignments/s).  The caller holds `busy`, has drained the streams (release_pools) and is bound to the device.
    // ... and holds g_stream_life exclusively (StreamTag): the events recorded on these streams are complete from here on
    void retire_streams(bool utility_too = false) {
        if (!stream_w) return;
        for (int q = 0; q < NA; ++q) {
            if (stream_a2[q]) { (void)hipStreamSynchronize(stream_a2[q]); (void)hipStreamDestroy(stream_a2[q]); if (tag_a2[q]) tag_a2[q]->alive = false; }
            stream_a2[q] = nullptr; stream_w2[q] = nullptr; decided_set[q] = false; tag_a2[q].reset();
        }
        if (stream_x) { (void)hipStreamSynchronize(stream_x); (void)hipStreamDestroy(stream_x); stream_x = nullptr; if (tag_x) tag_x->alive = false; tag_x.reset(); }
        kev_used = 0;              // the kernel-timing events of runs since the last collection were recorded on those streams
        // the pinned stages' events were recorded on those streams: complete (the streams were drained), and not to be
        // touched again (an event of a destroyed stream: see StreamTag)
        for (auto& st : stage) st.pending = false;
        phase_u();
        if (utility_too) {         // a context without a lease keeps no stream at all: init() makes the utility stream again
            (void)hipStreamSynchronize(stream_w); (void)hipStreamDestroy(stream_w);
            stream_w = nullptr; stream = nullptr;
        }
    }
    void go_tight() {
        memory_tight = true;
        tight_left = tight_spell;
        tight_spell = std::min(2 * tight_spell, 1024);
    }
};

// ---------------------------------------------------------------------------
// registry + leases
// ---------------------------------------------------------------------------
inline std::mutex g_ctx_mu;                         // the registry, every context's leased / planned / wanted
inline std::vector<Context*>& g_ctx_all = *new std::vector<Context*>;     // never destroyed: detached library threads outlive static destruction
inline thread_local Context* tl_ctx = nullptr;
inline thread_local int tl_device = 0;
inline thread_local int tl_bound_device = -1;
inline thread_local int tl_api_depth = 0;
inline thread_local std::vector<Context*> tl_locked;      // contexts whose `busy` this thread holds (released by the outermost ApiScope)

// the contexts this thread has on lease, one per device it has used.  Its destructor runs when the thread ends and must
// not call into HIP (the runtime's own per-thread state may be gone already): it ends the leases, nothing else.
struct LeaseList {
    std::vector<Context*> v;
    ~LeaseList() {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (Context* c : v) { c->planned = 0; c->wanted = 0; c->lessee_tid = 0; c->leased = false; }
    }
};
inline thread_local LeaseList tl_leases;

// every exported function that touches a context opens one; the outermost one closes the thread's call
struct ApiScope {
    ApiScope() { ++tl_api_depth; }
    ~ApiScope() {
        if (--tl_api_depth == 0) {
            for (Context* c : tl_locked) { c->in_call = false; c->busy.unlock(); }
            tl_locked.clear();
        }
    }
    ApiScope(const ApiScope&) = delete;
    ApiScope& operator=(const ApiScope&) = delete;
};

inline Context* lease_context(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    Context* best = nullptr;
    for (Context* c : g_ctx_all)
        if (c->device == device && !c->leased && (!best || c->held.load() > best->held.load())) best = c;
    if (!best) { best = new Context(device); g_ctx_all.push_back(best); }
    best->leased = true;
    best->lessee_tid = (int)syscall(SYS_gettid);
    best->planned = 0; best->wanted = 0;
    return best;
}

// the calling thread's context on tl_device, `busy` held until the outermost ApiScope closes
inline Context& ctx() {
    if (tl_api_depth <= 0) { fprintf(stderr, "[quicked_hip] internal error: ctx() outside an API scope\n"); abort(); }
    bool fresh_lease = false;
    if (!tl_ctx || tl_ctx->device != tl_device) {
        tl_ctx = nullptr;
        for (Context* c : tl_leases.v) if (c->device == tl_device) tl_ctx = c;
        if (!tl_ctx) {
            if (tl_device < 0 || tl_device >= QE_MAX_DEVICES) throw HipError{hipErrorInvalidDevice, "device index", __LINE__};
            tl_ctx = lease_context(tl_device);
            tl_leases.v.push_back(tl_ctx);
            fresh_lease = true;
        }
    }
    Context* C = tl_ctx;
    if (std::find(tl_locked.begin(), tl_locked.end(), C) == tl_locked.end()) {
        C->busy.lock();              // a reclaiming thread may hold it for the time it takes to drain and free this context's pools
        C->in_call = true;
        tl_locked.push_back(C);
    }
    if (fresh_lease) {
        // what the previous lessee left: its plan and its degraded mode are not this thread's.  Only with `busy` held: a
        // reclaiming or stream-retiring thread that try_lock'ed this context while it had no lessee may still be writing
        // the same fields (retire_streams resets kev_used) -- a fresh lease always reaches this point with tl_locked not
        // yet holding C, so the lock above has just been taken.
        C->memory_tight = false; C->tight_left = 0; C->tight_spell = 16;
        C->kev_used = 0; C->seen_total = 0; C->seen_epoch = ~(uint64_t)0;
        C->pressure_seen = g_book[tl_device].pressure.load();
    }
    if (tl_bound_device != tl_device) { HIP_CHECK(hipSetDevice(tl_device)); tl_bound_device = tl_device; }
    C->init();
    return *C;
}

// ---------------------------------------------------------------------------
// the plan: bytes this context's pools may hold together.  `wanted`: what they would grow to if the device were this
// thread's alone.  Another context counts with what it has planned -- up to an equal share of the device where the wishes
// add up to more than there is -- while it is at work (a call in progress, or runs on the device); an idle one with what
// its pools hold.  Contexts re-plan before every run, so a thread that had the device to itself is down to its share one
// run after a second one shows up.
// ---------------------------------------------------------------------------
inline size_t ledger_plan(Context* me, size_t free_now, size_t wanted, size_t* owed = nullptr) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    me->wanted = wanted;
    std::vector<Context*> active;
    for (Context* c : g_ctx_all)
        if (c != me && c->device == me->device && c->planned > c->held.load() && (c->in_call.load() || c->runs_on_device())) active.push_back(c);
    const double space = 0.92 * (double)(free_now + g_book[me->device].held.load());          // what all pools of this device may hold together
    double others = 0, growth = 0;
    for (Context* c : g_ctx_all) {
        if (c == me || c->device != me->device) continue;
        const double h = (double)c->held.load();
        double claim = h;
        if (std::find(active.begin(), active.end(), c) != active.end())
            claim = std::max(claim, std::min((double)c->planned, space / (double)(active.size() + 1)));
        others += claim;
        growth += claim - h;
    }
    const size_t mine = (size_t)std::max(space - others, 0.0);
    me->planned = std::min(mine, wanted);
    // what the others have planned to grow by and will not find free: only then does this context have to hand memory back
    // (a pool above its budget is left alone while nobody needs the room: giving back and re-allocating tens of GB because
    // another thread's plan came and went costs more than anything it could save)
    if (owed) *owed = (size_t)std::max(growth - 0.92 * (double)free_now, 0.0);
    return mine;
}

// pools of contexts nobody holds a lease on (their threads have ended): released when a plan could use the room
inline bool release_unleased(int device) { return reclaim(device, 1, nullptr); }
inline size_t unleased_held(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    size_t s = 0;
    for (Context* c : g_ctx_all) if (c->device == device && !c->leased) s += c->held.load();
    return s;
}

inline bool reclaim(int device, int level, DevicePool* keep) {
    Context* me = (tl_ctx && tl_ctx->device == device) ? tl_ctx : nullptr;
    if (level == 0) return keep ? keep->drop_unused_chunks() : false;
    if (level == 2) {
        // this thread's other pools (their runs are waited for); the current run's own W pool is still being read
        if (!me) return false;
        me->go_tight();
        if (pool_trace()) fprintf(stderr, "[qe-pool] reclaim: own pools of context %p (%.2f GB held)\n", (void*)me, me->held.load() / 1e9);
        return me->release_pools(keep, false);
    }
    // level 1: contexts without a lease; level 3: any context whose thread has no call in progress
    std::vector<Context*> list;
    { std::lock_guard<std::mutex> lk(g_ctx_mu); list = g_ctx_all; }
    bool freed = false;
    for (Context* c : list) {
        const bool pinned_left = !c->leased.load() && c->small_pin_set.load();      // an ended thread's pinned block (>= 1 MB) goes with its pools
        if (c == me || c->device != device || (c->held.load() == 0 && !pinned_left)) continue;
        if (level == 1 && c->leased.load()) continue;
        std::unique_lock<std::mutex> lk(c->busy, std::try_to_lock);
        if (!lk.owns_lock()) continue;
        if (level == 1 && c->leased.load()) continue;          // taken over meanwhile: its new thread waits for `busy`, and keeps the pools
        if (pool_trace()) fprintf(stderr, "[qe-pool] reclaim level %d: context %p (%s) gives %.2f GB back\n", level, (void*)c, c->leased.load() ? "idle thread" : "no lease", c->held.load() / 1e9);
        freed |= c->release_pools(nullptr, true);
        if (!c->leased.load()) {
            c->release_small_pinned();         // under `busy`; nothing of it is in flight between calls
            std::unique_lock<std::shared_mutex> life(g_stream_life, std::try_to_lock);
            if (life.owns_lock()) c->retire_streams(true);
        }
    }
    return freed;
}

// The set / side streams of every context that is doing nothing -- no lease, or a lease whose thread has no call in progress
// and no run on the device (an early-finish thread between jobs, a thread that has finished its batches) -- go back to the
// runtime (Context::retire_streams; pools stay; the owner creates them again when it next needs them): a stream that the
// calling thread is about to create then finds a hardware queue of its own
inline void retire_idle_streams(int device) {
    // lock order: g_ctx_mu is never taken while g_stream_life is held (ledger_plan holds g_ctx_mu and takes the life lock
    // shared through runs_on_device: the other order deadlocked a bench run of the round's proof)
    std::vector<Context*> list;
    { std::lock_guard<std::mutex> lk(g_ctx_mu); list = g_ctx_all; }
    std::unique_lock<std::shared_mutex> life(g_stream_life, std::try_to_lock);
    if (!life.owns_lock()) return;             // some thread is inside a call on an event of a set stream: another time
    for (Context* c : list) {
        if (c == tl_ctx || c->device != device || c->in_call.load()) continue;
        // the context's stream handles are its owner's to write (init, ensure_set): read only with its `busy` mutex held
        std::unique_lock<std::mutex> lk(c->busy, std::try_to_lock);
        if (!lk.owns_lock() || !c->stream_w) continue;
        bool any = c->stream_x != nullptr || !c->leased.load();
        for (auto q : c->stream_a2) any |= q != nullptr;
        if (!any || c->runs_on_device_locked()) continue;
        c->retire_streams(!c->leased.load());
    }
}

inline void Context::ensure_set(int q) {
    if (stream_a2[q]) return;
    retire_idle_streams(device);               // a new stream: first those that nobody is using
    HIP_CHECK(hipStreamCreateWithFlags(&stream_a2[q], hipStreamNonBlocking));
    stream_w2[q] = stream_a2[q];
    tag_a2[q] = std::make_shared<StreamTag>();
}

inline void oom_report(int device, size_t bytes, const DevicePool* pool) {
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    fprintf(stderr, "[quicked_hip] out of device memory: want %.2f GB, device %d has %.2f of %.2f GB free; pools hold %.2f GB",
            bytes / 1e9, device, f / 1e9, t / 1e9, g_book[device].held.load() / 1e9);
    if (pool) fprintf(stderr, "; this pool %.2f GB in %zu chunks", pool->cap / 1e9, pool->chunks.size());
    fprintf(stderr, "\n");
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (const Context* c : g_ctx_all)
        if (c->device == device)
            fprintf(stderr, "[quicked_hip]   context %p%s: %s, %s, holds %.2f GB, planned %.2f GB, wanted %.2f GB\n", (const void*)c,
                    c == tl_ctx ? " (this thread)" : "", c->leased.load() ? "leased" : "no lease", c->in_call.load() ? "in a call" : "idle",
                    c->held.load() / 1e9, c->planned / 1e9, c->wanted / 1e9);
}

}  // namespace qe
