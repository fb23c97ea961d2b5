// qe_batch.h -- what the two translation units of the library share besides qe_pool.h: the batch object, the aligner's
// stage timers, and the entry points of the device side (qe_driver.hip: kernels, stage runners, QuickEd flows, early finish,
// batch loading) that the C-ABI (qe_capi.cpp) calls.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "quicked.h"
#include "quicked_batch.h"
#include "qe_types.h"
#include "qe_pool.h"

// ---------------------------------------------------------------------------
// A resident batch: ASCII pools + per-pair arrays + planes in one arena that
// lives as long as the batch; results of the last run.
// ---------------------------------------------------------------------------
struct quicked_batch {
    int64_t n = 0;
    int device = 0;
    std::vector<int32_t> p_len, t_len;
    std::vector<int64_t> p_off, t_off;            // ASCII offsets
    std::vector<int64_t> plp_off, plt_off;        // plane word offsets
    std::vector<int32_t> order;                   // task -> pair, sorted by length (ragged batches)
    // device, persistent
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    uint8_t *d_asc_p = nullptr, *d_asc_t = nullptr;
    int64_t *d_p_off = nullptr, *d_t_off = nullptr, *d_plp_off = nullptr, *d_plt_off = nullptr;
    int32_t *d_p_len = nullptr, *d_t_len = nullptr;
    // planes and flags are double-buffered by run parity (see Context)
    static constexpr int NP = qe::Context::NA;    // plane sets: one per run of this batch that may be on the device at once
    int np_alloc = 3;                             // how many of them this batch has (batch_load: 3 for large batches, more for small ones)
    qe::u64 *d_pl_p[NP] = {}, *d_pl_t[NP] = {}, *d_pl_pr[NP] = {}, *d_pl_tr[NP] = {};
    qe::u32* d_flags[NP] = {};
    int parity = 0;
    int np_used = 2;                              // plane sets in rotation = stream / pool sets in rotation (run_batch)
    int na_cap = 0;                               // the rotation depth this batch's first queued run was planned with: later runs do not go above it
    int last_parity = -1;                         // plane set of the last run queued (its end orders the next run's stash)
    size_t last_mat_bytes = 0;                    // fill matrices of this batch's last CIGAR run (all leaves at once)
    size_t last_fixed_bytes = 0;                  // everything else its align stage took from the pool (runs, strings, workspaces)
    int last_groups = 0;                          // 64-task groups of that stage
    int est_bound = 0;                            // QuickEd: the cutoff the next run's align buffers are sized for (0: none yet, < 0: classic flow only)
    hipEvent_t ev_done[NP] = {};    // end of the A phase of the last run that used this parity
    // the streams those events were last recorded on (qe_pool.h: StreamTag); read by early-finish threads that poll a run while
    // the caller queues the next one: atomic_load / atomic_store
    qe::StreamTagRef ev_done_tag[NP], ev_unpacked_tag;
    // waits for / queries ev_done[q] unless its stream has gone back to the runtime (the run is over then); hipErrorNotReady
    // only from the query form
    hipError_t done_sync(int q) const {
        std::shared_lock<std::shared_mutex> life(qe::g_stream_life);
        return qe::stream_gone(std::atomic_load(&ev_done_tag[q])) ? hipSuccess : hipEventSynchronize(ev_done[q]);
    }
    hipError_t done_query(int q) const {
        std::shared_lock<std::shared_mutex> life(qe::g_stream_life);
        return qe::stream_gone(std::atomic_load(&ev_done_tag[q])) ? hipSuccess : hipEventQuery(ev_done[q]);
    }
    hipError_t done_wait_on(hipStream_t s, int q) const {           // s waits for the run that used plane set q last
        std::shared_lock<std::shared_mutex> life(qe::g_stream_life);
        return qe::stream_gone(std::atomic_load(&ev_done_tag[q])) ? hipSuccess : hipStreamWaitEvent(s, ev_done[q], 0);
    }
    bool ev_done_set[NP] = {};
    size_t pl_p_words = 0, pl_t_words = 0;
    bool have_rev[NP] = {};
    // Results on the host, indexed by pair.  Two sets: the getters read res[vis]; whoever brings a run's results to the host
    // writes through `wr` -- the caller's own sync run / fetch into the visible set, an early-finish thread (qe::finisher_*)
    // into the other one, which the caller's quicked_batch_fetch then makes visible (shadow_ready).  So a queued run never
    // changes what the getters and the zero-copy views show until the caller fetches.
    // The CIGAR strings live in pinned host memory: one DMA from the device's string pool, no per-pair copies (a
    // 100 k x 10 kb batch has ~400 MB of them)
    struct PinnedBuf {
        char* p = nullptr; size_t size = 0, cap = 0;
        void reserve(size_t n) {
            if (n <= cap) return;
            const size_t ncap = std::max(n, cap + cap / 2 + 4096);
            char* q = nullptr;
            HIP_CHECK(hipHostMalloc((void**)&q, ncap, hipHostMallocDefault));
            if (size) memcpy(q, p, size);
            if (p) (void)hipHostFree(p);
            p = q; cap = ncap;
        }
        ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    };
    struct HostResults {
        std::vector<int32_t> score, status;
        std::vector<int64_t> cigar_off;
        PinnedBuf cigar_pool;
        std::vector<int32_t> check_ok;            // 1 valid, 0 not, -1 no alignment
        int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t deferred_pairs = 0;               // QuickEd: pairs that were aligned after the run (quicked_batch_deferred_pairs)
        void clear() { score.clear(); status.clear(); cigar_off.clear(); cigar_pool.size = 0; check_ok.clear(); deferred_pairs = 0; for (auto& c : counters) c = 0; }
    } res[2];
    int vis = 0;
    HostResults* wr = &res[0];
    bool shadow_ready = false;
    bool only_score_run = true;
    bool packed = false;                          // created from wire words: planes are the resident input, no ASCII, no k_pack
    int cigar_style = 0;                          // SegFormatArgs::style of the runs to come (quicked_batch_configure)
    bool check = false;                           // validate every CIGAR on the device (k_check_segs)
    int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // of the run being queued / fetched (copied to wr->counters at its end)
    // per-pair share of the counters [n][8], kept only by the object that stands in for the pairs of SEVERAL batch objects
    // in a merged early finish (qe::merged_finish): every batch gets exactly the counts of its own pairs back
    std::vector<int64_t> credit;
    void note_pair(int pair, int slot, int64_t amount) { if (!credit.empty() && pair >= 0) credit[(size_t)pair * 8 + (size_t)slot] += amount; }
    // results of the last run, device side, indexed by task (what a timed run leaves in HBM)
    int32_t* d_score = nullptr;
    bool pending = false;
    // what quicked_batch_fetch needs to bring the results of the last sync == 0 run to the host (qe::PendingFetch)
    std::shared_ptr<void> pending_fetch;
    // where a sync == 0 run leaves its results on the device until they are fetched: the batch's own memory, not the
    // queueing thread's rotating pools -- so that thread may queue as many further runs (of other batches) as it likes
    uint8_t* result_arena = nullptr;
    size_t result_bytes = 0;
    // wire words of a packed batch (device), kept so that a reload can reuse the arena
    int wire = 0;
    // A packed batch's wire words become planes in the first run after a (re)load, on that run's stream -- not in the
    // load: a load is then DMA only and never waits for a free SIMD on a chip that other runs keep full (a kernel of the
    // load used to queue behind 256-VGPR alignment waves that live for 20 ms: 80 ms per reload in bench.py's streaming leg)
    qe::u64 *d_wire_p = nullptr, *d_wire_t = nullptr;
    int64_t *d_wire_p_off = nullptr, *d_wire_t_off = nullptr;
    bool unpack_pending = false, unpack_event_set = false;
    hipEvent_t ev_unpacked = nullptr;
    // Early finish (qe::finisher_*): a QuickEd run queued with sync == 0 may leave pairs that need the host-driven stages;
    // a library thread aligns them as soon as the run is over instead of the caller's quicked_batch_fetch.  fin_mu is held
    // by whoever works on the batch object: an API call of the caller, or the finisher.
    std::mutex fin_mu;
    std::condition_variable fin_cv;
    int fin_jobs = 0;                             // finisher jobs submitted for this batch and not retired yet (under fin_mu)
    quicked_status_t fin_status = QUICKED_OK;     // what an early finish of the current results returned

    ~quicked_batch() {
        qe::device_free(arena, device);
        qe::device_free(result_arena, device);
        for (auto e : ev_done) if (e) (void)hipEventDestroy(e);
        if (ev_unpacked) (void)hipEventDestroy(ev_unpacked);
    }
};

namespace qe {

// the aligner's stage timers (quicked.h:61-66) while one of its calls is running
struct HostTimers { profiler_timer_t *windowed_s = nullptr, *windowed_l = nullptr, *banded = nullptr, *align = nullptr; };
inline thread_local HostTimers tl_timers;
// host timers the ABI exposes (profiler_timer.c:53-73, profiler_counter.c:46-66)
inline void qe_timer_start(profiler_timer_t* t) { if (!t) return; t->accumulated = 0; clock_gettime(CLOCK_REALTIME, &t->begin_timer); }
inline void qe_timer_stop(profiler_timer_t* t) {
    if (!t) return;
    struct timespec e;
    clock_gettime(CLOCK_REALTIME, &e);
    const uint64_t ns = (uint64_t)((e.tv_sec * 1000000000ll + e.tv_nsec) - (t->begin_timer.tv_sec * 1000000000ll + t->begin_timer.tv_nsec));
    t->accumulated += ns;
    profiler_counter_t* c = &t->time_ns;
    const uint64_t amount = t->accumulated;
    c->total += amount;
    ++c->samples;
    if (c->samples == 1) { c->min = amount; c->max = amount; c->m_oldM = (double)amount; c->m_newM = (double)amount; c->m_oldS = 0.0; }
    else {
        c->min = std::min(c->min, amount); c->max = std::max(c->max, amount);
        c->m_newM = c->m_oldM + ((double)amount - c->m_oldM) / (double)c->samples;
        c->m_newS = c->m_oldS + ((double)amount - c->m_oldM) * ((double)amount - c->m_newM);
        c->m_oldM = c->m_newM; c->m_oldS = c->m_newS;
    }
    t->accumulated = 0;
}

inline double now_ms() { return mono_ms(); }
inline bool trace_on() { return env_set("QE_TRACE"); }
#define QE_TRACE_POINT(name) do { if (qe::trace_on()) { double t__ = qe::now_ms(); fprintf(stderr, "[qe t%03d @%.1f] %-22s +%.3f ms\n", (int)(syscall(SYS_gettid) % 1000), t__, name, t__ - tr_last); tr_last = t__; } } while (0)

// ---- the device side (qe_driver.hip)
// dispatch on params->algo (quicked_align, quicked.c:405-437) for every pair of the batch; fetch: synchronous, results to the host
quicked_status_t run_batch(quicked_batch& B, const quicked_params_t& p, bool fetch);
// the results of the batch's last queued run to the host (quicked_batch_fetch)
quicked_status_t fetch_results(quicked_batch& B);
// (re)loads a batch object with n pairs: host-side layout, arena (kept when it is large enough), H2D
void batch_load(quicked_batch* B, Context& C, int64_t n,
                const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                const char* text_pool, const int64_t* text_off, const int32_t* text_len);
void batch_load_packed(quicked_batch* B, Context& C, int64_t n, int wire,
                       const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                       const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len);
// every run of the batch that is still on the device (queued by any thread) is over
void batch_quiesce(quicked_batch* B);
// joins the early-finish threads while none of them has work (quicked_pool_trim; they start again on demand)
void finisher_retire();
// device-side cigar_check_alignment of caller-provided strings against the batch's resident pairs (quicked_batch_validate)
quicked_status_t batch_validate(quicked_batch* B, Context& C, const char* cigar_pool, int64_t pool_bytes, const int64_t* cigar_off, int32_t* ok_out);
void early_finish_stats(int64_t stats_out[4]);

}  // namespace qe
