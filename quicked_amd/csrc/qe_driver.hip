// qe_driver.hip -- host side of libquicked_hip.so: device pool, batch objects,
// the bound-and-align driver (run_quicked, quicked.c:163-306) as a staged batch
// pipeline, and the C-ABI (include/quicked.h, include/quicked_batch.h).
//
// Replaces, on this path: quicked/src/quicked.c (drivers), mm_allocator (by a
// HIP device-pool batch allocator), sequence_buffer (by the pooled batch format).
// There is no CPU alignment fallback anywhere in this file: every score and
// every CIGAR comes out of the kernels in qe_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "quicked.h"
#include "quicked_batch.h"
#include "qe_types.h"
#include "qe_pool.h"
#include "qe_kernels.hip"

#define QE_API extern "C" __attribute__((visibility("default")))

namespace qe {

// the aligner's stage timers (quicked.h:61-66) while one of its calls is running
struct HostTimers { profiler_timer_t *windowed_s = nullptr, *windowed_l = nullptr, *banded = nullptr, *align = nullptr; };
static thread_local HostTimers tl_timers;
static void qe_timer_start(profiler_timer_t* t);
static void qe_timer_stop(profiler_timer_t* t);

// one in-stream copy as a kernel (see k_copy_multi); both buffers are padded to 16 bytes (pool / arena / stage allocations are)
static void copy_kernel(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    CopyTable T;
    T.n = 1; T.dst[0] = (uint4*)dst; T.src[0] = (const uint4*)src; T.n_u4[0] = (int64_t)((bytes + 15) >> 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(256, (T.n_u4[0] + 255) / 256);
    hipLaunchKernelGGL(k_copy_multi, dim3(blocks, 1), dim3(256), 0, s, T);
}
template <typename T>
static void h2d(T* dst, const std::vector<T>& src, hipStream_t s) {
    if (src.empty()) return;
    const size_t bytes = src.size() * sizeof(T);
    Context* C = tl_ctx;
    if (C && C->staging && (s == C->sa() || s == C->sw() || s == C->stream_x)) {      // W-phase copies too: the run's A phase, whose end frees the stage, is behind them
        uint8_t* st = C->stage[C->si].take(bytes);
        memcpy(st, src.data(), bytes);
        copy_kernel(dst, st, bytes, s);                            // the device reads the pinned stage itself: no DMA engine involved
        return;
    }
    HIP_CHECK(hipMemcpyAsync(dst, src.data(), bytes, hipMemcpyHostToDevice, s));
}
template <typename T>
static void d2h(std::vector<T>& dst, const T* src, size_t n, hipStream_t s) {
    dst.resize(n);
    if (n) HIP_CHECK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, s));
}

// ---------------------------------------------------------------------------
// host mirror of the band geometry (bpm_banded.c:121-135) -- sizes only
// ---------------------------------------------------------------------------
struct HGeom { int cutoff, diff, prolog, ebb, ebb_local; };
static HGeom host_geometry(int m, int n, int cutoff_in) {
    HGeom g;
    const int kend = std::abs(n - m) + 1;
    g.cutoff = std::max(std::max(kend, cutoff_in), 65);
    g.diff = m - n;
    const int rel = (g.cutoff - std::abs(g.diff) + 1) / 2;
    if (g.diff >= 0) { g.prolog = (rel + 63) / 64; g.ebb = (rel + g.diff + 63) / 64 + 1 + g.prolog; }
    else { g.prolog = (rel - g.diff + 63) / 64; g.ebb = (rel + 63) / 64 + 1 + g.prolog; }
    g.ebb_local = (g.cutoff + 63) / 64 + 1;
    return g;
}

}  // namespace qe

using namespace qe;

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
    u64 *d_pl_p[NP] = {}, *d_pl_t[NP] = {}, *d_pl_pr[NP] = {}, *d_pl_tr[NP] = {};
    u32* d_flags[NP] = {};
    int parity = 0;
    int np_used = 2;                              // plane sets in rotation = stream / pool sets in rotation (run_batch)
    int last_parity = -1;                         // plane set of the last run queued (its end orders the next run's stash)
    size_t last_mat_bytes = 0;                    // fill matrices of this batch's last CIGAR run (all leaves at once)
    size_t last_fixed_bytes = 0;                  // everything else its align stage took from the pool (runs, strings, workspaces)
    int last_groups = 0;                          // 64-task groups of that stage
    int est_bound = 0;                            // QuickEd: the cutoff the next run's align buffers are sized for (0: none yet, < 0: classic flow only)
    hipEvent_t ev_done[NP] = {};    // end of the A phase of the last run that used this parity
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
    u64 *d_wire_p = nullptr, *d_wire_t = nullptr;
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

static PairView pair_view(const quicked_batch& B, bool reversed) {
    PairView v;
    v.asc_p = B.d_asc_p; v.asc_p_off = B.d_p_off; v.p_len = B.d_p_len;
    v.asc_t = B.d_asc_t; v.asc_t_off = B.d_t_off; v.t_len = B.d_t_len;
    const int q = B.parity;
    v.pl_p = reversed ? B.d_pl_pr[q] : B.d_pl_p[q]; v.pl_p_off = B.d_plp_off;
    v.pl_t = reversed ? B.d_pl_tr[q] : B.d_pl_t[q]; v.pl_t_off = B.d_plt_off;
    v.flags = B.d_flags[q];
    return v;
}

// ---------------------------------------------------------------------------
// Launch of a lane-per-alignment kernel: ngroups waves.  Such a kernel is bound by VALU issue per SIMD:
// one or two waves on a SIMD take the same time, three take 1.4 x as long (measured).  Left
// to the dispatcher, 1563 one-wave workgroups land three-deep on some SIMDs in a good share of the
// launches (27.4 vs 38.8 ms for the same kernel).  So the placement is made a matter of resources: a
// workgroup is 4 waves -- the CU spreads them one per SIMD -- and claims a third-plus of the CU's LDS, so
// at most two workgroups share a CU and no SIMD ever holds more than two of these waves, whichever
// kernels and streams they come from.  A second kernel on another stream then fills exactly the SIMD
// slots the first one left empty, at no cost to either.
// ---------------------------------------------------------------------------
template <typename Kernel, typename Args>
static void launch_groups(Context& C, Kernel kernel, const Args& args, size_t ngroups, int max_waves, size_t lds_per_wave, bool chain = false) {
    if (ngroups == 0) return;
    const int wpb = std::min(4, max_waves);
    const unsigned blocks = (unsigned)((ngroups + wpb - 1) / wpb);
    // 54 KB: 3 x 54 KB > 160 KB >= 2 x 54 KB, two workgroups per CU.  When the launches in flight have fewer workgroups than
    // the chip has CUs, 84 KB (one per CU): the dispatcher packs the workgroups of CONCURRENT small kernels two to a CU
    // while other CUs idle (three 49-workgroup launches in flight: 16.5 ms each at 54 KB, 11.7 ms at 84 KB, 11.4 ms alone)
    size_t pin = ((size_t)blocks * (size_t)std::max(1, C.in_flight) > 256) ? (size_t)54 * 1024 : (size_t)84 * 1024;
    // chain: a launch of few waves whose duration is one wave's serial chain (WindowEd on a few thousand long reads: 157
    // waves of 1563 windows each).  108 KB: no 54 KB workgroup fits beside it, so its waves have their SIMDs to themselves
    // instead of sharing them with the fill of the run before (config 4: the stage took 59 ms beside that fill, 36 alone)
    if (chain && (size_t)blocks * (size_t)std::max(1, C.in_flight) <= 128) pin = (size_t)108 * 1024;
    const size_t lds = std::max(pin, lds_per_wave * (size_t)wpb);
    static thread_local std::vector<std::pair<const void*, int>> configured;      // per host thread and device
    const void* fn = reinterpret_cast<const void*>(kernel);
    if (std::find(configured.begin(), configured.end(), std::make_pair(fn, tl_device)) == configured.end()) {
        HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured.emplace_back(fn, tl_device);
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * wpb), lds, C.stream, args);
    HIP_CHECK(hipGetLastError());             // a rejected launch (block shape, LDS) must not pass for zeroed results
}

static void launch_pack(quicked_batch& B, Context& C, bool reversed) {
    if (B.packed) {                       // forward planes are the input; reversed ones come from them
        if (!reversed) return;
        const int blocks = (int)((B.n + 3) / 4);
        RevArgs r;
        r.nseq = (int32_t)B.n;
        r.fwd = B.d_pl_p[0]; r.rev = B.d_pl_pr[0]; r.pl_off = B.d_plp_off; r.len = B.d_p_len;
        hipLaunchKernelGGL(k_reverse_planes, dim3(blocks), dim3(256), 0, C.stream, r);
        r.fwd = B.d_pl_t[0]; r.rev = B.d_pl_tr[0]; r.pl_off = B.d_plt_off; r.len = B.d_t_len;
        hipLaunchKernelGGL(k_reverse_planes, dim3(blocks), dim3(256), 0, C.stream, r);
        for (bool& h : B.have_rev) h = true;    // every set aliases the same buffers
        return;
    }
    PackArgs a;
    a.nseq = (int32_t)B.n;
    a.reverse = reversed ? 1 : 0;
    const int q = B.parity;
    a.flags = reversed ? nullptr : B.d_flags[q];
    const int blocks = (int)((B.n + 3) / 4);
    a.asc = B.d_asc_p; a.asc_off = B.d_p_off; a.len = B.d_p_len;
    a.planes = reversed ? B.d_pl_pr[q] : B.d_pl_p[q]; a.pl_off = B.d_plp_off;
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, C.stream, a);
    a.asc = B.d_asc_t; a.asc_off = B.d_t_off; a.len = B.d_t_len;
    a.planes = reversed ? B.d_pl_tr[q] : B.d_pl_t[q]; a.pl_off = B.d_plt_off;
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, C.stream, a);
}

// ---------------------------------------------------------------------------
// One stage = one task list (subset of pairs) laid out as 64-lane groups
// ---------------------------------------------------------------------------
struct TaskList {
    std::vector<int32_t> pair, p0, m, t0, n, cutoff, tfin;   // padded to a multiple of 64, pair = -1 in the padding
    int ngroups() const { return (int)(pair.size() / 64); }
    void push(int32_t pr, int32_t p0_, int32_t m_, int32_t t0_, int32_t n_, int32_t cut, int32_t tf) {
        pair.push_back(pr); p0.push_back(p0_); m.push_back(m_); t0.push_back(t0_); n.push_back(n_);
        cutoff.push_back(cut); tfin.push_back(tf);
    }
    void pad() { while (pair.size() % 64) push(-1, 0, 1, 0, 1, 0, 0); }
};

struct DevTasks {
    TaskView v;
    int32_t *pair, *p0, *m, *t0, *n, *cutoff, *tfin;
};
// one upload of raw bytes through the pinned stage of the run (or a plain async copy when staging is off)
static void h2d_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    Context* C = tl_ctx;
    if (C && C->staging && (s == C->sa() || s == C->sw() || s == C->stream_x)) {      // W-phase copies too: the run's A phase, whose end frees the stage, is behind them
        uint8_t* st = C->stage[C->si].take(bytes);
        memcpy(st, src, bytes);
        copy_kernel(dst, st, bytes, s);
        return;
    }
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
}
// the seven arrays of a task list in ONE device block and ONE copy (a copy costs ~5-10 us of host time whatever its size)
static DevTasks upload_tasks(const TaskList& L, Context& C) {
    DevTasks d;
    const size_t nt = L.pair.size();
    int32_t* blk = C.scratch_p->take<int32_t>(7 * nt);
    d.pair = blk; d.p0 = blk + nt; d.m = blk + 2 * nt; d.t0 = blk + 3 * nt; d.n = blk + 4 * nt; d.cutoff = blk + 5 * nt; d.tfin = blk + 6 * nt;
    static thread_local std::vector<int32_t> host;
    host.resize(7 * nt);
    const std::vector<int32_t>* src[7] = {&L.pair, &L.p0, &L.m, &L.t0, &L.n, &L.cutoff, &L.tfin};
    for (int q = 0; q < 7; ++q) memcpy(host.data() + q * nt, src[q]->data(), nt * sizeof(int32_t));
    h2d_bytes(blk, host.data(), 7 * nt * sizeof(int32_t), C.stream);
    d.v.ntasks = (int32_t)nt; d.v.pair = d.pair; d.v.p0 = d.p0; d.v.m = d.m; d.v.t0 = d.t0; d.v.n = d.n;
    d.v.cutoff = d.cutoff; d.v.tfin = d.tfin;
    return d;
}

// per-group workspace geometry of a BandEd launch
struct BandLayout {
    std::vector<int64_t> ws_off, mat_off, runs_off;
    std::vector<int32_t> nslots, nrows, nch, runs_cap;
    size_t ws_bytes = 0, mat_u4 = 0, runs_u32 = 0;
};
static BandLayout band_layout(const TaskList& L, bool fill, bool want_runs, bool tight_runs = false) {
    BandLayout B;
    const int ng = L.ngroups();
    B.ws_off.resize(ng); B.mat_off.resize(ng); B.runs_off.resize(ng);
    B.nslots.resize(ng); B.nrows.resize(ng); B.nch.resize(ng); B.runs_cap.resize(ng);
    for (int g = 0; g < ng; ++g) {
        int ns = 3, nr = 4, nch = 2, nmax = 1, cap = 2;
        for (int l = 0; l < 64; ++l) {
            const size_t t = (size_t)g * 64 + l;
            if (L.pair[t] < 0) continue;
            const HGeom G = host_geometry(L.m[t], L.n[t], L.cutoff[t]);
            const int nsl = fill ? G.ebb : G.ebb_local;
            const int nw = (L.m[t] + 63) / 64;
            ns = std::max(ns, nsl);
            nr = std::max(nr, nw + nsl + 4);
            nch = std::max(nch, L.n[t] / 64 + 3);
            nmax = std::max(nmax, L.n[t]);
            // an alignment with e edits has at most 2 e + 1 runs.  tight_runs: the cutoff is known to be >= the distance
            // (QuickEd's bound, Hirschberg's exact child distances), so e <= cutoff; k_traceback reports, instead of
            // storing, a path that has more.  A user-chosen bandwidth promises nothing (a 35 %-error pair aligns at
            // bandwidth 15 with 499 edits against a cutoff of 300): every op may be its own run
            const int64_t every = (int64_t)L.m[t] + L.n[t] + 2;
            cap = std::max(cap, (int)(tight_runs ? std::min<int64_t>(every, (int64_t)2 * G.cutoff + 8) : every));
        }
        // the fill records the band edges of every chunk as int16 (cf / cl): a band of more than 32 k blocks (a leaf of
        // ~14 Mb at 15 % bandwidth -- Hirschberg splits long before that) is refused, not silently truncated
        if (fill && ns > 32760) throw HipError{hipErrorInvalidValue, "band of more than 32760 blocks: not supported", __LINE__};
        B.nslots[g] = ns; B.nrows[g] = nr; B.nch[g] = nch; B.runs_cap[g] = cap;
        B.ws_off[g] = (int64_t)B.ws_bytes;
        size_t bytes = (size_t)2 * (ns + 1) * 64 * 8 + (size_t)nr * 64 * 4 + (size_t)2 * nch * 64 * 2;
        B.ws_bytes += (bytes + 255) & ~(size_t)255;
        B.mat_off[g] = (int64_t)B.mat_u4;
        if (fill) B.mat_u4 += (size_t)(QE_CPC + 1) * nch * ns * 64;        // checkpoints cp[QE_CPC nch][ns][64] + carry words hw[nch][ns][64]
        B.runs_off[g] = (int64_t)B.runs_u32;
        if (want_runs) B.runs_u32 += (size_t)cap * 64;
    }
    return B;
}

struct DevLayout {
    uint8_t* ws; int64_t *ws_off, *mat_off, *runs_off; int32_t *nslots, *nrows, *nch, *runs_cap; uint4* mat; u32* runs;
};
static DevLayout upload_layout(const BandLayout& B, Context& C) {
    DevLayout d;
    const size_t ng = B.ws_off.size();
    d.ws = C.scratch_p->take<uint8_t>(B.ws_bytes);
    d.mat = C.scratch_p->take<uint4>(B.mat_u4);
    d.runs = C.scratch_p->take<u32>(B.runs_u32);
    // three int64 + four int32 arrays per group: one device block, one copy
    uint8_t* blk = C.scratch_p->take<uint8_t>(ng * 40);
    d.ws_off = (int64_t*)blk; d.mat_off = d.ws_off + ng; d.runs_off = d.mat_off + ng;
    d.nslots = (int32_t*)(d.runs_off + ng); d.nrows = d.nslots + ng; d.nch = d.nrows + ng; d.runs_cap = d.nch + ng;
    static thread_local std::vector<uint8_t> host;
    host.resize(ng * 40);
    if (ng) {
        uint8_t* h = host.data();
        memcpy(h, B.ws_off.data(), ng * 8); memcpy(h + ng * 8, B.mat_off.data(), ng * 8); memcpy(h + ng * 16, B.runs_off.data(), ng * 8);
        memcpy(h + ng * 24, B.nslots.data(), ng * 4); memcpy(h + ng * 28, B.nrows.data(), ng * 4);
        memcpy(h + ng * 32, B.nch.data(), ng * 4); memcpy(h + ng * 36, B.runs_cap.data(), ng * 4);
        h2d_bytes(blk, h, ng * 40, C.stream);
    }
    return d;
}

struct TaskOut {   // device arrays per task
    int32_t *score, *first, *last, *posv, *hew, *nruns, *nops, *edits, *len;
    u32 *adv, *steps;
    int64_t* str_off;
};
static TaskOut take_out(Context& C, size_t nt) {
    TaskOut o;
    o.score = C.scratch_p->take<int32_t>(nt); o.first = C.scratch_p->take<int32_t>(nt); o.last = C.scratch_p->take<int32_t>(nt);
    o.posv = C.scratch_p->take<int32_t>(nt); o.hew = C.scratch_p->take<int32_t>(nt); o.nruns = C.scratch_p->take<int32_t>(nt);
    o.nops = C.scratch_p->take<int32_t>(nt); o.edits = C.scratch_p->take<int32_t>(nt); o.len = C.scratch_p->take<int32_t>(nt);
    o.adv = C.scratch_p->take<u32>(nt); o.steps = C.scratch_p->take<u32>(nt);
    o.str_off = C.scratch_p->take<int64_t>(nt + 1);
    // the work counters are summed over every slot of the list: padding slots (and tasks a kernel skips) count 0
    HIP_CHECK(hipMemsetAsync(o.adv, 0, nt * sizeof(u32), C.stream));
    HIP_CHECK(hipMemsetAsync(o.steps, 0, nt * sizeof(u32), C.stream));
    return o;
}

// ---------------------------------------------------------------------------
// Stage runners.  Each returns with its kernels enqueued on C.stream.
// ---------------------------------------------------------------------------
struct StageResult {
    std::vector<int32_t> score, hew, first, last, posv, nruns, nops, edits, len;
    std::vector<u32> adv, steps;
};

static uint64_t sum_u32(const std::vector<u32>& v) { uint64_t s = 0; for (u32 x : v) s += x; return s; }

// BandEd score-only over a task list (bpm_banded.c:791-964); the launch's device state stays
// addressable (Hirschberg reads the stopped bands)
struct ScoreLaunch {
    DevTasks T; DevLayout D; TaskOut O; size_t nt = 0;
    int G = 1;                                                          // >= 2: cooperative launch + fallback pass
    uint8_t* cws = nullptr; int64_t* c_off = nullptr; int32_t *c_ns = nullptr, *c_nr = nullptr, *c_nch = nullptr;
};
static BandState coop_state(const ScoreLaunch& S) {
    BandState b;
    b.G = S.G; b.ws = S.cws; b.g_ws_off = S.c_off; b.g_nslots = S.c_ns; b.g_nrows = S.c_nr; b.g_nch = S.c_nch;
    b.first = S.O.first; b.last = S.O.last; b.posv = S.O.posv; b.maxrow = S.O.len; b.abort = S.O.hew;
    return b;
}

static BandState band_state(const ScoreLaunch& S) {
    BandState b;
    b.G = 1; b.abort = nullptr;
    b.ws = S.D.ws; b.g_ws_off = S.D.ws_off; b.g_nslots = S.D.nslots; b.g_nrows = S.D.nrows; b.g_nch = S.D.nch;
    b.first = S.O.first; b.last = S.O.last; b.posv = S.O.posv; b.maxrow = S.O.len;   // O.len doubles as maxrow here
    return b;
}

static ScoreLaunch launch_banded_score(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int timed) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    const BandLayout lay = band_layout(L, false, false);
    S.T = upload_tasks(L, C);
    S.D = upload_layout(lay, C);
    S.O = take_out(C, S.nt);
    BandedArgs a;
    a.P = pair_view(B, reversed); a.T = S.T.v;
    a.ws = S.D.ws; a.g_ws_off = S.D.ws_off; a.g_nslots = S.D.nslots; a.g_nrows = S.D.nrows; a.g_nch = S.D.nch;
    a.mat = nullptr; a.g_mat_off = S.D.mat_off;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len;
    a.only_if = nullptr;
    a.lane_rel = env_int("QE_LANE_REL", 1);
    auto* ke = timed ? C.kernel_events(timed - 1) : nullptr;       // timed = kind + 1 (Context::kernel_events)
    if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
    launch_groups(C, k_banded<false>, a, L.ngroups(), 8, 0);
    if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
    return S;
}

// lanes per alignment for the cooperative score-only kernel: enough waves to fill the chip
// (>= ~4 per SIMD) while every lane keeps >= 2 band slots; QE_COOP_G overrides (0 / 1 = off)
static int coop_lanes(const TaskList& L, int in_flight = 1, bool fill = false) {
    const char* e = getenv(fill ? "QE_COOP_FILL_G" : "QE_COOP_G");
    int min_nsl = 1 << 30, n_max = 1;
    size_t live = 0;
    for (size_t t = 0; t < L.pair.size(); ++t) {
        if (L.pair[t] < 0) continue;
        ++live;
        const HGeom hg = host_geometry(L.m[t], L.n[t], L.cutoff[t]);
        min_nsl = std::min(min_nsl, fill ? hg.ebb : hg.ebb_local);
        n_max = std::max(n_max, L.n[t]);
    }
    if (live == 0) return 1;
    int G = 1;
    // Waves to aim for: ~700 for 10 kb reads (measured with the multi-slot one-lane kernel and overlapped runs:
    // 8 k / 16 k pairs are best at G = 4, 32 k at G = 2, 50 k and up at G = 1), more for longer reads, whose one-lane
    // latency grows with their length (100 kb half passes: 526 -> 430 ms from G = 8 to 32)
    const size_t target = std::min<size_t>(4096, (size_t)700 * (size_t)std::max(1, n_max / 10000));
    if (e) G = atoi(e);
    else
        while (G < 64 && ((live * G) / 64) * (size_t)std::max(1, in_flight) < target) G *= 2;      // runs in flight fill the chip together
    // the band-height test first + 2 < last must stay decidable G-2 chunks early: keep the band >= 3 G + 4 slots
    // the band-height test first + 2 < last must stay decidable G - 2 chunks early: a band of >= 3 G + 4 slots always is;
    // with 2 G + 4 a task whose band comes within G slots of its minimum height is flagged and recomputed by the one-lane
    // kernel -- rare, and worth it where the launch is short of waves anyway (4 000 pairs of 10 kb: 4.2 -> 2.8 ms with G = 8)
    while (G > 1 && min_nsl < 3 * G + 4) G /= 2;
    if (!e)
        while (G < 64 && min_nsl >= 2 * (2 * G) + 4 && ((live * G) / 64) * (size_t)std::max(1, in_flight) < 512) G *= 2;
    return G < 2 ? 1 : G;
}

// k_banded_coop over the list, then k_banded<false> over the tasks it flagged
static ScoreLaunch launch_banded_coop(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int G, int timed) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    const int NA = 64 / G;
    const size_t nwaves = S.nt / NA;
    std::vector<int64_t> w_off(nwaves);
    std::vector<int32_t> w_ns(nwaves), w_nr(nwaves), w_nch(nwaves);
    size_t ws_bytes = 0;
    for (size_t w = 0; w < nwaves; ++w) {
        int ns = 3, nr = 4, nch = 2;
        for (int q = 0; q < NA; ++q) {
            const size_t t = w * NA + q;
            if (L.pair[t] < 0) continue;
            const HGeom Gm = host_geometry(L.m[t], L.n[t], L.cutoff[t]);
            ns = std::max(ns, Gm.ebb_local);
            nr = std::max(nr, (L.m[t] + 63) / 64 + Gm.ebb_local + 4);
            nch = std::max(nch, L.n[t] / 64 + 3);
        }
        w_ns[w] = ns; w_nr[w] = nr; w_nch[w] = nch;
        w_off[w] = (int64_t)ws_bytes;
        const size_t bytes = (size_t)2 * (ns + 1) * NA * 8 + (size_t)2 * nr * NA * 4 + (size_t)2 * nch * NA * 2 + (size_t)2 * NA * 4;
        ws_bytes += (bytes + 255) & ~(size_t)255;
    }
    S.T = upload_tasks(L, C);
    S.O = take_out(C, S.nt);
    uint8_t* ws = C.scratch_p->take<uint8_t>(ws_bytes + 256);
    int64_t* d_off = C.scratch_p->take<int64_t>(nwaves); int32_t* d_ns = C.scratch_p->take<int32_t>(nwaves);
    int32_t* d_nr = C.scratch_p->take<int32_t>(nwaves); int32_t* d_nch = C.scratch_p->take<int32_t>(nwaves);
    h2d(d_off, w_off, C.stream); h2d(d_ns, w_ns, C.stream); h2d(d_nr, w_nr, C.stream); h2d(d_nch, w_nch, C.stream);
    S.G = G; S.cws = ws; S.c_off = d_off; S.c_ns = d_ns; S.c_nr = d_nr; S.c_nch = d_nch;
    CoopArgs a;
    a.P = pair_view(B, reversed); a.T = S.T.v; a.G = G;
    a.ws = ws; a.w_ws_off = d_off; a.w_nslots = d_ns; a.w_nrows = d_nr; a.w_nch = d_nch;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len; a.o_abort = S.O.hew;
    HIP_CHECK(hipMemsetAsync(S.O.hew, 0, S.nt * sizeof(int32_t), C.stream));
    auto* ke = timed ? C.kernel_events(timed - 1) : nullptr;       // timed = kind + 1 (Context::kernel_events)
    if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
    // band state on chip where a wave's tasks fit its share of the LDS (k_banded_coop_lds); QE_COOP_LDS = 0: never
    CoopLdsArgs x;
    memset(&x, 0, sizeof(x));
    x.A = a;
    x.lgG = 0; while ((1 << x.lgG) < G) ++x.lgG;
    x.ns = 3; for (int32_t v : w_ns) x.ns = std::max(x.ns, v);
    x.rr = x.ns + G + 4;
    x.cr = std::max(16, 4 * G);
    {
        const size_t bytes = (size_t)2 * (x.ns + 1) * NA * 8 + (size_t)2 * x.rr * NA * 4 + (size_t)2 * x.cr * NA * 2 + (size_t)2 * NA * 4;
        x.lds_per_wave = (int32_t)((bytes + 63) & ~(size_t)63);
    }
    const int lds_env = env_int("QE_COOP_LDS", 1);
    if (lds_env != 0 && (size_t)x.lds_per_wave <= (size_t)38 * 1024)
        launch_groups(C, k_banded_coop_lds<false>, x, (size_t)nwaves, 8, (size_t)x.lds_per_wave);
    else
        launch_groups(C, k_banded_coop, a, (size_t)nwaves, 8, 0);
    // fallback pass: one lane per task, only where a band-edge decision could not be resolved in time
    const BandLayout lay = band_layout(L, false, false);
    S.D = upload_layout(lay, C);
    BandedArgs b;
    b.P = a.P; b.T = S.T.v;
    b.ws = S.D.ws; b.g_ws_off = S.D.ws_off; b.g_nslots = S.D.nslots; b.g_nrows = S.D.nrows; b.g_nch = S.D.nch;
    b.mat = nullptr; b.g_mat_off = S.D.mat_off;
    b.o_score = S.O.score; b.o_first = S.O.first; b.o_last = S.O.last; b.o_posv = S.O.posv; b.o_adv = S.O.adv;
    b.o_maxrow = S.O.len; b.only_if = S.O.hew;
    b.lane_rel = env_int("QE_LANE_REL", 1);
    launch_groups(C, k_banded<false>, b, L.ngroups(), 8, 0);
    if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
    return S;
}

// upper bound of one pair's RLE string incl. terminator: every op its own run
static size_t cigar_bound(int m, int n) { return (size_t)2 * ((size_t)m + (size_t)n) + 12; }
// the same for an alignment made of `leaves` BandEd leaves whose run buffers hold at most `runs` runs in total: a run is
// "<= 10 digits + op"; never more than the every-op-its-own-run bound
static size_t cigar_bound_runs(int m, int n, int64_t runs, int leaves) {
    return std::min(cigar_bound(m, n), (size_t)11 * (size_t)(runs + 2 * leaves + 2) + 12);
}

// ---------------------------------------------------------------------------
// CIGAR assembly: per list entry ("root" = one pair's alignment) an ordered list of segments
// ---------------------------------------------------------------------------
struct SegList {
    std::vector<int64_t> off;                 // [nroots + 1]
    std::vector<int32_t> kind, a, b;
    std::vector<int32_t> root_pair;           // pair index of every root
    std::vector<size_t> bound;                // string bound of every root
};

struct AlignOut {                             // device, per root
    int32_t *len = nullptr, *edits = nullptr, *nops = nullptr;
    int64_t *str_off = nullptr, *total = nullptr;
    char* pool = nullptr;
    int32_t* ok = nullptr;                    // validator verdicts (null unless the batch asks for them)
    size_t nroots = 0;
    size_t pool_bytes = 0;                    // what `pool` was sized for (the host-side bound of the strings)
};

// Results of a sync == 0 run, still on the device: what quicked_batch_fetch() copies once the run is over.  The device
// pointers are those of the batch's result arena (stash_results): valid until the batch's next run, reload or destroy.
struct PendingFetch {
    int kind = 0;                             // 1: one score per task (score-only BandEd / WindowEd); 2: alignments (segments)
    quicked_status_t ok_status = QUICKED_WIP;
    bool want_strings = false;
    // kind 1
    std::vector<int32_t> task_pair;
    const int32_t* d_score = nullptr; const u32* d_adv = nullptr; const u32* d_steps = nullptr; const int32_t* d_abort = nullptr;
    int counter_slot = 0;                     // where sum(adv) / sum(steps) goes in counters[]
    // kind 2
    SegList SL; AlignOut AO; std::vector<int32_t> root_status;
    std::vector<int32_t> leaf_pair; const u32* d_leaf_adv = nullptr; const u32* d_leaf_steps = nullptr;
    int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // what the run's host-side stages already counted
    bool quicked = false;                     // run_quicked ignores the Hirschberg status (quicked.c:290-291)
    // QuickEd fast path (quicked_fast): what decides which pairs still need the classic flow
    bool fast = false;
    const int32_t* d_cut = nullptr; const int32_t* d_skip = nullptr; const u32* d_stage_steps = nullptr;
    quicked_params_t params; TaskList L; size_t matrix_budget = 0;
    int parity = 0;                           // the plane set / ev_done slot of the run
};

// One wavefront per alignment (k_banded_wave) is for few, long alignments: up to ~1000 tasks every task gets a wave of its
// own at once and the run takes one alignment's latency (measured, 10 kb reads: 4.2 ms against 5.1 ms for the
// cooperative form; beyond ~2000 tasks, or for 1 kb reads, the other forms win: tools/small_n_probe.py).  Needs whole
// passes (tfin == n: no stopped band to export) and a band that fits the wave.  QE_WAVE = 0 / 1 switches the form off /
// forces it wherever it is eligible (tests).
static bool wave_form_wanted(const TaskList& L) {
    const int force = env_int("QE_WAVE", -1);
    if (force == 0) return false;
    size_t live = 0;
    int n_max = 0;
    for (size_t t = 0; t < L.pair.size(); ++t) {
        if (L.pair[t] < 0) continue;
        ++live;
        n_max = std::max(n_max, L.n[t]);
        if (L.tfin[t] != L.n[t] || host_geometry(L.m[t], L.n[t], L.cutoff[t]).ebb_local > 62) return false;
    }
    return live > 0 && (force == 1 || (live <= 1024 && n_max >= 4096));
}

static ScoreLaunch launch_banded_wave(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int timed) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    S.T = upload_tasks(L, C);
    S.O = take_out(C, S.nt);
    BandedArgs a;
    memset(&a, 0, sizeof(a));
    a.P = pair_view(B, reversed); a.T = S.T.v;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len;
    auto* ke = timed ? C.kernel_events(timed - 1) : nullptr;       // timed = kind + 1 (Context::kernel_events)
    if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
    launch_groups(C, k_banded_wave, a, S.nt, 4, 0);                     // one wave per task
    if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
    return S;
}

static void run_banded_score(quicked_batch& B, Context& C, const TaskList& L, bool reversed, StageResult* R,
                             bool fetch, int32_t** d_score_out, PendingFetch* pf = nullptr) {
    // one wavefront per alignment only where the cooperative on-chip form has no room (a band of fewer than 8 slots): with
    // G = 8 lanes per alignment and 4-slot passes that form does a 10 kb pair in 2.7 ms, the wave form in 4.3
    const int G0 = coop_lanes(L, fetch ? 1 : C.in_flight);
    const bool wave = wave_form_wanted(L) && (G0 < 2 || env_int("QE_WAVE", -1) == 1);
    const int G = wave ? 1 : G0;
    const ScoreLaunch S = wave ? launch_banded_wave(B, C, L, reversed, 1)
                               : ((G >= 2) ? launch_banded_coop(B, C, L, reversed, G, 1) : launch_banded_score(B, C, L, reversed, 1));
    if (d_score_out) *d_score_out = S.O.score;
    if (pf && !fetch) {
        pf->kind = 1; pf->task_pair = L.pair; pf->d_score = S.O.score; pf->d_adv = S.O.adv; pf->counter_slot = 0;
        pf->d_abort = (G >= 2) ? S.O.hew : nullptr;
    }
    if (fetch && R) {
        if (G >= 2) d2h(R->hew, S.O.hew, S.nt, C.stream);       // abort flags (diagnostics)
        d2h(R->score, S.O.score, S.nt, C.stream); d2h(R->adv, S.O.adv, S.nt, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
    }
}

// few alignments with many runs each (long reads): one wave per alignment; else one lane per alignment.  Decided before the
// traceback runs: the wave form wants every task's runs in a stretch of their own (TraceArgs::runs_by_task)
static bool wave_formatter_wanted(const quicked_batch& B, const SegList& SL, bool want_strings) {
    const size_t nr = SL.root_pair.size(), nseg = SL.kind.size();
    size_t pool_bytes = 0;
    if (want_strings) for (size_t b : SL.bound) pool_bytes += b;
    const int wave_env = env_int("QE_FORMAT_WAVE", -1);      // tests force either form
    return B.cigar_style != 2 && nr > 0 && (wave_env >= 0 ? wave_env != 0 : (nr <= 32768 && nseg > 0 && pool_bytes / nr >= 16384));
}

static AlignOut format_segments(const quicked_batch& B, Context& C, const SegList& SL, const u32* runs, const int64_t* g_runs_off,
                                const int32_t* nruns, bool want_strings, bool wave = false, const int32_t* g_runs_cap = nullptr, bool runs_by_task = false) {
    AlignOut A;
    A.nroots = SL.root_pair.size();
    const size_t nr = A.nroots, nseg = SL.kind.size();
    int64_t* d_off = C.scratch_p->take<int64_t>(nr + 1);
    int32_t* d_kind = C.scratch_p->take<int32_t>(nseg + 1); int32_t* d_a = C.scratch_p->take<int32_t>(nseg + 1);
    int32_t* d_b = C.scratch_p->take<int32_t>(nseg + 1);
    int32_t* d_rootpair = C.scratch_p->take<int32_t>(nr + 1);
    h2d(d_off, SL.off, C.stream); h2d(d_kind, SL.kind, C.stream); h2d(d_a, SL.a, C.stream); h2d(d_b, SL.b, C.stream);
    h2d(d_rootpair, SL.root_pair, C.stream);
    A.len = C.scratch_p->take<int32_t>(nr + 1); A.edits = C.scratch_p->take<int32_t>(nr + 1); A.nops = C.scratch_p->take<int32_t>(nr + 1);
    A.str_off = C.scratch_p->take<int64_t>(nr + 1); A.total = C.scratch_p->take<int64_t>(1);
    size_t pool_bytes = 0;
    if (want_strings) for (size_t b : SL.bound) pool_bytes += b;
    A.pool = C.scratch_p->take<char>(pool_bytes + 16);
    A.pool_bytes = pool_bytes + 16;
    SegFormatArgs f;
    f.npairs = (int32_t)nr; f.seg_off = d_off; f.seg_kind = d_kind; f.seg_a = d_a; f.seg_b = d_b;
    f.runs = runs; f.g_runs_off = g_runs_off; f.nruns = nruns;
    f.g_runs_cap = g_runs_cap; f.runs_by_task = runs_by_task ? 1 : 0;
    f.o_len = A.len; f.o_edits = A.edits; f.o_nops = A.nops; f.str_off = A.str_off; f.pool = A.pool;
    f.style = B.cigar_style;
    const int blocks = (int)((nr + 63) / 64);
    if (B.check && want_strings) {
        A.ok = C.scratch_p->take<int32_t>(nr + 1);
        SegCheckArgs ck;
        ck.F = f; ck.P = pair_view(B, false); ck.root_pair = d_rootpair; ck.o_ok = A.ok;
        hipLaunchKernelGGL(k_check_segs, dim3(blocks), dim3(64), 0, C.stream, ck);
    }
    (void)nseg;
    if (wave) hipLaunchKernelGGL(k_format_segs_wave<false>, dim3((unsigned)nr), dim3(64), 0, C.stream, f);
    else hipLaunchKernelGGL(k_format_segs<false>, dim3(blocks), dim3(64), 0, C.stream, f);
    if (want_strings) {
        hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(1024), 0, C.stream, A.len, d_rootpair, A.str_off, A.total, (int)nr);
        if (wave) hipLaunchKernelGGL(k_format_segs_wave<true>, dim3((unsigned)nr), dim3(64), 0, C.stream, f);
        else hipLaunchKernelGGL(k_format_segs<true>, dim3(blocks), dim3(64), 0, C.stream, f);
    }
    return A;
}

// D2H of a formatted stage into the batch's host-side result arrays
static void fetch_alignments(quicked_batch& B, Context& C, const SegList& SL, const AlignOut& A, bool want_strings,
                             int32_t ok_status, const std::vector<int32_t>* root_status) {
    std::vector<int32_t> len, edits, nops; std::vector<int64_t> off;
    d2h(len, A.len, A.nroots, C.stream); d2h(edits, A.edits, A.nroots, C.stream); d2h(nops, A.nops, A.nroots, C.stream);
    if (want_strings) d2h(off, A.str_off, A.nroots, C.stream);
    std::vector<int32_t> okv;
    if (A.ok) d2h(okv, A.ok, A.nroots, C.stream);
    HIP_CHECK(hipStreamSynchronize(C.stream));
    int64_t total = 0;
    if (want_strings) for (size_t i = 0; i < A.nroots; ++i) total = std::max<int64_t>(total, off[i] + len[i] + 1);
    const size_t base = B.wr->cigar_pool.size;
    if (total) {
        B.wr->cigar_pool.reserve(base + (size_t)total);
        HIP_CHECK(hipMemcpyAsync(B.wr->cigar_pool.p + base, A.pool, (size_t)total, hipMemcpyDeviceToHost, C.stream));
        HIP_CHECK(hipStreamSynchronize(C.stream));
        B.wr->cigar_pool.size = base + (size_t)total;
    }
    for (size_t i = 0; i < A.nroots; ++i) {
        const int pr = SL.root_pair[i];
        B.wr->score[pr] = edits[i];
        B.wr->status[pr] = root_status ? (*root_status)[i] : ok_status;
        if (edits[i] < 0) { B.wr->score[pr] = -1; B.wr->status[pr] = QUICKED_ERROR; }      // run-buffer overflow: cutoff below the distance
        B.counters[4] += nops[i];
        B.note_pair(pr, 4, nops[i]);
        if (A.ok) B.wr->check_ok[pr] = okv[i];
        if (want_strings && len[i] > 0) B.wr->cigar_off[pr] = (int64_t)base + off[i];      // NUL-terminated in the pool
    }
}

// WindowEd over a task list (bpm_windowed.c:563-628)
static void run_windowed(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int W, int O_, int hew_threshold,
                         bool score_only, bool sse, StageResult* R, bool fetch, bool want_cigar, int32_t** d_score_out,
                         PendingFetch* pf = nullptr, TaskOut* dev_out = nullptr, DevTasks* dev_tasks = nullptr) {
    const size_t nt = L.pair.size();
    const int ng = L.ngroups();
    // per group: Pv/Mv [W][64] u64 + tiled history of (64W+3) columns x W blocks
    const size_t g_bytes = ((size_t)2 * W * 64 * 8 + (size_t)(8 * W + 2) * W * 512 * 16 + 255) & ~(size_t)255;
    BandLayout lay;
    lay.ws_off.resize(ng); lay.mat_off.assign(ng, 0); lay.runs_off.resize(ng);
    lay.nslots.assign(ng, W); lay.nrows.assign(ng, 0); lay.nch.assign(ng, 0); lay.runs_cap.resize(ng);
    for (int g = 0; g < ng; ++g) {
        int cap = 2;
        for (int l = 0; l < 64; ++l) {
            const size_t t = (size_t)g * 64 + l;
            if (L.pair[t] >= 0) cap = std::max(cap, L.m[t] + L.n[t] + 2);
        }
        lay.ws_off[g] = (int64_t)lay.ws_bytes; lay.ws_bytes += g_bytes;
        lay.runs_cap[g] = cap; lay.runs_off[g] = (int64_t)lay.runs_u32;
        if (!score_only) lay.runs_u32 += (size_t)cap * 64;
    }
    const DevTasks T = upload_tasks(L, C);
    const DevLayout D = upload_layout(lay, C);
    const TaskOut O = take_out(C, nt);
    WindowArgs a;
    a.P = pair_view(B, reversed); a.T = T.v;
    a.W = W; a.O = O_; a.hew_threshold = hew_threshold; a.score_only = score_only ? 1 : 0; a.sse = sse ? 1 : 0; a.reversed = reversed ? 1 : 0;
    a.ws = D.ws; a.g_ws_off = D.ws_off; a.runs = D.runs; a.g_runs_off = D.runs_off; a.g_runs_cap = D.runs_cap;
    a.o_score = O.score; a.o_hew = O.hew; a.o_nruns = O.nruns; a.o_nops = O.nops; a.o_edits = O.edits; a.o_steps = O.steps;
    // (2, 1) windows stay on chip (k_windowed); every other shape runs the checkpointed general path
    a.cp_path = env_int("QE_WINDOWED_CP", 1);
    if (W == 2 && O_ == 1) launch_groups(C, k_windowed, a, (size_t)ng, 8, 8192, /* chain */ true);
    else launch_groups(C, k_windowed_cp, a, (size_t)ng, 8, 8192, /* chain */ true);
    if (d_score_out) *d_score_out = O.score;
    if (dev_out) *dev_out = O;
    if (dev_tasks) *dev_tasks = T;
    SegList SL; AlignOut AO;
    if (!score_only) {
        SL.off.push_back(0);
        for (size_t t = 0; t < nt; ++t) {
            if (L.pair[t] < 0) continue;
            SL.kind.push_back(0); SL.a.push_back((int32_t)t); SL.b.push_back(0);
            SL.off.push_back((int64_t)SL.kind.size());
            SL.root_pair.push_back(L.pair[t]); SL.bound.push_back(cigar_bound(L.m[t], L.n[t]));
        }
        AO = format_segments(B, C, SL, D.runs, D.runs_off, O.nruns, want_cigar, wave_formatter_wanted(B, SL, want_cigar), D.runs_cap, false);
        if (d_score_out) *d_score_out = AO.edits;
    }
    if (pf && !fetch) {
        pf->task_pair = L.pair; pf->d_score = O.score; pf->d_steps = O.steps; pf->counter_slot = 2;
        pf->kind = score_only ? 1 : 2;
        if (!score_only) { pf->SL = std::move(SL); pf->AO = AO; pf->want_strings = want_cigar; pf->ok_status = QUICKED_WIP; }
        return;
    }
    if (fetch && R) {
        d2h(R->score, O.score, nt, C.stream); d2h(R->hew, O.hew, nt, C.stream); d2h(R->steps, O.steps, nt, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        if (!score_only) fetch_alignments(B, C, SL, AO, want_cigar, QUICKED_WIP, nullptr);
    }
}

// ---------------------------------------------------------------------------
// The align step (bpm_compute_matrix_hirschberg, bpm_hirschberg.c:33-270) over a list of
// roots = (pair, cutoff).  The recursion becomes a level-by-level work list: every level is
// one batch of forward + reverse score-only half passes and one join kernel; the leaves of
// all levels are then filled and traced back in sub-batches that fit the pool, and every
// pair's leaves are stitched into one CIGAR in text order.
// ---------------------------------------------------------------------------
static double now_ms() {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// BUFFER_SIZE_16M of bpm_hirschberg.c:65; QE_SPLIT_BYTES lowers it so tests can force many split levels on small inputs
static uint64_t split_threshold() {
    const char* e = getenv("QE_SPLIT_BYTES");
    return e ? (uint64_t)strtoull(e, nullptr, 10) : ((uint64_t)1 << 24);
}
static bool trace_on() { static int v = -1; if (v < 0) v = getenv("QE_TRACE") ? 1 : 0; return v == 1; }
#define QE_TRACE_POINT(name) do { if (trace_on()) { double t__ = now_ms(); fprintf(stderr, "[qe t%03d @%.1f] %-22s +%.3f ms\n", (int)(syscall(SYS_gettid) % 1000), t__, name, t__ - tr_last); tr_last = t__; } } while (0)

static void reset_host_results(quicked_batch& B) {
    B.wr->score.assign((size_t)B.n, -1);
    B.wr->status.assign((size_t)B.n, QUICKED_EMPTY_SEQUENCE);
    B.wr->cigar_off.assign((size_t)B.n, -1);
    B.wr->cigar_pool.size = 0;
    B.wr->check_ok.assign((size_t)B.n, -1);
    B.wr->deferred_pairs = 0;
}

struct HNode { int32_t pair, p0, m, t0, n, cutoff, left, right, leaf_task; };

struct AlignStats { uint64_t fill_adv = 0, tb_steps = 0, score_adv = 0, splits = 0, leaves = 0; };

static void run_align(quicked_batch& B, Context& C, const TaskList& roots, bool fetch, bool want_cigar,
                      size_t matrix_budget, uint64_t split_bytes, int32_t ok_status, int32_t** d_score_out, AlignStats* stats,
                      PendingFetch* pf = nullptr, bool tight_runs = false, const int32_t* d_cut = nullptr, const int32_t* d_skip = nullptr) {
    double tr_last = now_ms();
    std::vector<HNode> nodes;
    std::vector<int32_t> root_node, root_status;
    for (size_t t = 0; t < roots.pair.size(); ++t) {
        if (roots.pair[t] < 0) continue;
        root_node.push_back((int32_t)nodes.size());
        root_status.push_back(ok_status);
        nodes.push_back(HNode{roots.pair[t], roots.p0[t], roots.m[t], roots.t0[t], roots.n[t], roots.cutoff[t], -1, -1, -1});
    }
    std::vector<int32_t> node_root(nodes.size());
    for (size_t i = 0; i < root_node.size(); ++i) node_root[root_node[i]] = (int32_t)i;
    // ---- split levels
    std::vector<int32_t> frontier(root_node);
    while (true) {
        std::vector<int32_t> split;
        for (int32_t id : frontier) {
            const HNode& nd = nodes[id];
            if (nd.m == 0 || nd.n == 0) continue;
            const HGeom G = host_geometry(nd.m, nd.n, nd.cutoff);
            if ((uint64_t)G.ebb * (uint64_t)nd.n * 16u > split_bytes) split.push_back(id);     // bpm_hirschberg.c:63-65
        }
        if (split.empty()) break;
        const DevicePool::Mark mark = C.scratch_p->mark();
        TaskList F, V;
        std::vector<int32_t> hm, hn1, hn2;
        for (int32_t id : split) {
            const HNode& nd = nodes[id];
            const int n1 = (nd.n + 1) / 2, n2 = nd.n - n1;                                      // bpm_hirschberg.c:68-69
            // both half passes use the FULL (m, n, cutoff) geometry and stop at their half (85-100)
            F.push(nd.pair, nd.p0, nd.m, nd.t0, nd.n, nd.cutoff, n1);
            V.push(nd.pair, B.p_len[nd.pair] - (nd.p0 + nd.m), nd.m, B.t_len[nd.pair] - (nd.t0 + nd.n), nd.n, nd.cutoff, n2);
            hm.push_back(nd.m); hn1.push_back(n1); hn2.push_back(n2);
        }
        F.pad(); V.pad();
        if (!B.have_rev[B.parity]) {
            hipStream_t cur = C.stream; C.stream = C.sw();
            launch_pack(B, C, true);
            HIP_CHECK(hipStreamSynchronize(C.sw()));
            C.stream = cur; B.have_rev[B.parity] = true;
        }
        const int Gf = coop_lanes(F);
        // forward half passes on the run's stream, reverse ones beside them on the side stream, joined before k_join
        hipStream_t main_s = C.stream, side = C.side_stream();
        if (side != main_s) { HIP_CHECK(hipEventRecord(C.ev_fork, main_s)); HIP_CHECK(hipStreamWaitEvent(side, C.ev_fork, 0)); }
        const ScoreLaunch SF = (Gf >= 2) ? launch_banded_coop(B, C, F, false, Gf, 3) : launch_banded_score(B, C, F, false, 3);
        C.stream = side;
        const ScoreLaunch SV = (Gf >= 2) ? launch_banded_coop(B, C, V, true, Gf, 3) : launch_banded_score(B, C, V, true, 3);
        C.stream = main_s;
        if (side != main_s) { HIP_CHECK(hipEventRecord(C.ev_join, side)); HIP_CHECK(hipStreamWaitEvent(main_s, C.ev_join, 0)); }
        const size_t ns = split.size();
        JoinArgs J;
        J.nnodes = (int32_t)ns;
        int32_t* dm = C.scratch_p->take<int32_t>(ns); int32_t* dn1 = C.scratch_p->take<int32_t>(ns); int32_t* dn2 = C.scratch_p->take<int32_t>(ns);
        h2d(dm, hm, C.stream); h2d(dn1, hn1, C.stream); h2d(dn2, hn2, C.stream);
        J.m = dm; J.n1 = dn1; J.n2 = dn2;
        J.Ffb = band_state(SF); J.Rfb = band_state(SV);
        J.F = (Gf >= 2) ? coop_state(SF) : J.Ffb; J.R = (Gf >= 2) ? coop_state(SV) : J.Rfb;
        J.o_best = C.scratch_p->take<int32_t>(ns); J.o_score_l = C.scratch_p->take<int32_t>(ns);
        J.o_score_r = C.scratch_p->take<int32_t>(ns); J.o_ok = C.scratch_p->take<int32_t>(ns);
        hipLaunchKernelGGL(k_join, dim3((unsigned)ns), dim3(64), 0, C.stream, J);       // one wave per node
        std::vector<int32_t> best, sl, sr, ok; std::vector<u32> advf, advv;
        d2h(best, J.o_best, ns, C.stream); d2h(sl, J.o_score_l, ns, C.stream); d2h(sr, J.o_score_r, ns, C.stream);
        d2h(ok, J.o_ok, ns, C.stream); d2h(advf, SF.O.adv, ns, C.stream); d2h(advv, SV.O.adv, ns, C.stream);
        QE_TRACE_POINT("  level: queued");
        HIP_CHECK(hipStreamSynchronize(C.stream));
        QE_TRACE_POINT("  level: half passes+join");
        C.scratch_p->release(mark);
        if (stats) { stats->score_adv += sum_u32(advf) + sum_u32(advv); stats->splits += ns; }
        for (size_t k = 0; k < ns; ++k) B.note_pair(nodes[split[k]].pair, 0, (int64_t)advf[k] + (int64_t)advv[k]);
        frontier.clear();
        for (size_t k = 0; k < ns; ++k) {
            const int32_t id = split[k];
            const HNode nd = nodes[id];
            if (!ok[k]) {                                                                       // bpm_hirschberg.c:116-122
                root_status[node_root[id]] = QUICKED_FAIL_NON_CONVERGENCE;
                nodes[id].m = 0; nodes[id].n = 0;                                               // contributes nothing
                continue;
            }
            const int n1 = (nd.n + 1) / 2;
            const int32_t l = (int32_t)nodes.size(), r = l + 1;
            nodes.push_back(HNode{nd.pair, nd.p0, best[k], nd.t0, n1, sl[k], -1, -1, -1});
            nodes.push_back(HNode{nd.pair, nd.p0 + best[k], nd.m - best[k], nd.t0 + n1, nd.n - n1, sr[k], -1, -1, -1});
            node_root.push_back(node_root[id]); node_root.push_back(node_root[id]);
            nodes[id].left = l; nodes[id].right = r;
            frontier.push_back(l); frontier.push_back(r);
        }
    }
    // ---- leaves in text order, per root
    TaskList LL;
    SegList SL;
    SL.off.push_back(0);
    std::vector<int32_t> stack;
    for (size_t i = 0; i < root_node.size(); ++i) {
        stack.clear();
        stack.push_back(root_node[i]);
        int64_t root_runs = 0; int root_segs = 0;
        while (!stack.empty()) {
            const int32_t id = stack.back(); stack.pop_back();
            HNode& nd = nodes[id];
            if (nd.left >= 0) { stack.push_back(nd.right); stack.push_back(nd.left); continue; }
            if (nd.m == 0 && nd.n == 0) continue;
            ++root_segs;
            if (nd.m == 0) { SL.kind.push_back(1); SL.a.push_back((int32_t)OP_I); SL.b.push_back(nd.n); continue; }
            if (nd.n == 0) { SL.kind.push_back(1); SL.a.push_back((int32_t)OP_D); SL.b.push_back(nd.m); continue; }
            nd.leaf_task = (int32_t)LL.pair.size();
            LL.push(nd.pair, nd.p0, nd.m, nd.t0, nd.n, nd.cutoff, nd.n);
            SL.kind.push_back(0); SL.a.push_back(nd.leaf_task); SL.b.push_back(0);
            root_runs += tight_runs ? std::min<int64_t>((int64_t)nd.m + nd.n + 2, (int64_t)2 * host_geometry(nd.m, nd.n, nd.cutoff).cutoff + 8)
                                    : (int64_t)nd.m + nd.n + 2;                                                        // band_layout's cap
        }
        SL.off.push_back((int64_t)SL.kind.size());
        const HNode& rt = nodes[root_node[i]];
        SL.root_pair.push_back(rt.pair);
        SL.bound.push_back(cigar_bound_runs(rt.m, rt.n, root_runs, root_segs));
    }
    const size_t n_leaves = LL.pair.size();
    LL.pad();
    QE_TRACE_POINT("  leaves listed");
    if (stats) stats->leaves += LL.pair.size();
    // ---- leaves: fill + traceback in sub-batches; runs and per-leaf outputs persist
    const size_t nt = LL.pair.size();
    const int ng = LL.ngroups();
    const BandLayout lay = band_layout(LL, true, true, tight_runs);
    B.last_mat_bytes = lay.mat_u4 * 16;
    // what the stage takes from the pool besides the matrices: run buffers, string pool, per-task arrays, segment lists
    size_t fixed_bytes = lay.runs_u32 * 4 + (size_t)nt * 160 + SL.kind.size() * 16 + ((size_t)4 << 20);
    if (want_cigar) for (size_t b : SL.bound) fixed_bytes += b;
    B.last_fixed_bytes = fixed_bytes + lay.ws_bytes;
    B.last_groups = ng;
    // partition the groups so that each sub-batch's matrices (and workspaces) fit what is left of the pool's budget
    // (matrix_budget = the pool's whole budget, plan_pools in run_batch); sub-batches are made equal so that none is a
    // sliver; offsets restart per sub-batch
    std::vector<int> sub_start{0};
    std::vector<int64_t> ws_off(ng), mat_off(ng);
    {
        const size_t room = matrix_budget > fixed_bytes + ((size_t)64 << 20) ? matrix_budget - fixed_bytes : (size_t)64 << 20;
        const size_t total = lay.mat_u4 * 16 + lay.ws_bytes;
        const size_t nsub = std::max<size_t>(1, (total + room - 1) / room);
        const size_t target = (total + nsub - 1) / nsub;                  // bytes per sub-batch when split evenly
        size_t ws = 0, mat = 0;
        for (int g = 0; g < ng; ++g) {
            const size_t gws = (size_t)((g + 1 < ng ? lay.ws_off[g + 1] : (int64_t)lay.ws_bytes) - lay.ws_off[g]);
            const size_t gmat = (size_t)((g + 1 < ng ? lay.mat_off[g + 1] : (int64_t)lay.mat_u4) - lay.mat_off[g]);
            const size_t after = (mat + gmat) * 16 + ws + gws;
            if (g > sub_start.back() && (after > room || (nsub > 1 && after > target + target / 16))) { sub_start.push_back(g); ws = 0; mat = 0; }
            ws_off[g] = (int64_t)ws; mat_off[g] = (int64_t)mat;
            ws += gws; mat += gmat;
        }
        sub_start.push_back(ng);
    }
    C.last_sub_batches = (int)sub_start.size() - 1;
    {   // what is still to be taken from the pool: per-task arrays, run buffers, strings, one sub-batch of matrices
        size_t sub_max = 0;
        for (size_t sb = 0; sb + 1 < sub_start.size(); ++sb) {
            size_t b = 0;
            for (int g = sub_start[sb]; g < sub_start[sb + 1]; ++g)
                b += (size_t)((g + 1 < ng ? lay.ws_off[g + 1] : (int64_t)lay.ws_bytes) - lay.ws_off[g]) +
                     16 * (size_t)((g + 1 < ng ? lay.mat_off[g + 1] : (int64_t)lay.mat_u4) - lay.mat_off[g]);
            sub_max = std::max(sub_max, b);
        }
        if (fixed_bytes + sub_max > ((size_t)1 << 30)) C.scratch_p->reserve(fixed_bytes + sub_max);
    }
    const DevTasks T = upload_tasks(LL, C);
    const TaskOut O = take_out(C, nt);
    if (d_cut) {
        // the roots' cutoffs are still being computed on the device (quicked_fast): the host sized everything for the
        // estimates in roots.cutoff; no root may have split or vanished, so leaf k is root k
        if (n_leaves != root_node.size() || nodes.size() != root_node.size())
            throw HipError{hipErrorInvalidValue, "device-side cutoffs need one leaf per root", __LINE__};
        HIP_CHECK(hipMemsetAsync(O.nruns, 0xFF, nt * sizeof(int32_t), C.stream));      // a task taken out of the list has no runs (-1)
        hipLaunchKernelGGL(k_apply_cutoffs, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, C.stream, (int)nt, T.cutoff, T.pair, d_cut, d_skip);
    }
    int64_t* d_ws_off = C.scratch_p->take<int64_t>(ng + 1); int64_t* d_mat_off = C.scratch_p->take<int64_t>(ng + 1);
    int64_t* d_runs_off = C.scratch_p->take<int64_t>(ng + 1);
    int32_t* d_nslots = C.scratch_p->take<int32_t>(ng + 1); int32_t* d_nrows = C.scratch_p->take<int32_t>(ng + 1);
    int32_t* d_nch = C.scratch_p->take<int32_t>(ng + 1); int32_t* d_runs_cap = C.scratch_p->take<int32_t>(ng + 1);
    h2d(d_ws_off, ws_off, C.stream); h2d(d_mat_off, mat_off, C.stream); h2d(d_runs_off, lay.runs_off, C.stream);
    h2d(d_nslots, lay.nslots, C.stream); h2d(d_nrows, lay.nrows, C.stream); h2d(d_nch, lay.nch, C.stream);
    h2d(d_runs_cap, lay.runs_cap, C.stream);
    u32* d_runs = C.scratch_p->take<u32>(lay.runs_u32 + 64);
    const bool wave_fmt = wave_formatter_wanted(B, SL, want_cigar);       // also the layout the traceback leaves its runs in
    // Lanes per leaf for the fill: 1 where the leaves fill the chip -- and wherever the cutoff is a tight bound of the distance
    // (QuickEd's bound, Hirschberg's exact child distances): such a band is pruned down to a few slots, its height test
    // (first + 2 < last) cannot be decided chunks ahead, and the cooperative protocol hands most leaves back to the
    // one-lane kernel (config 4's leaves: both kernels ran, 45 + 36 ms instead of 38).  A user bandwidth leaves the band
    // tall: few long BandEd alignments with CIGAR fill with G lanes each.  QE_COOP_FILL_G forces a width (tests).
    const bool fill_forced = getenv("QE_COOP_FILL_G") != nullptr;
    const int Gfill = (env_int("QE_COOP_LDS", 1) == 0 || (tight_runs && !fill_forced)) ? 1 : coop_lanes(LL, fetch ? 1 : C.in_flight, true);
    for (size_t sb = 0; sb + 1 < sub_start.size(); ++sb) {
        const int g0 = sub_start[sb], g1 = sub_start[sb + 1];
        if (g1 <= g0) continue;
        size_t ws_bytes = 0, mat_u4 = 0;
        for (int g = g0; g < g1; ++g) {
            ws_bytes = std::max(ws_bytes, (size_t)ws_off[g] + (size_t)((g + 1 < ng ? lay.ws_off[g + 1] : (int64_t)lay.ws_bytes) - lay.ws_off[g]));
            mat_u4 = std::max(mat_u4, (size_t)mat_off[g] + (size_t)((g + 1 < ng ? lay.mat_off[g + 1] : (int64_t)lay.mat_u4) - lay.mat_off[g]));
        }
        const DevicePool::Mark mark = C.scratch_p->mark();
        uint8_t* ws = C.scratch_p->take<uint8_t>(ws_bytes + 256);
        uint4* mat = C.scratch_p->take<uint4>(mat_u4 + 16);
        const size_t o = (size_t)g0 * 64;
        BandedArgs a;
        a.P = pair_view(B, false);
        a.T = T.v;
        a.T.ntasks = (int32_t)((size_t)(g1 - g0) * 64);
        a.T.pair = T.pair + o; a.T.p0 = T.p0 + o; a.T.m = T.m + o; a.T.t0 = T.t0 + o; a.T.n = T.n + o;
        a.T.cutoff = T.cutoff + o; a.T.tfin = T.tfin + o;
        a.ws = ws; a.g_ws_off = d_ws_off + g0; a.g_nslots = d_nslots + g0; a.g_nrows = d_nrows + g0; a.g_nch = d_nch + g0;
        a.mat = mat; a.g_mat_off = d_mat_off + g0;
        a.o_score = O.score + o; a.o_first = O.first + o; a.o_last = O.last + o; a.o_posv = O.posv + o; a.o_adv = O.adv + o;
        a.o_maxrow = O.len + o;
        a.only_if = nullptr;
        auto* ke = C.kernel_events(1);
        if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
        if (Gfill >= 2) {
            // few leaves: G lanes per leaf, band state on chip, the same checkpoints / carry words / band edges in the
            // traceback's layout (k_banded_coop_lds<true>); leaves it flags are refilled by the one-lane kernel
            const int NAf = 64 / Gfill;
            CoopLdsArgs x;
            memset(&x, 0, sizeof(x));
            x.A.P = a.P; x.A.T = a.T; x.A.G = Gfill;
            x.A.o_score = a.o_score; x.A.o_first = a.o_first; x.A.o_last = a.o_last; x.A.o_posv = a.o_posv; x.A.o_adv = a.o_adv;
            x.A.o_maxrow = a.o_maxrow; x.A.o_abort = O.hew + o;
            x.lgG = 0; while ((1 << x.lgG) < Gfill) ++x.lgG;
            x.ns = 3; for (int g = g0; g < g1; ++g) x.ns = std::max(x.ns, lay.nslots[g]);
            x.rr = x.ns + Gfill + 4;
            x.cr = std::max(16, 4 * Gfill);
            const size_t bytes = (size_t)2 * (x.ns + 1) * NAf * 8 + (size_t)2 * x.rr * NAf * 4 + (size_t)2 * x.cr * NAf * 2 + (size_t)2 * NAf * 4;
            x.lds_per_wave = (int32_t)((bytes + 63) & ~(size_t)63);
            x.mat = mat; x.g_mat_off = a.g_mat_off; x.gws = ws; x.g_ws_off = a.g_ws_off;
            x.g_nslots = a.g_nslots; x.g_nrows = a.g_nrows; x.g_nch = a.g_nch;
            if ((size_t)x.lds_per_wave <= (size_t)38 * 1024) {
                HIP_CHECK(hipMemsetAsync(O.hew + o, 0, (size_t)(g1 - g0) * 64 * sizeof(int32_t), C.stream));
                launch_groups(C, k_banded_coop_lds<true>, x, (size_t)(g1 - g0) * 64 / NAf, 8, (size_t)x.lds_per_wave);
                a.only_if = O.hew + o;
            }
        }
        a.fill_multi = env_int("QE_FILL_MULTI", 1);
        a.lane_rel = env_int("QE_LANE_REL", 1);
        launch_groups(C, k_banded<true>, a, (size_t)(g1 - g0), 8, 0);     // everything, or what the cooperative fill flagged
        if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
        TraceArgs tr;
        tr.P = a.P; tr.T = a.T;
        tr.ws = ws; tr.g_ws_off = a.g_ws_off; tr.g_nslots = a.g_nslots; tr.g_nrows = a.g_nrows; tr.g_nch = a.g_nch;
        tr.mat = mat; tr.g_mat_off = a.g_mat_off;
        tr.runs = d_runs; tr.g_runs_off = d_runs_off + g0; tr.g_runs_cap = d_runs_cap + g0;
        tr.o_nruns = O.nruns + o; tr.o_nops = O.nops + o; tr.o_edits = O.edits + o; tr.o_steps = O.steps + o;
        tr.runs_by_task = wave_fmt ? 1 : 0;
        launch_groups(C, k_traceback, tr, (size_t)(g1 - g0), 8, 0);
        // the next sub-batch reuses this scratch: its kernels are behind this sub-batch's in the stream, no host wait
        if (sb + 2 < sub_start.size()) C.scratch_p->release(mark);
    }
    QE_TRACE_POINT("  fill+traceback queued");
    const AlignOut AO = format_segments(B, C, SL, d_runs, d_runs_off, O.nruns, want_cigar, wave_fmt, d_runs_cap, wave_fmt);
    QE_TRACE_POINT("  format queued");
    if (d_score_out) *d_score_out = AO.edits;
    if (pf && !fetch) {
        pf->kind = 2; pf->SL = std::move(SL); pf->AO = AO; pf->want_strings = want_cigar;
        pf->ok_status = (quicked_status_t)ok_status; pf->root_status = std::move(root_status);
        pf->leaf_pair = LL.pair; pf->d_leaf_adv = O.adv; pf->d_leaf_steps = O.steps;
        return;
    }
    if (fetch) {
        if (stats) {
            std::vector<u32> adv, steps;
            d2h(adv, O.adv, nt, C.stream); d2h(steps, O.steps, nt, C.stream);
            HIP_CHECK(hipStreamSynchronize(C.stream));
            for (size_t t = 0; t < nt; ++t) if (LL.pair[t] >= 0) { stats->fill_adv += adv[t]; stats->tb_steps += steps[t]; B.note_pair(LL.pair[t], 1, adv[t]); B.note_pair(LL.pair[t], 3, steps[t]); }
        }
        fetch_alignments(B, C, SL, AO, want_cigar, ok_status, &root_status);
    }
}

static int max_cutoff(unsigned bandwidth, int m, int n) {
    return (int)(((unsigned)std::max(m, n) * bandwidth) / 100u);    // quicked.c:64,131,246 (unsigned arithmetic)
}

// whole-batch task list in sorted order
static TaskList all_pairs(const quicked_batch& B, const quicked_params_t& p) {
    TaskList L;
    L.pair.reserve((size_t)B.n + 64);
    for (int64_t i = 0; i < B.n; ++i) {
        const int pr = B.order[(size_t)i];
        const int m = B.p_len[pr], n = B.t_len[pr];
        if (m == 0 || n == 0) continue;                             // QUICKED_EMPTY_SEQUENCE (quicked.c:411-414)
        L.push(pr, 0, m, 0, n, max_cutoff(p.bandwidth, m, n), n);
    }
    L.pad();
    return L;
}

static void scatter_scores(quicked_batch& B, const TaskList& L, const std::vector<int32_t>& s, int32_t ok_status) {
    for (size_t t = 0; t < L.pair.size(); ++t) {
        const int pr = L.pair[t];
        if (pr < 0) continue;
        B.wr->score[pr] = s[t];
        B.wr->status[pr] = ok_status;
    }
}

// ---------------------------------------------------------------------------
// The batch entry point: dispatch on params->algo (quicked_align, quicked.c:405-437)
// ---------------------------------------------------------------------------
// The classic QUICKED / HIRSCHBERG flow over the tasks of L (run_quicked, quicked.c:163-306; run_hirschberg, 125-161):
// the bound stages are host-synchronous -- stage 1's results regroup the pairs for stages 2 and 3 -- then the align step
// runs with the bounds as cutoffs.  Called for a whole batch, or (after the fast path below) for the pairs it left.
// QuickEd sizes the align step's buffers for an ESTIMATE of the bounds that is the same whether the bounds are known on
// the host (classic flow) or still being computed on the device (fast flow): the pools then see one request sequence.
static int quicked_estimate(int top) { return top + top / 8 + 16; }
// ... of a run's bounds: the largest plus a margin -- unless a few pairs lie far above the rest (reads with large indels among
// ordinary ones: bound 3 700 against 480), where sizing EVERY pair's buffers for them costs 8 x the memory (100 k pairs:
// 170 GB per run) to keep 1 % of the pairs in the fast flow.  Those go through the overflow path instead: the estimate
// stays within twice the median bound.
static int quicked_estimate(std::vector<int32_t>& bounds) {
    if (bounds.empty()) return quicked_estimate(0);
    const int top = *std::max_element(bounds.begin(), bounds.end());
    std::nth_element(bounds.begin(), bounds.begin() + bounds.size() / 2, bounds.end());
    const int median = bounds[bounds.size() / 2];
    return std::min(quicked_estimate(top), 2 * median + 64);
}
// per task: no bound exceeds max(m, n) (the bandwidth percentage only enters stage 3, quicked.c:246)
static int quicked_task_estimate(int est_bound, int longest) { return std::max(1, std::min(est_bound, std::max(longest, 65))); }
static bool quicked_fast_enabled(const Context& C) { return !C.memory_tight && env_int("QE_QUICKED_FAST", 1) != 0; }

// known_s1: stage 1 is known already (the fast flow's leftovers: bound and "goes on to stage 2" per task) -- not run again
struct KnownStage1 { std::vector<int32_t> score; std::vector<uint8_t> stage2; };
static void quicked_classic(quicked_batch& B, Context& C, const quicked_params_t& p, const TaskList& L, bool fetch,
                            size_t matrix_budget, PendingFetch* pf, const std::function<void()>& enter_a, bool whole_batch = true,
                            const KnownStage1* known_s1 = nullptr) {
    double tr_last = now_ms();
    const bool sse = !p.force_scalar;
    const bool want_cigar = !p.only_score;
    // the bound stages need their results on the host to regroup; the driver is synchronous here
    std::vector<int32_t> bound(L.pair.size(), 0);
    if (p.algo == QUICKED) {
        StageResult S1;
        if (known_s1) {
            S1.score = known_s1->score;               // the caller has counted these pairs' stage-1 steps
        } else {
            qe_timer_start(tl_timers.windowed_s);
            run_windowed(B, C, L, false, QUICKED_FAST_WINDOW_SIZE, QUICKED_FAST_WINDOW_OVERLAP, (int)p.hew_threshold[0],
                         true, sse, &S1, true, false, nullptr);
            qe_timer_stop(tl_timers.windowed_s);
            QE_TRACE_POINT("stage 1 windowed");
            B.counters[2] += (int64_t)sum_u32(S1.steps);
        }
        bound = S1.score;
        // stage 2 for the pairs with too many high-error windows (quicked.c:201-202)
        TaskList L2; std::vector<size_t> idx2;
        std::vector<int32_t> stage1_bounds;
        for (size_t t = 0; t < L.pair.size(); ++t) {
            if (L.pair[t] < 0) continue;
            const unsigned mx = (unsigned)std::max(L.m[t], L.n[t]);
            const bool stage2 = known_s1 ? known_s1->stage2[t] != 0 : (uint64_t)S1.hew[t] * 64u > (uint64_t)(mx * p.hew_percentage[0] / 100u);
            if (stage2) {
                L2.push(L.pair[t], 0, L.m[t], 0, L.n[t], 0, L.n[t]); idx2.push_back(t);
            } else stage1_bounds.push_back(S1.score[t]);
        }
        if (whole_batch && B.est_bound >= 0) B.est_bound = quicked_estimate(stage1_bounds);      // what the fast path sizes the next run's align step for
        B.counters[6] = (int64_t)idx2.size();
        for (size_t t : idx2) B.note_pair(L.pair[t], 6, 1);
        if (!idx2.empty()) {
            L2.pad();
            if (!B.have_rev[B.parity]) { launch_pack(B, C, true); B.have_rev[B.parity] = true; }
            StageResult F, V;
            const int W = (int)p.window_size, O = (int)p.overlap_size;
            qe_timer_start(tl_timers.windowed_l);
            {
                // forward and reverse WindowEd(L) side by side (quicked.c:204-235 runs them one after the other): each is a
                // few hundred waves of serial window chains, i.e. latency, and neither needs the other's result
                TaskOut OF, OV;
                hipStream_t main_s = C.stream, side = C.side_stream();
                HIP_CHECK(hipEventRecord(C.ev_fork, main_s)); HIP_CHECK(hipStreamWaitEvent(side, C.ev_fork, 0));
                run_windowed(B, C, L2, false, W, O, (int)p.hew_threshold[1], true, sse, nullptr, false, false, nullptr, nullptr, &OF);
                C.stream = side;
                run_windowed(B, C, L2, true, W, O, (int)p.hew_threshold[1], true, sse, nullptr, false, false, nullptr, nullptr, &OV);
                const size_t nt2 = L2.pair.size();
                d2h(V.score, OV.score, nt2, side); d2h(V.hew, OV.hew, nt2, side); d2h(V.steps, OV.steps, nt2, side);
                C.stream = main_s;
                d2h(F.score, OF.score, nt2, main_s); d2h(F.hew, OF.hew, nt2, main_s); d2h(F.steps, OF.steps, nt2, main_s);
                HIP_CHECK(hipStreamSynchronize(side));
                HIP_CHECK(hipStreamSynchronize(main_s));
            }
            qe_timer_stop(tl_timers.windowed_l);
            B.counters[2] += (int64_t)sum_u32(F.steps) + (int64_t)sum_u32(V.steps);
            for (size_t k = 0; k < idx2.size(); ++k) B.note_pair(L2.pair[k], 2, (int64_t)F.steps[k] + (int64_t)V.steps[k]);
            TaskList L3; std::vector<size_t> idx3;
            for (size_t k = 0; k < idx2.size(); ++k) {
                const size_t t = idx2[k];
                const int64_t sf = F.score[k], sr = V.score[k];
                const int64_t sc = std::min(sf, sr);
                const uint64_t hw = (sc >= sr) ? (uint64_t)V.hew[k] : (uint64_t)F.hew[k];   // quicked.c:229-230
                bound[t] = (int32_t)sc;
                const unsigned mx = (unsigned)std::max(L.m[t], L.n[t]);
                if (hw * 64u * (uint64_t)(p.window_size - p.overlap_size) > (uint64_t)(mx * p.hew_percentage[1] / 100u)) {
                    // stage 3: score-only BandEd, cutoff min(bandwidth%, bound) (quicked.c:246)
                    const int64_t c0 = std::min<int64_t>((int64_t)(mx * p.bandwidth / 100u), sc);
                    bound[t] = (int32_t)c0;
                    L3.push(L.pair[t], 0, L.m[t], 0, L.n[t], (int32_t)c0, L.n[t]); idx3.push_back(t);
                }
            }
            B.counters[7] = (int64_t)idx3.size();
            for (size_t t : idx3) B.note_pair(L.pair[t], 7, 1);
            // band doubling (quicked.c:248-278): relaunch on the subset that has not converged
            int rounds = 0;
            while (!idx3.empty()) {
                if (++rounds > 40) {      // cutoffs double from >= 1: 40 rounds cannot happen for int32 lengths
                    for (size_t k = 0; k < idx3.size() && k < 8; ++k)
                        fprintf(stderr, "[quicked_hip] stage 3 does not converge: pair %d m %d n %d cutoff %d\n",
                                L3.pair[k], L3.m[k], L3.n[k], L3.cutoff[k]);
                    throw HipError{hipErrorUnknown, "QuickEd stage 3 band doubling", __LINE__};
                }
                L3.pad();
                StageResult S3;
                qe_timer_start(tl_timers.banded);
                run_banded_score(B, C, L3, false, &S3, true, nullptr);
                qe_timer_stop(tl_timers.banded);
                B.counters[0] += (int64_t)sum_u32(S3.adv);
                for (size_t k = 0; k < idx3.size(); ++k) B.note_pair(L3.pair[k], 0, (int64_t)S3.adv[k]);
                TaskList Ln; std::vector<size_t> idxn;
                for (size_t k = 0; k < idx3.size(); ++k) {
                    const size_t t = idx3[k];
                    const int64_t ns = S3.score[k], sc = L3.cutoff[k];
                    const int64_t mx = std::max(L.m[t], L.n[t]);
                    if (trace_on() && rounds > 3) fprintf(stderr, "[qe] stage 3 round %d: pair %d m %d n %d cutoff %lld -> %lld\n", rounds, L.pair[t], L.m[t], L.n[t], (long long)sc, (long long)ns);
                    if ((ns > mx / 4 && sc * 3 / 2 < ns) || ns < 0) {
                        // a cutoff of 0 (bandwidth % of a short read rounds to 0) doubles to 0 forever in the reference
                        // (quicked.c:248-278 never terminates there); defined here and in the oracle: doubling starts from 1
                        Ln.push(L.pair[t], 0, L.m[t], 0, L.n[t], (int32_t)std::max<int64_t>(sc * 2, 1), L.n[t]); idxn.push_back(t);
                    } else {
                        bound[t] = (int32_t)ns;
                    }
                }
                L3 = Ln; idx3 = idxn;
            }
        }
    }
    QE_TRACE_POINT("stage 2/3 decisions");
    // align step: bpm_compute_matrix_hirschberg with the bound (quicked.c:283-294)
    TaskList LA;
    // sized like the fast flow's align step where that is possible (no task may split): buffers for the estimate, the
    // bounds themselves handed over as device-side cutoffs
    bool est_sized = p.algo == QUICKED && whole_batch && quicked_fast_enabled(C) && B.est_bound > 0 && !tl_timers.align;
    std::vector<int32_t> est_t;
    if (est_sized) {
        const uint64_t split = split_threshold();
        est_t.assign(L.pair.size(), 0);
        uint64_t mat_bytes = 0;
        for (size_t t = 0; t < L.pair.size() && est_sized; ++t) {
            if (L.pair[t] < 0) continue;
            est_t[t] = std::max(bound[t], quicked_task_estimate(B.est_bound, std::max(L.m[t], L.n[t])));
            const HGeom G = host_geometry(L.m[t], L.n[t], est_t[t]);
            if ((uint64_t)G.ebb * (uint64_t)L.n[t] * 16u > split) est_sized = false;
            mat_bytes += (uint64_t)(QE_CPC + 1) * (uint64_t)(L.n[t] / 64 + 3) * (uint64_t)G.ebb * 16u;     // band_layout's checkpoints
        }
        // Buffers for the estimate are wider than buffers for the bounds (every group is as wide as the batch's widest
        // pair, plus the margin; 400 k pairs of 10 kb: 115 instead of 92 GB per run, cut into fill sub-batches by the
        // planner either way).  A batch whose checkpoints exceed the HBM several times over keeps the classic flow (its
        // runs take seconds: one host round trip is nothing there) until it is reloaded
        if (est_sized && (double)mat_bytes > 1.5 * (double)C.seen_total) { est_sized = false; B.est_bound = -1; }
    }
    for (size_t t = 0; t < L.pair.size(); ++t) {
        if (L.pair[t] < 0) continue;
        LA.push(L.pair[t], 0, L.m[t], 0, L.n[t], est_sized ? est_t[t] : ((p.algo == QUICKED) ? bound[t] : L.cutoff[t]), L.n[t]);
    }
    QE_TRACE_POINT("align task list");
    enter_a();
    qe_timer_start(tl_timers.align);
    AlignStats AS;
    int32_t* d_cut = nullptr; int32_t* d_skip = nullptr;
    if (est_sized) {
        const size_t nt = L.pair.size();
        d_cut = C.scratch_p->take<int32_t>(nt); d_skip = C.scratch_p->take<int32_t>(nt);
        h2d(d_cut, bound, C.stream);
        HIP_CHECK(hipMemsetAsync(d_skip, 0, nt * sizeof(int32_t), C.stream));
    }
    // run_quicked ignores the Hirschberg status (quicked.c:290-291, A.7(8)); run_hirschberg returns it (149-160)
    run_align(B, C, LA, fetch, want_cigar, matrix_budget, split_threshold(), p.algo == QUICKED ? QUICKED_WIP : QUICKED_OK,
              &B.d_score, &AS, pf, /* the bound is an upper bound of the distance */ p.algo == QUICKED, d_cut, d_skip);
    if (pf) pf->quicked = p.algo == QUICKED;
    qe_timer_stop(tl_timers.align);
    QE_TRACE_POINT("align launch(+fetch)");
    B.counters[0] += (int64_t)AS.score_adv; B.counters[1] += (int64_t)AS.fill_adv; B.counters[3] += (int64_t)AS.tb_steps;
    if (fetch && p.algo == QUICKED)
        for (auto& st : B.wr->status) if (st == QUICKED_FAIL_NON_CONVERGENCE) st = QUICKED_WIP;
}

// ---------------------------------------------------------------------------
// QuickEd without the host round trip after stage 1.  On data like the benchmark's no pair ever leaves stage 1, yet the
// classic flow makes the host wait for the WindowEd(2,1) kernel before it can size and queue the align step.  Here the
// stage-1 rule runs on the device (k_stage1_decide), the align step is queued at once with its buffers sized for an
// ESTIMATE of the bounds (1.25 x the largest bound of the previous run; the bandwidth cutoff the first time) and reads
// its cutoffs from the device; pairs that go on to stage 2, or whose bound exceeds the estimate, are taken out of the
// task list on the device and aligned afterwards through the classic flow -- when the results are fetched.  Results are
// those of the classic flow bit for bit: the bound IS the stage-1 score, sizes never enter a result.
// ---------------------------------------------------------------------------
static bool quicked_fast_wanted(const quicked_batch& B, const Context& C, const quicked_params_t& p, const TaskList& L,
                                std::vector<int32_t>& est) {
    if (p.algo != QUICKED || !quicked_fast_enabled(C) || B.est_bound <= 0) return false;   // the first run of a batch is a classic one
    if (tl_timers.align) return false;          // quicked_align: the aligner's stage timers bracket host-synchronous stages
    const int forced = env_int("QE_QUICKED_EST", 0);                   // tests: a small estimate sends pairs through the overflow path
    const uint64_t split = split_threshold();
    est.assign(L.pair.size(), 0);
    for (size_t t = 0; t < L.pair.size(); ++t) {
        if (L.pair[t] < 0) continue;
        const int e = quicked_task_estimate(forced > 0 ? forced : B.est_bound, std::max(L.m[t], L.n[t]));
        est[t] = e;
        // an align step that might split (bpm_hirschberg.c:63-65) needs its real cutoff on the host
        const HGeom G = host_geometry(L.m[t], L.n[t], e);
        if ((uint64_t)G.ebb * (uint64_t)L.n[t] * 16u > split) return false;
    }
    return true;
}

// after a fast run has completed: which pairs still need the classic flow (they keep their stage-1 results: the bound and
// whether stage 2 follows); the next run's estimate
struct FastLeft { TaskList Ls; KnownStage1 K1; };
static bool fast_finish_collect(quicked_batch& B, Context& C, const TaskList& L, const int32_t* d_cut, const int32_t* d_skip,
                                const u32* d_steps, FastLeft& W) {
    const size_t nt = L.pair.size();
    std::vector<int32_t> cut, skip; std::vector<u32> steps;
    d2h(cut, d_cut, nt, C.stream); d2h(skip, d_skip, nt, C.stream); d2h(steps, d_steps, nt, C.stream);
    HIP_CHECK(hipStreamSynchronize(C.stream));
    std::vector<int32_t> stage1_bounds;
    for (size_t t = 0; t < nt; ++t) {
        if (L.pair[t] < 0) continue;
        if (!(skip[t] & 1)) stage1_bounds.push_back(cut[t]);
        if (skip[t]) { W.Ls.push(L.pair[t], 0, L.m[t], 0, L.n[t], L.cutoff[t], L.n[t]); W.K1.score.push_back(cut[t]); W.K1.stage2.push_back((uint8_t)(skip[t] & 1)); }
        B.counters[2] += steps[t];
    }
    B.est_bound = quicked_estimate(stage1_bounds);
    B.wr->deferred_pairs = (int64_t)W.Ls.pair.size();
    return !W.Ls.pair.empty();
}

// what the pairs of a list will need from the pools before their bounds are known (stages 2 / 3 come first): twice what the
// bandwidth cutoff would take -- pairs end up in the host-driven flow because their bounds are large
static size_t classic_need_estimate(const quicked_params_t& p, const TaskList& Ls) {
    size_t want = (size_t)256 << 20;
    for (size_t t = 0; t < Ls.pair.size(); ++t) {
        if (Ls.pair[t] < 0) continue;
        const int m = Ls.m[t], n = Ls.n[t];
        const HGeom G = host_geometry(m, n, 2 * max_cutoff(p.bandwidth, m, n));
        want += (size_t)std::min<uint64_t>((uint64_t)(QE_CPC + 1) * (uint64_t)(n / 64 + 3) * (uint64_t)G.ebb * 16, (uint64_t)18 << 20);
        want += (size_t)std::min<int64_t>((int64_t)m + n + 2, (int64_t)2 * G.cutoff + 8) * 15 + 512;
    }
    return want;
}

// The classic flow for the pairs a fast run left (W), on idle streams.  X is the batch object the pairs belong to, or the
// stand-in for the pairs of several (merged_finish).
static void fast_finish_classic(quicked_batch& X, Context& C, const quicked_params_t& p, FastLeft& W, size_t matrix_budget, int parity) {
    W.Ls.pad();
    W.K1.score.resize(W.Ls.pair.size(), 0); W.K1.stage2.resize(W.Ls.pair.size(), 0);
    C.sync_all();
    struct Restore {            // also when a HIP error unwinds through here
        quicked_batch& B; Context& C; int parity; bool staging; int32_t* score; DevicePool::Mark mw, ma;
        ~Restore() { C.pw().release(mw); C.pa().release(ma); C.phase_u(); B.parity = parity; C.staging = staging; B.d_score = score; }
    } restore{X, C, X.parity, C.staging, X.d_score, C.pw().mark(), C.pa().mark()};
    X.parity = parity;
    C.staging = false;
    // Everything this thread's pools hold is dead by now: its streams are idle (sync_all above), the run being fetched has
    // its results on the host or in the batch's result arena, and the three small arrays read by the collect step were the
    // last thing needed from the pools.  The classic flow for the pairs left starts the pools over instead of stacking its
    // buffers on the fast flow's (76 k of 100 k indel-heavy pairs: 100 GB on top of 170 GB did not fit)
    C.pw().release(DevicePool::Mark{0, 0});
    C.pa().release(DevicePool::Mark{0, 0});
    // the budget is planned again, in THIS context (the caller's fetch, or an early-finish thread's): what the queueing
    // thread's plan allowed one of its pools is an upper limit, the book decides what is there now
    {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t mine = ledger_plan(&C, free_b, classic_need_estimate(p, W.Ls));
        const size_t other_pools = C.held.load() - std::min(C.held.load(), C.pa().cap + C.pw().cap);
        matrix_budget = std::min(matrix_budget, std::max(mine > other_pools ? mine - other_pools : (size_t)0, (size_t)256 << 20));
    }
    C.phase_w();
    auto enter_a = [&]() { C.phase_a(); };
    quicked_classic(X, C, p, W.Ls, true, matrix_budget, nullptr, enter_a, false, &W.K1);
    C.sync_all();
}

static void quicked_fast_finish(quicked_batch& B, Context& C, const quicked_params_t& p, const TaskList& L, const int32_t* d_cut,
                                const int32_t* d_skip, const u32* d_steps, size_t matrix_budget, int parity) {
    FastLeft W;
    if (fast_finish_collect(B, C, L, d_cut, d_skip, d_steps, W)) fast_finish_classic(B, C, p, W, matrix_budget, parity);
}

// The results of a run queued with sync == 0 move from the run's pool set into the batch's result arena at the end of the
// run (device-to-device, on the run's stream: scores and counters are a few hundred KB; the CIGAR strings are copied to
// their real length, which only the device knows).  The fetch then depends on nothing but the batch object.
struct StashItem { void** slot; size_t bytes; };
template <typename T> static void stash_add(std::vector<StashItem>& items, T*& p, size_t bytes) {
    if (p && bytes) items.push_back(StashItem{(void**)&p, bytes});
}
static void stash_results(quicked_batch& B, Context& C, PendingFetch& F) {
    std::vector<StashItem> items;
    const size_t nt = F.task_pair.size(), nr = F.AO.nroots, nl = F.leaf_pair.size(), nq = F.L.pair.size();
    stash_add(items, F.d_score, nt * 4); stash_add(items, F.d_adv, nt * 4); stash_add(items, F.d_steps, nt * 4); stash_add(items, F.d_abort, nt * 4);
    if (F.kind == 2) {
        stash_add(items, F.AO.len, nr * 4); stash_add(items, F.AO.edits, nr * 4); stash_add(items, F.AO.nops, nr * 4);
        stash_add(items, F.AO.ok, nr * 4); stash_add(items, F.AO.str_off, nr * 8);
        stash_add(items, F.d_leaf_adv, nl * 4); stash_add(items, F.d_leaf_steps, nl * 4);
        stash_add(items, F.d_cut, nq * 4); stash_add(items, F.d_skip, nq * 4); stash_add(items, F.d_stage_steps, nq * 4);
    }
    const bool strings = F.kind == 2 && F.want_strings && F.AO.pool && F.AO.total;
    size_t need = 256;
    for (const StashItem& it : items) need += (it.bytes + 255) & ~(size_t)255;
    if (strings) need += ((F.AO.pool_bytes + 255) & ~(size_t)255) + 256;
    if (B.result_bytes < need) {
        if (B.result_arena) { HIP_CHECK(hipDeviceSynchronize()); device_free(B.result_arena, B.device); B.result_arena = nullptr; B.result_bytes = 0; }
        const size_t cap = need + need / 8;
        device_malloc((void**)&B.result_arena, cap, C.device, &C.pa(), "hipMalloc(batch result arena)", __LINE__);
        B.result_bytes = cap;
    }
    size_t top = 0;
    CopyTable T;
    T.n = 0;
    int64_t most = 1;
    auto flush = [&]() {
        if (T.n == 0) return;
        hipLaunchKernelGGL(k_copy_multi, dim3((unsigned)std::min<int64_t>(64, (most + 255) / 256), (unsigned)T.n), dim3(256), 0, C.stream, T);
        T.n = 0; most = 1;
    };
    auto copy = [&](void* dst, const void* src, size_t bytes) {
        T.dst[T.n] = (uint4*)dst; T.src[T.n] = (const uint4*)src; T.n_u4[T.n] = (int64_t)((bytes + 15) >> 4);
        most = std::max(most, T.n_u4[T.n]);
        if (++T.n == CopyTable::MAX) flush();
    };
    for (const StashItem& it : items) {
        void* dst = B.result_arena + top;
        copy(dst, *it.slot, it.bytes);
        *it.slot = dst;
        top += (it.bytes + 255) & ~(size_t)255;
    }
    if (strings) {
        int64_t* d_total = (int64_t*)(B.result_arena + top); top += 256;
        copy(d_total, F.AO.total, 8);
        flush();
        char* dst = (char*)(B.result_arena + top);
        hipLaunchKernelGGL(k_copy_total, dim3(2048), dim3(256), 0, C.stream, (uint4*)dst, (const uint4*)F.AO.pool, (const int64_t*)F.AO.total,
                           (int64_t)(F.AO.pool_bytes >> 4));
        F.AO.total = d_total; F.AO.pool = dst;
    }
    flush();
}

// sets of {streams, pools, planes} that rotate for a batch of n pairs: enough runs in flight for ~2048 waves (two per SIMD)
static int rotation_depth(int64_t n, int floor_sets = 5) {
    const int64_t groups = std::max<int64_t>(1, (n + 63) / 64);
    return (int)std::max<int64_t>(floor_sets, std::min<int64_t>(Context::NA, (2048 + groups - 1) / groups));
}

static void finisher_submit(quicked_batch& B, const std::shared_ptr<void>& pf);

static quicked_status_t run_batch(quicked_batch& B, const quicked_params_t& p, bool fetch) {
    double tr_last = now_ms();
    tl_device = B.device;
    Context& C = ctx();

    // ---- phase W: pack + bound stages on stream_w with pool_w.  The planes of this set were last read by the A
    // phase of the run that used it before (np_used runs ago): wait for it on the device, not on the host.
    // BandEd and WindowEd have no bound stage: their pack goes on stream_a, in order with the kernel.  Overlapping it
    // with the previous run's kernel would save ~0.5 ms, but a 25 k-workgroup kernel dispatched next to the 1563
    // one-wave workgroups of k_banded skews their placement over the SIMDs and doubles the kernel's time.
    const bool serial = p.algo == BANDED || p.algo == WINDOWED;
    // Three sets: up to three runs of a thread are on the device at once (measured on 100 k x 10 kb: 6.69 -> 6.88 M/s
    // BandEd, 4.70 -> 5.31 M/s QuickEd + CIGAR against two; four are slower again).  Two when three fill matrices of the
    // size this batch needed last time would not fit (config 4: 94 GB each): sub-batching the fill costs more.
    DeviceBook& book = g_book[C.device];
    // another thread ran out of memory after every reclaim: this one gives its pools back (its runs are waited for) and runs
    // with one set for a while
    if (book.pressure.load() != C.pressure_seen) {
        C.pressure_seen = book.pressure.load();
        C.go_tight();
        (void)C.release_pools(nullptr, true);
    }
    // single quicked_align calls plan with the last reading of the device's free memory while nothing has been allocated
    // or freed by the library since (the query costs tens of microseconds)
    size_t free0 = C.seen_free, total0 = C.seen_total;
    if (C.seen_total == 0 || B.arena_bytes > ((size_t)64 << 20) || C.seen_epoch != book.epoch.load()) {
        const uint64_t ep = book.epoch.load();
        HIP_CHECK(hipMemGetInfo(&free0, &total0));
        C.seen_free = free0; C.seen_total = total0; C.seen_epoch = ep;
    }
    // ---- the device-pool planner (replaces mm_allocator's "never fails" arena, mm_allocator.c:251-334, by a budget):
    // how many {stream, pool, planes} sets rotate, and how many bytes one pool may hold, from what this batch's last
    // CIGAR run needed (or, first time, from a bandwidth-based estimate).  A run whose fill matrices do not fit its
    // pool's budget is cut into sub-batches by run_align -- before anything is allocated, not after an out-of-memory.
    size_t pools_held = 0;
    for (const auto& q : C.pool_a2) pools_held += q.cap;
    size_t need_mat = B.last_mat_bytes, need_fixed = B.last_fixed_bytes;
    int need_groups = B.last_groups;
    if (need_groups == 0 && !p.only_score && p.algo != WINDOWED) {
        // first CIGAR run of this batch: leaves as wide as the bandwidth cutoff allows (QuickEd's bounds are tighter)
        for (int64_t i = 0; i < B.n; ++i) {
            const int m = B.p_len[(size_t)i], n = B.t_len[(size_t)i];
            if (m == 0 || n == 0) continue;
            const HGeom G = host_geometry(m, n, max_cutoff(p.bandwidth, m, n));
            const uint64_t full = (uint64_t)(QE_CPC + 1) * (uint64_t)(n / 64 + 3) * (uint64_t)G.ebb * 16;
            need_mat += (size_t)std::min<uint64_t>(full, (uint64_t)18 << 20);             // per pair; splits cap a leaf at 16 MiB of matrix
            need_fixed += (size_t)(p.algo == QUICKED ? std::min<int64_t>((int64_t)m + n + 2, (int64_t)2 * G.cutoff + 8) : (int64_t)m + n + 2) * 15 + 512;
        }
        need_groups = (int)((B.n + 63) / 64);
    }
    // a sub-batch should still fill the chip: >= ~1600 groups (two waves on every SIMD) where the batch has that many
    const double frac = need_groups > 1600 ? 1600.0 / (double)need_groups : 1.0;
    const size_t min_set = need_fixed + (size_t)((double)need_mat * frac);
    // depth of the rotation: three sets for batches that fill the chip; a small batch (12.5 k pairs = 196 waves of ~11 ms)
    // needs more runs in flight to keep two waves on every SIMD.  A synchronous run is alone on the device anyway.
    // (large batches: three sets for the one-kernel flows -- a 100 k-pair BandEd kernel nearly fills the chip, a fourth run only
    // queues; five for QuickEd / Hirschberg, whose runs are chains of kernels of different shapes: 5.96 -> 6.24 M alignments/s)
    const int depth_wanted = fetch ? 3 : rotation_depth(B.n, serial ? 3 : 5);
    const size_t wanted = (size_t)(1.05 * (double)std::min(depth_wanted, B.np_alloc) * (double)(need_fixed + need_mat)) + ((size_t)256 << 20);
    // what threads that have ended left in their contexts is reused by the next thread that takes the context over; it
    // goes back to the device when this plan could use the room
    if (wanted > pools_held && wanted - pools_held > free0 / 2 && unleased_held(C.device) > 0 && release_unleased(C.device)) {
        HIP_CHECK(hipMemGetInfo(&free0, &total0));
        C.seen_free = free0; C.seen_total = total0; C.seen_epoch = book.epoch.load();
    }
    // what the A pools of this thread may hold together: the device's free memory plus what its pools hold already, less
    // what the process's other contexts hold or have planned (the book, qe_pool.h), less this context's other pools
    size_t owed = 0;
    const size_t mine = ledger_plan(&C, free0, wanted, &owed);
    const size_t not_a = C.held.load() - std::min(C.held.load(), pools_held);
    const size_t avail = mine > not_a ? mine - not_a : 0;
    int na = 1;
    // beyond three sets only with room to spare -- the plan does not see the batches' result arenas or what the caller
    // allocates next -- and within 40 % of the device: depth is for small batches, whose sets are small
    int sets_held = 0;                                    // sets whose pools exist already: rotating over them costs nothing
    for (int q = 0; q < Context::NA; ++q) if (C.pool_a2[q].cap > ((size_t)1 << 28)) sets_held = q + 1;
    for (int k = std::min(depth_wanted, B.np_alloc); k >= 1; --k) {
        const bool deep = k > 3;
        if (deep && (double)min_set * k > 0.4 * (double)total0) continue;
        if ((double)min_set * k <= ((deep && k > sets_held) ? 0.6 : 1.0) * (double)avail) { na = k; break; }
    }
    if (C.memory_tight && C.tight_left-- <= 0) { C.memory_tight = false; C.tight_left = 0; }      // the spell is over: plan normally again
    if (C.memory_tight) na = 1;
    // Sets outside the rotation keep their pools while this run's plan works without that memory -- the next batch may
    // widen the rotation again, and freeing / re-allocating tens of GB per run costs more than any of this saves (a stream
    // of batches whose plans alternated between 3 and 5 sets ran at 0.7 M alignments/s) -- and give them back when it does not
    size_t idle_held = 0;
    for (int q = na; q < Context::NA; ++q) idle_held += C.pool_a2[q].cap + C.pool_w2[q].cap;
    // (a fill that does not fit its pool's budget is cut into sub-batches; that is cheaper than giving pools back too)
    const bool keep_idle = idle_held > 0 && (avail > idle_held) && (avail - idle_held) / (size_t)na > need_fixed + ((size_t)4 << 30);
    C.pool_budget = (keep_idle ? avail - idle_held : avail) / (size_t)na;
    C.last_na = na;
    C.in_flight = fetch ? 1 : na;
    B.np_used = na;
    for (int q = 0; q < na && owed > 0; ++q)            // a set that outgrew this thread's share while another thread waits for the room
        if (C.pool_a2[q].cap > ((size_t)1 << 30) && (double)C.pool_a2[q].cap > 1.25 * (double)C.pool_budget) {
            if (C.stream_a2[q]) HIP_CHECK(hipStreamSynchronize(C.stream_a2[q]));
            C.pool_a2[q].release_all();
        }
    for (int q = na; q < Context::NA && !keep_idle; ++q) {      // a set that left the rotation gives its memory back
        if (C.stream_a2[q] && (C.pool_a2[q].cap > ((size_t)1 << 30) || C.pool_w2[q].cap > ((size_t)1 << 30))) HIP_CHECK(hipStreamSynchronize(C.stream_a2[q]));
        if (C.pool_a2[q].cap > ((size_t)1 << 30)) C.pool_a2[q].release_all();
        if (C.pool_w2[q].cap > ((size_t)1 << 30)) C.pool_w2[q].release_all();
    }
    C.ai = (C.ai + 1) % na;
    C.ensure_set(C.ai);
    const int par = B.parity = (B.parity + 1) % B.np_used;
    C.si = (C.si + 1) % (2 * na);
    {
        PinnedStage& st = C.stage[C.si];
        if (!st.done) HIP_CHECK(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
        st.reset();                           // its last user is 2 na runs back: over unless the host is that far ahead
        C.staging = true;
    }
    if (serial) {
        C.phase_a(); C.pa().reset();
        if (B.ev_done_set[par]) HIP_CHECK(hipStreamWaitEvent(C.sa(), B.ev_done[par], 0));
    }
    else {
        C.phase_w();
        C.pw().reset();
        if (C.decided_set[C.ai]) { HIP_CHECK(hipStreamWaitEvent(C.sw(), C.ev_decided[C.ai], 0)); C.decided_set[C.ai] = false; }
        if (B.ev_done_set[par]) HIP_CHECK(hipStreamWaitEvent(C.sw(), B.ev_done[par], 0));
    }
    B.only_score_run = p.only_score;
    // sync == 0 leaves the host-side results of the last fetched run untouched (quicked_batch_fetch brings this run's)
    B.pending_fetch.reset();
    B.shadow_ready = false;                    // an early finish of the previous queued run is superseded
    B.wr = &B.res[B.vis];
    std::shared_ptr<PendingFetch> pfp;
    if (fetch) reset_host_results(B);
    else pfp = std::make_shared<PendingFetch>();
    PendingFetch* const pf = pfp.get();
    for (auto& c : B.counters) c = 0;
    if ((unsigned)p.algo > (unsigned)HIRSCHBERG) {
        if (fetch) std::fill(B.wr->status.begin(), B.wr->status.end(), (int32_t)QUICKED_UNKNOWN_ALGO);
        C.staging = false;
        C.phase_u();
        return QUICKED_UNKNOWN_ALGO;
    }
    HIP_CHECK(hipEventRecord(C.ev0, C.stream));
    if (B.packed && B.unpack_pending) {
        // wire words -> planes, once per (re)load, on this run's stream; later runs of the batch (other streams) wait for it
        HIP_CHECK(hipMemsetAsync(B.d_flags[0], 0, (size_t)B.n * sizeof(u32), C.stream));
        const int blocks = (int)((B.n + 3) / 4);
        WireArgs w;
        w.nseq = (int32_t)B.n; w.wire = B.wire; w.flags = B.d_flags[0];
        w.words = B.d_wire_p; w.w_off = B.d_wire_p_off; w.len = B.d_p_len; w.planes = B.d_pl_p[0]; w.pl_off = B.d_plp_off;
        hipLaunchKernelGGL(k_unpack_wire, dim3(blocks), dim3(256), 0, C.stream, w);
        w.words = B.d_wire_t; w.w_off = B.d_wire_t_off; w.len = B.d_t_len; w.planes = B.d_pl_t[0]; w.pl_off = B.d_plt_off;
        hipLaunchKernelGGL(k_unpack_wire, dim3(blocks), dim3(256), 0, C.stream, w);
        HIP_CHECK(hipEventRecord(B.ev_unpacked, C.stream));
        B.unpack_pending = false; B.unpack_event_set = true;
    } else if (B.packed && B.unpack_event_set) {
        HIP_CHECK(hipStreamWaitEvent(C.stream, B.ev_unpacked, 0));
    }
    if (!B.packed) HIP_CHECK(hipMemsetAsync(B.d_flags[par], 0, (size_t)B.n * sizeof(u32), C.stream));
    launch_pack(B, C, false);
    if (!serial) HIP_CHECK(hipEventRecord(C.ev_pack, C.sw()));
    // phase A starts on the device when the planes are there and (stream order) the previous run's A phase is over;
    // its pool can be reset now because everything it launches is ordered behind that previous A phase
    auto enter_a = [&]() {
        if (serial) return;
        C.phase_a();
        C.pa().reset();
        HIP_CHECK(hipStreamWaitEvent(C.sa(), C.ev_pack, 0));
    };
    const bool sse = !p.force_scalar;
    const bool want_cigar = !p.only_score;
    const size_t matrix_budget = C.pool_budget;      // run_align subtracts what the stage needs besides the matrices
    quicked_status_t ret = QUICKED_WIP;
    QE_TRACE_POINT("setup+pack launch");
    TaskList L = all_pairs(B, p);
    QE_TRACE_POINT("task list");
    if (L.pair.empty()) { C.staging = false; HIP_CHECK(hipStreamSynchronize(C.stream)); C.phase_u(); return QUICKED_EMPTY_SEQUENCE; }
    StageResult R;

    switch (p.algo) {
    case BANDED:                                                    // run_banded, quicked.c:58-89
        enter_a();
        if (p.only_score) {
            run_banded_score(B, C, L, false, &R, fetch, &B.d_score, pf);
            if (fetch) {
                scatter_scores(B, L, R.score, QUICKED_WIP);
                B.counters[0] = (int64_t)sum_u32(R.adv);
                for (int32_t x : R.hew) B.counters[6] += (x != 0);     // tasks the cooperative kernel handed to the fallback pass
            }
        } else {
            AlignStats AS;          // run_banded never splits: one fill + traceback whatever the size
            run_align(B, C, L, fetch, want_cigar, matrix_budget, ~(uint64_t)0, QUICKED_WIP, &B.d_score, &AS, pf);
            B.counters[1] = (int64_t)AS.fill_adv; B.counters[3] = (int64_t)AS.tb_steps;
        }
        break;
    case WINDOWED:                                                  // run_windowed, quicked.c:91-123
        enter_a();
        run_windowed(B, C, L, false, (int)p.window_size, (int)p.overlap_size, 0, p.only_score, sse, &R, fetch,
                     want_cigar, &B.d_score, pf);
        if (fetch) { scatter_scores(B, L, R.score, QUICKED_WIP); B.counters[2] = (int64_t)sum_u32(R.steps); }
        break;
    case QUICKED:                                                   // run_quicked, quicked.c:163-306
    case HIRSCHBERG: {                                              // run_hirschberg, quicked.c:125-161
        std::vector<int32_t> est;
        if (quicked_fast_wanted(B, C, p, L, est)) {
            TaskOut W1; DevTasks T1;
            qe_timer_start(tl_timers.windowed_s);
            run_windowed(B, C, L, false, QUICKED_FAST_WINDOW_SIZE, QUICKED_FAST_WINDOW_OVERLAP, (int)p.hew_threshold[0], true, sse,
                         nullptr, false, false, nullptr, nullptr, &W1, &T1);
            qe_timer_stop(tl_timers.windowed_s);
            QE_TRACE_POINT("fast: stage 1 queued");
            HIP_CHECK(hipEventRecord(C.ev_stage, C.sw()));
            enter_a();
            QE_TRACE_POINT("fast: phase A entered");
            HIP_CHECK(hipStreamWaitEvent(C.sa(), C.ev_stage, 0));
            const size_t nt = L.pair.size();
            int32_t* d_cut = C.scratch_p->take<int32_t>(nt); int32_t* d_skip = C.scratch_p->take<int32_t>(nt);
            u32* d_steps = C.scratch_p->take<u32>(nt); int32_t* d_est = C.scratch_p->take<int32_t>(nt);
            h2d(d_est, est, C.stream);
            Stage1Args sa;
            sa.nt = (int32_t)nt; sa.pair = T1.pair; sa.m = T1.m; sa.n = T1.n; sa.score = W1.score; sa.hew = W1.hew; sa.steps = W1.steps;
            sa.est = d_est; sa.hew_percentage = p.hew_percentage[0]; sa.o_cut = d_cut; sa.o_skip = d_skip; sa.o_steps = d_steps;
            hipLaunchKernelGGL(k_stage1_decide, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, C.stream, sa);
            HIP_CHECK(hipEventRecord(C.ev_decided[C.ai], C.stream));    // the stage's outputs live in the set's W pool, which the set's next run recycles
            C.decided_set[C.ai] = true;
            QE_TRACE_POINT("fast: decide queued");
            TaskList LA;
            for (size_t t = 0; t < nt; ++t) {
                if (L.pair[t] < 0) continue;
                LA.push(L.pair[t], 0, L.m[t], 0, L.n[t], est[t], L.n[t]);
            }
            QE_TRACE_POINT("fast: stage 1 + align list queued");
            qe_timer_start(tl_timers.align);
            AlignStats AS;
            run_align(B, C, LA, fetch, want_cigar, matrix_budget, split_threshold(), QUICKED_WIP, &B.d_score, &AS, pf, true, d_cut, d_skip);
            qe_timer_stop(tl_timers.align);
            B.counters[1] += (int64_t)AS.fill_adv; B.counters[3] += (int64_t)AS.tb_steps;
            if (pf) {
                pf->quicked = true; pf->fast = true; pf->d_cut = d_cut; pf->d_skip = d_skip; pf->d_stage_steps = d_steps;
                pf->params = p; pf->L = L; pf->matrix_budget = matrix_budget;
            }
            if (fetch) {
                quicked_fast_finish(B, C, p, L, d_cut, d_skip, d_steps, matrix_budget, par);
                C.phase_a();
                for (auto& st : B.wr->status) if (st == QUICKED_FAIL_NON_CONVERGENCE) st = QUICKED_WIP;
            }
            QE_TRACE_POINT("fast: align launched(+fetch)");
        } else
            quicked_classic(B, C, p, L, fetch, matrix_budget, pf, enter_a);
        ret = (p.algo == QUICKED) ? QUICKED_WIP : QUICKED_OK;
        break;
    }
    default: break;
    }
    if (pf && pf->kind != 0) {
        C.phase_a();
        // the batch has ONE result arena: the previous queued run of this batch (another stream of the rotation, possibly a
        // longer chain of kernels) must have put its results there before this run's overwrite them
        if (B.last_parity >= 0 && B.last_parity != par && B.ev_done_set[B.last_parity]) HIP_CHECK(hipStreamWaitEvent(C.stream, B.ev_done[B.last_parity], 0));
        stash_results(B, C, *pf);
    }
    HIP_CHECK(hipEventRecord(C.ev1, C.stream));
    HIP_CHECK(hipEventRecord(C.ev_last, C.sa()));
    HIP_CHECK(hipEventRecord(B.ev_done[par], C.sa()));
    B.last_parity = par;
    if (C.staging) { HIP_CHECK(hipEventRecord(C.stage[C.si].done, C.sa())); C.stage[C.si].pending = true; C.staging = false; }
    QE_TRACE_POINT("stages launched");
    {   // pre-size the other pools of the rotation -- only while that is cheap: big fill matrices are left to grow on demand
        size_t cap_all = 0;
        for (const auto& q : C.pool_a2) cap_all += q.cap;
        if ((double)(cap_all + (size_t)(na - 1) * C.pa().cap) < 0.78 * (double)total0)
            for (int q = 0; q < na; ++q) if (q != C.ai) C.pool_a2[q].mirror(C.pa());
    }
    QE_TRACE_POINT("pool mirror");
    B.ev_done_set[par] = true;
    if (pf && pf->kind != 0) {
        pf->parity = par;
        for (int q = 0; q < 8; ++q) pf->counters[q] = B.counters[q];
        B.pending_fetch = pfp;
        B.fin_status = QUICKED_OK;
        if (pf->fast) finisher_submit(B, pfp);         // pairs that left stage 1 are finished as soon as the run is over
    }
    C.phase_u();
    B.pending = true;
    if (fetch) {
        C.sync_all();
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, C.ev0, C.ev1));
        B.counters[5] = (int64_t)(ms * 1e6);
        B.pending = false;
        memcpy(B.wr->counters, B.counters, sizeof(B.counters));
    }
    return ret;
}


// quicked_batch_fetch: waits for the batch's last sync == 0 run and copies its results to the host-side arrays the
// getters read -- what a sync != 0 run does at its end, only later (so that further runs can be queued meanwhile)
// left != nullptr: the pairs a fast QuickEd run left for the host-driven flow are only LISTED (merged early finish: one flow
// for the pairs of several batch objects); the caller then runs it and calls fetch_finalize
static void fetch_finalize(quicked_batch& B, bool quicked) {
    if (quicked) for (auto& st : B.wr->status) if (st == QUICKED_FAIL_NON_CONVERGENCE) st = QUICKED_WIP;
    memcpy(B.wr->counters, B.counters, sizeof(B.counters));
    B.pending = false;
}
static quicked_status_t fetch_pending(quicked_batch& B, FastLeft* left = nullptr) {
    tl_device = B.device;
    Context& C = ctx();
    if (!B.pending_fetch) return B.pending ? QUICKED_ERROR : QUICKED_OK;      // nothing queued asynchronously
    std::shared_ptr<void> hold = B.pending_fetch;
    PendingFetch& F = *static_cast<PendingFetch*>(hold.get());
    B.pending_fetch.reset();
    HIP_CHECK(hipEventSynchronize(B.ev_done[F.parity]));
    C.phase_u();
    reset_host_results(B);
    for (int q = 0; q < 8; ++q) B.counters[q] = F.counters[q];
    if (F.kind == 1) {
        const size_t nt = F.task_pair.size();
        std::vector<int32_t> sc, ab; std::vector<u32> w;
        d2h(sc, F.d_score, nt, C.stream);
        if (F.d_adv) d2h(w, F.d_adv, nt, C.stream); else if (F.d_steps) d2h(w, F.d_steps, nt, C.stream);
        if (F.d_abort) d2h(ab, F.d_abort, nt, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        for (size_t t = 0; t < nt; ++t) {
            const int pr = F.task_pair[t];
            if (pr < 0) continue;
            B.wr->score[pr] = sc[t]; B.wr->status[pr] = F.ok_status;
        }
        B.counters[F.counter_slot] += (int64_t)sum_u32(w);
        for (int32_t x : ab) B.counters[6] += (x != 0);
    } else {
        if (F.d_leaf_adv) {
            const size_t nt = F.leaf_pair.size();
            std::vector<u32> adv, steps;
            d2h(adv, F.d_leaf_adv, nt, C.stream); d2h(steps, F.d_leaf_steps, nt, C.stream);
            HIP_CHECK(hipStreamSynchronize(C.stream));
            for (size_t t = 0; t < nt; ++t) if (F.leaf_pair[t] >= 0) { B.counters[1] += adv[t]; B.counters[3] += steps[t]; }
        } else if (F.d_steps) {
            std::vector<u32> steps;
            d2h(steps, F.d_steps, F.task_pair.size(), C.stream);
            HIP_CHECK(hipStreamSynchronize(C.stream));
            B.counters[2] += (int64_t)sum_u32(steps);
        }
        fetch_alignments(B, C, F.SL, F.AO, F.want_strings, F.ok_status, F.root_status.empty() ? nullptr : &F.root_status);
        if (F.fast && left) {
            if (fast_finish_collect(B, C, F.L, F.d_cut, F.d_skip, F.d_stage_steps, *left)) return QUICKED_OK;      // the caller goes on
        } else if (F.fast) quicked_fast_finish(B, C, F.params, F.L, F.d_cut, F.d_skip, F.d_stage_steps, F.matrix_budget, F.parity);
    }
    fetch_finalize(B, F.kind != 1 && F.quicked);
    return QUICKED_OK;
}

// ---------------------------------------------------------------------------
// Early finish.  A QuickEd run queued with sync == 0 goes through the fast flow; the pairs that leave stage 1 (or outgrow
// the estimate) are aligned through the host-driven flow when the run's results are fetched -- a chain of small launches,
// ~70 ms for a few hundred pairs of 10 kb whatever their number.  Left to the caller's fetch, a stream of batches with 1 %
// of such pairs ran at one batch per chain (0.25-1 M alignments/s against 6 M without them) unless the caller fetched
// from several threads.  So a few library threads do it as soon as a run is over: a job waits for the run's event, looks
// at the skip flags in the batch's result arena and, when there are pairs to finish, does what quicked_batch_fetch would
// (results to the host-side arrays, the deferred pairs through quicked_classic in the finisher's own context).  The
// caller's fetch then finds the work done.  Runs without deferred pairs are left alone (one 4-byte-per-pair read).
// QE_FINISHERS = 0 switches it off; default 3 threads, started on demand, detached (they sleep on the queue).
// ---------------------------------------------------------------------------
struct FinishJob { quicked_batch* B; std::shared_ptr<void> pf; };
static std::mutex& g_fin_mu = *new std::mutex;                      // never destroyed: detached threads wait on them at exit
static std::condition_variable& g_fin_cv = *new std::condition_variable;
static std::deque<FinishJob>& g_fin_q = *new std::deque<FinishJob>;
static int g_fin_threads = 0, g_fin_idle = 0;
static std::atomic<int> g_fin_busy{0};               // jobs being worked on
static std::atomic<bool> g_fin_stop{false};          // the process is exiting: no new work

// One host-driven flow for the pairs that SEVERAL queued runs (of different batch objects) left: the flow's duration is a
// chain of launch latencies -- 70-90 ms for 800 pairs or 4 000 alike -- so the pairs of every run that is over when an
// early-finish thread gets to work are aligned together.  A stand-in batch object V holds those pairs: their lengths, the
// addresses of their raw bytes in the batches' arenas (the validator and non-canonical input compare bytes), and a COPY of
// their forward bit-planes, gathered into one buffer (a few MB); reversed planes come from those (k_reverse_planes, as for
// packed input).  Results and the per-pair share of every work counter go back to the batch each pair came from.
struct MergeItem {
    quicked_batch* B; std::shared_ptr<void> pf; FastLeft W;
    std::unique_lock<std::mutex> lk;                  // the batch's fin_mu, held from the collect step to the end
    std::shared_ptr<void> keep;                       // what a failure puts back as the batch's pending fetch
};
// (the first `ni` entries of items)
static void merged_finish(std::vector<MergeItem>& items, size_t ni, Context& C) {
    const PendingFetch& F0 = *static_cast<const PendingFetch*>(items[0].pf.get());
    const quicked_params_t p = F0.params;
    C.phase_u();
    quicked_batch*& V = *reinterpret_cast<quicked_batch**>(&C.merge_batch);
    if (!V) V = new quicked_batch();
    // ---- the pairs, renumbered 0 .. n-1 in item order
    size_t n = 0;
    for (size_t x = 0; x < ni; ++x) n += items[x].W.Ls.pair.size();
    std::vector<int32_t> src_item(n), src_pair(n);
    FastLeft WV;
    V->n = (int64_t)n; V->device = C.device;
    V->p_len.resize(n); V->t_len.resize(n); V->p_off.resize(n); V->t_off.resize(n); V->plp_off.resize(n); V->plt_off.resize(n);
    V->order.resize(n);
    std::iota(V->order.begin(), V->order.end(), 0);
    std::vector<int64_t> g_src(2 * n), g_dst(2 * n);
    std::vector<int32_t> g_nw(2 * n);
    std::vector<u32> flags(n, 0);
    const uint8_t* asc_base_p = items[0].B->d_asc_p; const uint8_t* asc_base_t = items[0].B->d_asc_t;
    size_t wp = 0, wt = 0, j = 0;
    size_t budget = 0;
    for (size_t x = 0; x < ni; ++x) {
        quicked_batch& B = *items[x].B;
        const PendingFetch& F = *static_cast<const PendingFetch*>(items[x].pf.get());
        budget = std::max(budget, F.matrix_budget);
        std::vector<u32> bf;                              // the batch's pack flags (N / non-canonical symbols) of the run's plane set
        d2h(bf, (const u32*)B.d_flags[F.parity], (size_t)B.n, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        const FastLeft& W = items[x].W;
        for (size_t t = 0; t < W.Ls.pair.size(); ++t, ++j) {
            const int pr = W.Ls.pair[t];
            src_item[j] = (int32_t)x; src_pair[j] = pr;
            V->p_len[j] = B.p_len[(size_t)pr]; V->t_len[j] = B.t_len[(size_t)pr];
            V->p_off[j] = (B.d_asc_p + B.p_off[(size_t)pr]) - asc_base_p;         // byte offsets from the first batch's pools: any sign
            V->t_off[j] = (B.d_asc_t + B.t_off[(size_t)pr]) - asc_base_t;
            const int32_t nwp = 3 * ((V->p_len[j] + 63) / 64 + 2), nwt = 3 * ((V->t_len[j] + 63) / 64 + 2);
            V->plp_off[j] = (int64_t)wp; V->plt_off[j] = (int64_t)wt;
            g_src[2 * j] = (int64_t)(uintptr_t)(B.d_pl_p[F.parity] + B.plp_off[(size_t)pr]); g_nw[2 * j] = nwp;
            g_src[2 * j + 1] = (int64_t)(uintptr_t)(B.d_pl_t[F.parity] + B.plt_off[(size_t)pr]); g_nw[2 * j + 1] = nwt;
            wp += (size_t)nwp; wt += (size_t)nwt;
            flags[j] = bf[(size_t)pr];
            WV.Ls.push((int32_t)j, 0, W.Ls.m[t], 0, W.Ls.n[t], W.Ls.cutoff[t], W.Ls.n[t]);
            WV.K1.score.push_back(W.K1.score[t]); WV.K1.stage2.push_back(W.K1.stage2[t]);
        }
    }
    for (size_t q = 0; q < n; ++q) { g_dst[2 * q] = V->plp_off[q]; g_dst[2 * q + 1] = (int64_t)wp + V->plt_off[q]; }
    // ---- V's device side: carved from this thread's utility pool for the length of the flow
    struct PoolMark { Context& C; DevicePool::Mark m; ~PoolMark() { C.pool_w.release(m); C.util_pinned = false; } } pm{C, C.pool_w.mark()};
    C.util_pinned = true;
    DevicePool& U = C.pool_w;
    V->d_asc_p = const_cast<uint8_t*>(asc_base_p); V->d_asc_t = const_cast<uint8_t*>(asc_base_t);
    V->d_p_off = U.take<int64_t>(n); V->d_t_off = U.take<int64_t>(n); V->d_plp_off = U.take<int64_t>(n); V->d_plt_off = U.take<int64_t>(n);
    V->d_p_len = U.take<int32_t>(n); V->d_t_len = U.take<int32_t>(n);
    u64* planes = U.take<u64>(wp + wt + 8); u64* planes_r = U.take<u64>(wp + wt + 8);
    u32* d_fl = U.take<u32>(n);
    int64_t* d_gsrc = U.take<int64_t>(2 * n); int64_t* d_gdst = U.take<int64_t>(2 * n); int32_t* d_gnw = U.take<int32_t>(2 * n);
    V->pl_p_words = wp; V->pl_t_words = wt;
    for (int q = 0; q < quicked_batch::NP; ++q) {          // every plane set is the same gathered copy
        V->d_pl_p[q] = planes; V->d_pl_t[q] = planes + wp; V->d_pl_pr[q] = planes_r; V->d_pl_tr[q] = planes_r + wp;
        V->d_flags[q] = d_fl; V->have_rev[q] = false; V->ev_done_set[q] = false;
    }
    V->np_alloc = quicked_batch::NP; V->np_used = 1; V->parity = 0; V->last_parity = -1;
    V->packed = true;                                      // forward planes are the input: no pack stage, reversed planes from them
    V->unpack_pending = false; V->unpack_event_set = false;
    V->cigar_style = items[0].B->cigar_style; V->check = false;
    V->est_bound = 0; V->pending = false; V->pending_fetch.reset();
    h2d(V->d_p_off, V->p_off, C.stream); h2d(V->d_t_off, V->t_off, C.stream);
    h2d(V->d_plp_off, V->plp_off, C.stream); h2d(V->d_plt_off, V->plt_off, C.stream);
    h2d(V->d_p_len, V->p_len, C.stream); h2d(V->d_t_len, V->t_len, C.stream);
    h2d(d_fl, flags, C.stream); h2d(d_gsrc, g_src, C.stream); h2d(d_gdst, g_dst, C.stream); h2d(d_gnw, g_nw, C.stream);
    hipLaunchKernelGGL(k_gather_words, dim3((unsigned)((2 * n + 3) / 4)), dim3(256), 0, C.stream, (int)(2 * n), d_gsrc, d_gdst, d_gnw, planes);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(C.stream));
    // ---- the flow, once
    V->wr = &V->res[0]; V->vis = 0; V->shadow_ready = false;
    reset_host_results(*V);
    for (auto& c : V->counters) c = 0;
    V->credit.assign(n * 8, 0);
    fast_finish_classic(*V, C, p, WV, budget, 0);
    // ---- results and counters back to where the pairs came from
    const quicked_batch::HostResults& R = V->res[0];
    std::vector<size_t> extra(ni, 0);
    for (size_t q = 0; q < n; ++q) if (R.cigar_off[q] >= 0) extra[(size_t)src_item[q]] += strlen(R.cigar_pool.p + R.cigar_off[q]) + 1;
    for (size_t x = 0; x < ni; ++x) items[x].B->wr->cigar_pool.reserve(items[x].B->wr->cigar_pool.size + extra[x]);
    for (size_t q = 0; q < n; ++q) {
        quicked_batch& B = *items[(size_t)src_item[q]].B;
        const size_t pr = (size_t)src_pair[q];
        B.wr->score[pr] = R.score[q]; B.wr->status[pr] = R.status[q];
        if (R.cigar_off[q] >= 0) {
            const char* str = R.cigar_pool.p + R.cigar_off[q];
            const size_t len = strlen(str) + 1;
            memcpy(B.wr->cigar_pool.p + B.wr->cigar_pool.size, str, len);
            B.wr->cigar_off[pr] = (int64_t)B.wr->cigar_pool.size;
            B.wr->cigar_pool.size += len;
        } else B.wr->cigar_off[pr] = -1;
        for (int slot : {0, 1, 2, 3, 4}) B.counters[slot] += V->credit[q * 8 + (size_t)slot];
    }
    for (size_t x = 0; x < ni; ++x) { items[x].B->counters[6] = 0; items[x].B->counters[7] = 0; }     // quicked_classic ASSIGNS these two
    for (size_t q = 0; q < n; ++q) {
        quicked_batch& B = *items[(size_t)src_item[q]].B;
        B.counters[6] += V->credit[q * 8 + 6]; B.counters[7] += V->credit[q * 8 + 7];
    }
    V->credit.clear();
}

static bool same_flow(const PendingFetch& a, const PendingFetch& b) {
    const quicked_params_t &x = a.params, &y = b.params;
    return x.algo == y.algo && x.bandwidth == y.bandwidth && x.window_size == y.window_size && x.overlap_size == y.overlap_size &&
           x.hew_threshold[0] == y.hew_threshold[0] && x.hew_threshold[1] == y.hew_threshold[1] &&
           x.hew_percentage[0] == y.hew_percentage[0] && x.hew_percentage[1] == y.hew_percentage[1] &&
           x.only_score == y.only_score && x.force_scalar == y.force_scalar;
}

// -> every job this call has dealt with (the one handed in and those it took from the queue): the caller retires them
static std::atomic<int64_t> g_fin_stats[4];        // flows run, batches they finished, merged flows (>= 2 batches), batches in merged flows
static void finisher_work(const FinishJob& job, std::vector<FinishJob>& taken) {
    quicked_batch& B0 = *job.B;
    PendingFetch& F0 = *static_cast<PendingFetch*>(job.pf.get());
    // the run is over (the batch is alive: destroy waits for fin_jobs).  Polled with short sleeps: hipEventSynchronize spins,
    // and these threads wait for every queued QuickEd run of the process.  No context is held meanwhile.
    HIP_CHECK(hipSetDevice(B0.device));
    tl_bound_device = B0.device;
    for (;;) {
        const hipError_t e = hipEventQuery(B0.ev_done[F0.parity]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) throw HipError{e, "hipEventQuery(B.ev_done[F.parity])", __LINE__};
        if (g_fin_stop.load()) return;                               // the process is exiting
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    // other queued jobs whose runs are over too: the same flow serves their pairs (merged_finish)
    static const int merge_max = std::max(1, env_int("QE_FINISH_MERGE", 4));
    std::vector<FinishJob> group{job};
    {
        std::lock_guard<std::mutex> lk(g_fin_mu);
        for (auto it = g_fin_q.begin(); it != g_fin_q.end() && (int)group.size() < merge_max;) {
            const PendingFetch& F = *static_cast<const PendingFetch*>(it->pf.get());
            const bool fits = it->B->device == B0.device && it->B != &B0 && same_flow(F0, F) && it->B->cigar_style == B0.cigar_style &&
                              !it->B->check && !B0.check && hipEventQuery(it->B->ev_done[F.parity]) == hipSuccess;
            if (fits) { group.push_back(*it); taken.push_back(*it); it = g_fin_q.erase(it); }
            else ++it;
        }
    }
    (void)hipGetLastError();
    ApiScope scope;
    tl_device = B0.device;
    Context& C = ctx();
    // ---- per batch: its fin_mu (the first is waited for, the others only taken when free: a caller that is fetching one
    // right now does that one's work itself), still the run this job was made for, pairs left at all
    std::vector<MergeItem> items;
    for (size_t x = 0; x < group.size(); ++x) {
        quicked_batch& B = *group[x].B;
        PendingFetch& F = *static_cast<PendingFetch*>(group[x].pf.get());
        std::unique_lock<std::mutex> lk(B.fin_mu, std::defer_lock);
        if (x == 0) lk.lock(); else if (!lk.try_lock()) continue;
        if (B.pending_fetch.get() != group[x].pf.get()) { if (trace_on()) fprintf(stderr, "[qe] early finish: batch %p already fetched / superseded\n", (void*)&B); continue; }
        std::vector<int32_t> skip;
        d2h(skip, F.d_skip, F.L.pair.size(), C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        bool any = false;
        for (size_t t = 0; t < skip.size() && !any; ++t) any = skip[t] != 0 && F.L.pair[t] >= 0;
        if (!any) continue;                                              // nothing to finish: the caller's fetch is a copy
        MergeItem it;
        it.B = &B; it.pf = group[x].pf; it.lk = std::move(lk); it.keep = B.pending_fetch;
        items.push_back(std::move(it));
    }
    if (trace_on()) fprintf(stderr, "[qe t%03d @%.1f] early finish: %zu job(s), %zu with pairs left\n", (int)(syscall(SYS_gettid) % 1000), now_ms(), group.size(), items.size());
    if (items.empty()) return;
    // The results go to the batches' shadow sets (quicked_batch::res): a caller may be reading the visible ones.  This
    // thread's context is in the book like any other (fast_finish_classic plans its pools there).
    auto trim_own = [&]() { (void)C.release_pools(nullptr, true); };
    const double t_job = now_ms();
    for (MergeItem& it : items) it.B->wr = &it.B->res[1 - it.B->vis];
    try {
        if (items.size() == 1) {
            quicked_batch& B = *items[0].B;
            B.fin_status = fetch_pending(B);
            B.shadow_ready = true;
        } else {
            std::vector<MergeItem> left;                  // batches whose runs did leave pairs (the skip flags said so; the collect step agrees)
            for (MergeItem& it : items) {
                it.B->fin_status = fetch_pending(*it.B, &it.W);
                if (!it.W.Ls.pair.empty()) left.push_back(std::move(it)); else { it.B->shadow_ready = true; it.B->wr = &it.B->res[it.B->vis]; it.lk.unlock(); }
            }
            items.swap(left);
            // one flow for the batches that left FEW pairs -- there the flow's duration is launch latency, whatever the
            // number of pairs (12.5 k-pair batches with 1 % hard pairs: 0.54 -> 1.08 M alignments/s) -- a flow of its own
            // for a batch that left thousands (20 k indel-heavy pairs each: merged, three of them ran 5 x slower than apart)
            std::sort(items.begin(), items.end(), [](const MergeItem& a, const MergeItem& b) { return a.W.Ls.pair.size() < b.W.Ls.pair.size(); });
            static const size_t merge_pairs = (size_t)std::max(0, env_int("QE_FINISH_MERGE_PAIRS", 8192));
            size_t nm = 0, pairs = 0;
            while (nm < items.size() && pairs + items[nm].W.Ls.pair.size() <= merge_pairs) pairs += items[nm++].W.Ls.pair.size();
            if (nm < 2) nm = 0;
            if (nm >= 2) { merged_finish(items, nm, C); ++g_fin_stats[2]; g_fin_stats[3] += (int64_t)nm; }
            for (size_t x = nm; x < items.size(); ++x) {
                PendingFetch& F = *static_cast<PendingFetch*>(items[x].pf.get());
                fast_finish_classic(*items[x].B, C, F.params, items[x].W, F.matrix_budget, F.parity);
            }
            for (MergeItem& it : items) {
                fetch_finalize(*it.B, static_cast<PendingFetch*>(it.pf.get())->quicked);
                it.B->shadow_ready = true;
            }
        }
    }
    catch (const HipError&) {
        // e.g. out of memory next to the other threads' pools: nothing is lost -- the runs' results are still in the batches'
        // result arenas, and the callers' fetches do the same work in their own contexts
        (void)hipGetLastError();
        for (MergeItem& it : items) {
            if (!it.B || !it.lk.owns_lock() || it.B->shadow_ready) continue;
            quicked_batch& B = *it.B;
            B.wr = &B.res[B.vis];
            B.pending_fetch = it.keep; B.pending = true; B.fin_status = QUICKED_OK; B.shadow_ready = false;
        }
        trim_own();
        throw;
    }
    for (MergeItem& it : items) it.B->wr = &it.B->res[it.B->vis];
    if (!items.empty()) { ++g_fin_stats[0]; g_fin_stats[1] += (int64_t)items.size(); }
    if (trace_on()) fprintf(stderr, "[qe t%03d @%.1f] early finish: %zu batch(es) done in %.1f ms (context %p holds %.2f GB)\n", (int)(syscall(SYS_gettid) % 1000), now_ms(), items.size(), now_ms() - t_job, (void*)&C, C.held.load() / 1e9);
    // what this thread keeps between jobs is in the book like anybody's pools (and an allocation that finds the device full
    // takes it, qe_pool.h); it goes back on its own only where the device is short
    if (C.held.load() > ((size_t)8 << 30)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 5) trim_own();
    }
}

static int g_fin_limit = 3;                          // threads that may take work now (QE_FINISHERS at the last submit; under g_fin_mu)
static void finisher_main(int index) {
    for (;;) {
        FinishJob job;
        {
            std::unique_lock<std::mutex> lk(g_fin_mu);
            ++g_fin_idle;
            g_fin_cv.wait(lk, [index] { return !g_fin_q.empty() && index < g_fin_limit; });
            --g_fin_idle;
            job = std::move(g_fin_q.front());
            g_fin_q.pop_front();
            ++g_fin_busy;
        }
        std::vector<FinishJob> taken;                     // jobs the work took from the queue besides this one
        try { if (!g_fin_stop.load()) finisher_work(job, taken); }
        catch (const HipError& e) {
            fprintf(stderr, "[quicked_hip] early finish: HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
            (void)hipGetLastError();
        }
        catch (const std::exception& e) { fprintf(stderr, "[quicked_hip] early finish: %s\n", e.what()); }
        --g_fin_busy;
        if (g_fin_stop.load()) continue;             // exiting: the batch objects may be gone
        taken.push_back(job);
        for (const FinishJob& j : taken) {
            {
                std::lock_guard<std::mutex> lk(j.B->fin_mu);
                --j.B->fin_jobs;
            }
            j.B->fin_cv.notify_all();
        }
    }
}
// at process exit (this library's destructors run before the HIP runtime's, which it depends on): no new early-finish work,
// and a job in progress gets a few seconds to leave the runtime alone
__attribute__((destructor)) static void finisher_shutdown() {
    g_fin_stop.store(true);
    { std::lock_guard<std::mutex> lk(g_fin_mu); g_fin_q.clear(); }
    for (int i = 0; i < 5000 && g_fin_busy.load() > 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

// called by run_batch with B.fin_mu held
static void finisher_submit(quicked_batch& B, const std::shared_ptr<void>& pf) {
    const int max_threads = env_int("QE_FINISHERS", 3);           // read per call: tests switch it
    if (max_threads <= 0) return;
    ++B.fin_jobs;
    std::lock_guard<std::mutex> lk(g_fin_mu);
    g_fin_limit = max_threads;
    g_fin_q.push_back(FinishJob{&B, pf});
    if (g_fin_idle == 0 && g_fin_threads < max_threads) { std::thread(finisher_main, g_fin_threads).detach(); ++g_fin_threads; }
    g_fin_cv.notify_all();
}

}  // namespace qe

// ---------------------------------------------------------------------------
// Host -> HBM upload of a byte span.  Pinned (hipHostMalloc / registered) memory goes straight to the DMA
// engine; pageable memory is pipelined through pinned staging slots by a few worker threads, each with its
// own stream, so the copy into staging overlaps the DMA of the previous slot.
// ---------------------------------------------------------------------------
namespace qe {
static bool host_is_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

static void upload_span(uint8_t* dst, const uint8_t* src, size_t bytes, int device) {
    if (bytes == 0) return;
    // small spans (single quicked_align calls): one plain copy; the staging threads below cost ~12 ms to set up
    if (bytes < ((size_t)32 << 20) || host_is_pinned(src)) {
        HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return;
    }
    const size_t SLOT = (size_t)16 << 20;
    const int nthreads = (int)std::min<size_t>(8, std::max<size_t>(1, bytes / (4 * SLOT)));
    std::vector<std::thread> th;
    std::vector<int> err((size_t)nthreads, 0);
    for (int w = 0; w < nthreads; ++w) {
        th.emplace_back([=, &err]() {
            if (hipSetDevice(device) != hipSuccess) { err[(size_t)w] = 1; return; }
            const size_t lo = bytes * (size_t)w / (size_t)nthreads, hi = bytes * (size_t)(w + 1) / (size_t)nthreads;
            hipStream_t st; uint8_t* stage[2] = {nullptr, nullptr}; hipEvent_t ev[2];
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { err[(size_t)w] = 1; return; }
            for (int i = 0; i < 2; ++i) {
                if (hipHostMalloc((void**)&stage[i], SLOT, hipHostMallocDefault) != hipSuccess) err[(size_t)w] = 1;
                if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) err[(size_t)w] = 1;
            }
            int slot = 0;
            for (size_t o = lo; o < hi && !err[(size_t)w]; o += SLOT, slot ^= 1) {
                const size_t len = std::min(SLOT, hi - o);
                if (hipEventSynchronize(ev[slot]) != hipSuccess) err[(size_t)w] = 1;     // the slot's previous DMA is done
                memcpy(stage[slot], src + o, len);
                if (hipMemcpyAsync(dst + o, stage[slot], len, hipMemcpyHostToDevice, st) != hipSuccess) err[(size_t)w] = 1;
                if (hipEventRecord(ev[slot], st) != hipSuccess) err[(size_t)w] = 1;
            }
            (void)hipStreamSynchronize(st);
            for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); (void)hipEventDestroy(ev[i]); }
            (void)hipStreamDestroy(st);
        });
    }
    for (auto& t : th) t.join();
    for (int e : err) if (e) throw HipError{hipErrorUnknown, "upload_span", __LINE__};
}
}  // namespace qe

// ===========================================================================
// C-ABI
// ===========================================================================
QE_API void* quicked_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
QE_API void quicked_host_free(void* p) { if (p) (void)hipHostFree(p); }

QE_API int quicked_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return std::min(count, (int)QE_MAX_DEVICES);
}

QE_API quicked_status_t quicked_set_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= QE_MAX_DEVICES) return QUICKED_ERROR;
    tl_device = device;
    return QUICKED_OK;
}

// every call that works on a batch object holds its fin_mu: an early finish of the batch's last run (qe::finisher_work) is
// waited for, and cannot start in the middle of the call
static quicked_status_t guard(quicked_batch* B, quicked_status_t (*fn)(quicked_batch*, void*), void* arg) {
    std::unique_lock<std::mutex> lk;
    if (B) lk = std::unique_lock<std::mutex>(B->fin_mu);
    ApiScope scope;
    try { return fn(B, arg); }
    catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
        return QUICKED_ERROR;
    }
    catch (const std::bad_alloc&) { fprintf(stderr, "[quicked_hip] out of host memory\n"); return QUICKED_ERROR; }
}

// (re)loads a batch object with n pairs: host-side layout, arena (kept when it is large enough), H2D.  The caller has
// made sure no run of the batch is still on the device.
namespace qe {
static void batch_reset_state(quicked_batch* B) {
    B->pl_p_words = 0; B->pl_t_words = 0;
    for (bool& h : B->have_rev) h = false;
    for (bool& e : B->ev_done_set) e = false;
    B->parity = 0; B->pending = false; B->pending_fetch.reset(); B->d_score = nullptr;
    if (B->est_bound < 0) B->est_bound = 0;          // other pairs: QuickEd's sizing decision is taken again (a streamed batch keeps its estimate)
    B->res[0].clear(); B->res[1].clear(); B->vis = 0; B->wr = &B->res[0]; B->shadow_ready = false; B->last_parity = -1;
}
static void batch_arena(quicked_batch* B, size_t need) {
    if (B->arena && B->arena_bytes >= need) return;
    if (B->arena) { HIP_CHECK(hipDeviceSynchronize()); device_free(B->arena, B->device); B->arena = nullptr; B->arena_bytes = 0; }
    device_malloc((void**)&B->arena, need, B->device, nullptr, "hipMalloc(batch arena)", __LINE__);
    B->arena_bytes = need;
}
static void batch_load(quicked_batch* B, Context& C, int64_t n,
                       const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                       const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    batch_reset_state(B);
    B->n = n; B->device = C.device; B->packed = false; B->wire = 0;
    B->p_len.assign(pattern_len, pattern_len + n); B->t_len.assign(text_len, text_len + n);
    // A pool whose pairs lie (nearly) back to back is uploaded as the byte span it is, offsets kept;
    // a sparse one is compacted first.
    B->p_off.resize((size_t)n); B->t_off.resize((size_t)n); B->plp_off.resize((size_t)n); B->plt_off.resize((size_t)n);
    size_t pb = 0, tb = 0;
    int64_t p_lo = INT64_MAX, p_hi = 0, t_lo = INT64_MAX, t_hi = 0;
    for (int64_t i = 0; i < n; ++i) {
        pb += (size_t)pattern_len[i]; tb += (size_t)text_len[i];
        if (pattern_len[i]) { p_lo = std::min(p_lo, pattern_off[i]); p_hi = std::max(p_hi, pattern_off[i] + pattern_len[i]); }
        if (text_len[i]) { t_lo = std::min(t_lo, text_off[i]); t_hi = std::max(t_hi, text_off[i] + text_len[i]); }
        B->plp_off[i] = (int64_t)B->pl_p_words; B->pl_p_words += (size_t)3 * ((size_t)(pattern_len[i] + 63) / 64 + 2);
        B->plt_off[i] = (int64_t)B->pl_t_words; B->pl_t_words += (size_t)3 * ((size_t)(text_len[i] + 63) / 64 + 2);
    }
    if (p_lo == INT64_MAX) { p_lo = 0; p_hi = 0; }
    if (t_lo == INT64_MAX) { t_lo = 0; t_hi = 0; }
    const bool p_dense = (size_t)(p_hi - p_lo) <= pb + pb / 4 + ((size_t)1 << 20);
    const bool t_dense = (size_t)(t_hi - t_lo) <= tb + tb / 4 + ((size_t)1 << 20);
    {
        size_t po = 0, to = 0;
        for (int64_t i = 0; i < n; ++i) {
            B->p_off[i] = p_dense ? pattern_off[i] - p_lo : (int64_t)po; po += (size_t)pattern_len[i];
            B->t_off[i] = t_dense ? text_off[i] - t_lo : (int64_t)to; to += (size_t)text_len[i];
        }
    }
    const size_t p_bytes = p_dense ? (size_t)(p_hi - p_lo) : pb, t_bytes = t_dense ? (size_t)(t_hi - t_lo) : tb;
    B->order.resize((size_t)n);
    std::iota(B->order.begin(), B->order.end(), 0);
    bool ragged = false;
    for (int64_t i = 1; i < n && !ragged; ++i)
        ragged = std::max(B->p_len[i], B->t_len[i]) > std::max(B->p_len[i - 1], B->t_len[i - 1]);
    if (ragged)
        std::stable_sort(B->order.begin(), B->order.end(), [&](int a, int b) {
            const int la = std::max(B->p_len[a], B->t_len[a]), lb = std::max(B->p_len[b], B->t_len[b]);
            return la > lb;
        });
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    // plane sets: one per run of this batch that may be on the device at once (run_batch's rotation depth)
    B->np_alloc = rotation_depth(n);
    {   // five plane sets only while they are small change (100 k pairs of 10 kb: 1.5 GB each); a 400 k-pair batch keeps three
        const size_t set_bytes = 2 * (pad((B->pl_p_words + 8) * 8) + pad((B->pl_t_words + 8) * 8));
        if (B->np_alloc <= 5 && set_bytes * 5 > ((size_t)16 << 30)) B->np_alloc = 3;
    }
    const size_t need = pad(p_bytes + 64) + pad(t_bytes + 64) + 4 * pad((size_t)n * 8) + 2 * pad((size_t)n * 4) +
                        2 * (size_t)B->np_alloc * (pad((B->pl_p_words + 8) * 8) + pad((B->pl_t_words + 8) * 8)) + (size_t)B->np_alloc * pad((size_t)n * 4) + 4096;
    batch_arena(B, need);
    qe::ArenaCarver A{B->arena, 0};
    B->d_asc_p = A.take<uint8_t>(p_bytes + 64); B->d_asc_t = A.take<uint8_t>(t_bytes + 64);
    B->d_p_off = A.take<int64_t>((size_t)n); B->d_t_off = A.take<int64_t>((size_t)n);
    B->d_plp_off = A.take<int64_t>((size_t)n); B->d_plt_off = A.take<int64_t>((size_t)n);
    B->d_p_len = A.take<int32_t>((size_t)n); B->d_t_len = A.take<int32_t>((size_t)n);
    for (int q = 0; q < B->np_alloc; ++q) {
        B->d_pl_p[q] = A.take<u64>(B->pl_p_words + 8); B->d_pl_t[q] = A.take<u64>(B->pl_t_words + 8);
        B->d_pl_pr[q] = A.take<u64>(B->pl_p_words + 8); B->d_pl_tr[q] = A.take<u64>(B->pl_t_words + 8);
        B->d_flags[q] = A.take<u32>((size_t)n);
        if (!B->ev_done[q]) HIP_CHECK(hipEventCreateWithFlags(&B->ev_done[q], hipEventDisableTiming));
    }
    auto send = [&](uint8_t* dst, const char* pool, const int64_t* off, const int32_t* len, const std::vector<int64_t>& doff,
                    bool dense, int64_t lo, size_t bytes) {
        if (dense) { upload_span(dst, (const uint8_t*)pool + lo, bytes, C.device); return; }
        std::vector<uint8_t> h(bytes + 64, 0);
        for (int64_t i = 0; i < n; ++i) if (len[i]) memcpy(h.data() + doff[(size_t)i], pool + off[i], (size_t)len[i]);
        upload_span(dst, h.data(), bytes, C.device);
    };
    send(B->d_asc_p, pattern_pool, pattern_off, pattern_len, B->p_off, p_dense, p_lo, p_bytes);
    send(B->d_asc_t, text_pool, text_off, text_len, B->t_off, t_dense, t_lo, t_bytes);
    h2d(B->d_p_off, B->p_off, C.stream); h2d(B->d_t_off, B->t_off, C.stream);
    h2d(B->d_plp_off, B->plp_off, C.stream); h2d(B->d_plt_off, B->plt_off, C.stream);
    h2d(B->d_p_len, B->p_len, C.stream); h2d(B->d_t_len, B->t_len, C.stream);
    HIP_CHECK(hipStreamSynchronize(C.stream));
}

// every run of the batch that is still on the device (queued by any thread) is over
static void batch_quiesce(quicked_batch* B) {
    for (int q = 0; q < quicked_batch::NP; ++q)
        if (B->ev_done[q] && B->ev_done_set[q]) HIP_CHECK(hipEventSynchronize(B->ev_done[q]));
}
}  // namespace qe

static quicked_batch* guarded_new(const std::function<void(quicked_batch*)>& load) {
    quicked_batch* B = nullptr;
    ApiScope scope;
    try {
        B = new quicked_batch();
        load(B);
        return B;
    } catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
    } catch (const std::bad_alloc&) {
        fprintf(stderr, "[quicked_hip] out of host memory\n");
    }
    delete B;
    return nullptr;
}

QE_API quicked_batch_t* quicked_batch_create(int64_t n,
                                             const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                             const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    if (n < 0) return nullptr;
    return guarded_new([&](quicked_batch* B) {
        Context& C = ctx();
        batch_load(B, C, n, pattern_pool, pattern_off, pattern_len, text_pool, text_off, text_len);
    });
}

QE_API quicked_status_t quicked_batch_reload(quicked_batch_t* batch, int64_t n,
                                             const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                             const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    if (!batch || n < 0) return QUICKED_ERROR;
    struct Arg { int64_t n; const char* pp; const int64_t* po; const int32_t* pl; const char* tp; const int64_t* to; const int32_t* tl; }
        arg{n, pattern_pool, pattern_off, pattern_len, text_pool, text_off, text_len};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        batch_quiesce(B);
        batch_load(B, C, x->n, x->pp, x->po, x->pl, x->tp, x->to, x->tl);
        return QUICKED_OK;
    }, &arg);
}

// ---- packed wire format (supersedes sequence_buffer_t, sequence_buffer.h:30-50) ----------------
QE_API int64_t quicked_wire_words(int32_t len, int wire) {
    if (len < 0) return -1;
    if (wire == QUICKED_WIRE_2BIT) return ((int64_t)len + 31) / 32;
    if (wire == QUICKED_WIRE_PLANES3) return 3 * (((int64_t)len + 63) / 64);
    return -1;
}

// host-side serializer of one sequence (upper-case A, C, G, T; N only in PLANES3): the reference's code table
// (dna_text.c:41-46) restricted to the symbols whose raw-byte and encoded comparisons agree
QE_API quicked_status_t quicked_wire_pack(const char* seq, int32_t len, int wire, uint64_t* out) {
    const int64_t nwords = quicked_wire_words(len, wire);
    if (nwords < 0 || (len > 0 && (!seq || !out))) return QUICKED_ERROR;
    for (int64_t i = 0; i < nwords; ++i) out[i] = 0;
    for (int32_t i = 0; i < len; ++i) {
        int code;
        switch (seq[i]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                          case 'N': code = 4; break; default: return QUICKED_ERROR; }
        if (wire == QUICKED_WIRE_2BIT) {
            if (code == 4) return QUICKED_ERROR;
            out[i >> 5] |= (uint64_t)code << (2 * (i & 31));
        } else {
            uint64_t* row = out + 3 * (int64_t)(i >> 6);
            const uint64_t bit = (uint64_t)1 << (i & 63);
            if (code == 4) row[2] |= bit;
            else { if (code & 1) row[0] |= bit; if (code & 2) row[1] |= bit; }
        }
    }
    return QUICKED_OK;
}

namespace qe {
static void batch_load_packed(quicked_batch* B, Context& C, int64_t n, int wire,
                              const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                              const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    batch_reset_state(B);
    B->n = n; B->device = C.device; B->packed = true; B->wire = wire;
    B->p_len.assign(pattern_len, pattern_len + n); B->t_len.assign(text_len, text_len + n);
    B->p_off.assign((size_t)n, 0); B->t_off.assign((size_t)n, 0);
    B->plp_off.resize((size_t)n); B->plt_off.resize((size_t)n);
    // the wire pools are uploaded as the word spans they are
    int64_t p_lo = INT64_MAX, p_hi = 0, t_lo = INT64_MAX, t_hi = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t pw = quicked_wire_words(pattern_len[i], wire), tw = quicked_wire_words(text_len[i], wire);
        if (pw > 0) { p_lo = std::min(p_lo, pattern_word_off[i]); p_hi = std::max(p_hi, pattern_word_off[i] + pw); }
        if (tw > 0) { t_lo = std::min(t_lo, text_word_off[i]); t_hi = std::max(t_hi, text_word_off[i] + tw); }
        B->plp_off[i] = (int64_t)B->pl_p_words; B->pl_p_words += (size_t)3 * ((size_t)(pattern_len[i] + 63) / 64 + 2);
        B->plt_off[i] = (int64_t)B->pl_t_words; B->pl_t_words += (size_t)3 * ((size_t)(text_len[i] + 63) / 64 + 2);
    }
    if (p_lo == INT64_MAX) { p_lo = 0; p_hi = 0; }
    if (t_lo == INT64_MAX) { t_lo = 0; t_hi = 0; }
    const size_t pw_total = (size_t)(p_hi - p_lo), tw_total = (size_t)(t_hi - t_lo);
    std::vector<int64_t> pwo((size_t)n), two((size_t)n);
    for (int64_t i = 0; i < n; ++i) { pwo[(size_t)i] = pattern_word_off[i] - p_lo; two[(size_t)i] = text_word_off[i] - t_lo; }
    B->order.resize((size_t)n);
    std::iota(B->order.begin(), B->order.end(), 0);
    bool ragged = false;
    for (int64_t i = 1; i < n && !ragged; ++i)
        ragged = std::max(B->p_len[i], B->t_len[i]) > std::max(B->p_len[i - 1], B->t_len[i - 1]);
    if (ragged)
        std::stable_sort(B->order.begin(), B->order.end(), [&](int a, int b) {
            return std::max(B->p_len[a], B->t_len[a]) > std::max(B->p_len[b], B->t_len[b]);
        });
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    const size_t need = pad((pw_total + 8) * 8) + pad((tw_total + 8) * 8) + 6 * pad((size_t)n * 8) + 2 * pad((size_t)n * 4) +
                        2 * (pad((B->pl_p_words + 8) * 8) + pad((B->pl_t_words + 8) * 8)) + pad((size_t)n * 4) + 4096;
    batch_arena(B, need);
    qe::ArenaCarver A{B->arena, 0};
    u64* d_pw = A.take<u64>(pw_total + 8); u64* d_tw = A.take<u64>(tw_total + 8);
    int64_t* d_pwo = A.take<int64_t>((size_t)n); int64_t* d_two = A.take<int64_t>((size_t)n);
    B->d_p_off = A.take<int64_t>((size_t)n); B->d_t_off = A.take<int64_t>((size_t)n);
    B->d_plp_off = A.take<int64_t>((size_t)n); B->d_plt_off = A.take<int64_t>((size_t)n);
    B->d_p_len = A.take<int32_t>((size_t)n); B->d_t_len = A.take<int32_t>((size_t)n);
    B->d_pl_p[0] = A.take<u64>(B->pl_p_words + 8); B->d_pl_t[0] = A.take<u64>(B->pl_t_words + 8);
    B->d_pl_pr[0] = A.take<u64>(B->pl_p_words + 8); B->d_pl_tr[0] = A.take<u64>(B->pl_t_words + 8);
    B->d_flags[0] = A.take<u32>((size_t)n);
    B->d_asc_p = nullptr; B->d_asc_t = nullptr;
    B->np_alloc = quicked_batch::NP;
    for (int q = 0; q < quicked_batch::NP; ++q) {          // every set is the same resident planes
        B->d_pl_p[q] = B->d_pl_p[0]; B->d_pl_t[q] = B->d_pl_t[0]; B->d_pl_pr[q] = B->d_pl_pr[0]; B->d_pl_tr[q] = B->d_pl_tr[0];
        B->d_flags[q] = B->d_flags[0];
        if (!B->ev_done[q]) HIP_CHECK(hipEventCreateWithFlags(&B->ev_done[q], hipEventDisableTiming));
    }
    double tr_last = now_ms();
    QE_TRACE_POINT("load_packed: host layout");
    if (pw_total) upload_span((uint8_t*)d_pw, (const uint8_t*)(pattern_words + p_lo), pw_total * 8, C.device);
    if (tw_total) upload_span((uint8_t*)d_tw, (const uint8_t*)(text_words + t_lo), tw_total * 8, C.device);
    QE_TRACE_POINT("load_packed: word upload");
    h2d(d_pwo, pwo, C.stream); h2d(d_two, two, C.stream);
    h2d(B->d_p_off, B->p_off, C.stream); h2d(B->d_t_off, B->t_off, C.stream);
    h2d(B->d_plp_off, B->plp_off, C.stream); h2d(B->d_plt_off, B->plt_off, C.stream);
    h2d(B->d_p_len, B->p_len, C.stream); h2d(B->d_t_len, B->t_len, C.stream);
    HIP_CHECK(hipStreamSynchronize(C.stream));
    QE_TRACE_POINT("load_packed: arrays");
    B->d_wire_p = d_pw; B->d_wire_t = d_tw; B->d_wire_p_off = d_pwo; B->d_wire_t_off = d_two;
    B->unpack_pending = n > 0; B->unpack_event_set = false;
    if (!B->ev_unpacked) HIP_CHECK(hipEventCreateWithFlags(&B->ev_unpacked, hipEventDisableTiming));
    HIP_CHECK(hipStreamSynchronize(C.stream));
}
}  // namespace qe

QE_API quicked_batch_t* quicked_batch_create_packed(int64_t n, int wire,
                                                    const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                                    const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    if (n < 0 || (wire != QUICKED_WIRE_2BIT && wire != QUICKED_WIRE_PLANES3)) return nullptr;
    return guarded_new([&](quicked_batch* B) {
        Context& C = ctx();
        batch_load_packed(B, C, n, wire, pattern_words, pattern_word_off, pattern_len, text_words, text_word_off, text_len);
    });
}

QE_API quicked_status_t quicked_batch_reload_packed(quicked_batch_t* batch, int64_t n, int wire,
                                                    const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                                    const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    if (!batch || n < 0 || (wire != QUICKED_WIRE_2BIT && wire != QUICKED_WIRE_PLANES3)) return QUICKED_ERROR;
    struct Arg { int64_t n; int wire; const uint64_t* pw; const int64_t* po; const int32_t* pl; const uint64_t* tw; const int64_t* to; const int32_t* tl; }
        arg{n, wire, pattern_words, pattern_word_off, pattern_len, text_words, text_word_off, text_len};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        batch_quiesce(B);
        batch_load_packed(B, C, x->n, x->wire, x->pw, x->po, x->pl, x->tw, x->to, x->tl);
        return QUICKED_OK;
    }, &arg);
}

QE_API quicked_status_t quicked_batch_fetch(quicked_batch_t* batch) {
    if (!batch) return QUICKED_ERROR;
    return guard(batch, [](quicked_batch* B, void*) {
        if (!B->pending_fetch && B->shadow_ready) {
            // an early-finish thread has brought the run's results to the host already, into the set the getters do not
            // read: it becomes the visible one
            B->shadow_ready = false;
            B->vis ^= 1;
            B->wr = &B->res[B->vis];
            return B->fin_status < 0 ? B->fin_status : QUICKED_OK;
        }
        B->wr = &B->res[B->vis];
        return fetch_pending(*B);
    }, nullptr);
}

QE_API void quicked_batch_destroy(quicked_batch_t* batch) {
    if (!batch) return;
    ApiScope scope;
    try {
        tl_device = batch->device;
        (void)ctx();                           // binds the batch's device to this thread
        batch_quiesce(batch);                  // runs queued by any thread; hipFree then synchronises the device itself
    } catch (const HipError&) { (void)hipGetLastError(); }
    {   // early-finish jobs still queued for this batch find nothing to do and retire
        std::unique_lock<std::mutex> lk(batch->fin_mu);
        batch->pending_fetch.reset();
        batch->fin_cv.wait(lk, [&] { return batch->fin_jobs == 0; });
    }
    delete batch;
}

QE_API quicked_status_t quicked_batch_run(quicked_batch_t* batch, const quicked_params_t* params, int sync) {
    struct Arg { const quicked_params_t* p; int sync; } arg{params, sync};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        return run_batch(*B, *x->p, x->sync != 0);
    }, &arg);
}

QE_API quicked_status_t quicked_batch_sync(quicked_batch_t* batch) {
    return guard(batch, [](quicked_batch* B, void*) {
        tl_device = B->device;
        Context& C = ctx();
        C.sync_all();
        B->pending = false;
        return QUICKED_OK;
    }, nullptr);
}

QE_API quicked_status_t quicked_batch_kernel_times(quicked_batch_t* batch, double ms_sum[4], int64_t launches[4]) {
    struct Arg { double* ms; int64_t* n; } arg{ms_sum, launches};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        C.sync_all();
        for (int k = 0; k < 4; ++k) { x->ms[k] = 0; x->n[k] = 0; }
        for (size_t i = 0; i < C.kev_used; ++i) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, C.kev[i].first, C.kev[i].second));
            const int k = C.kev_kind[i] & 3;
            x->ms[k] += ms; ++x->n[k];
        }
        C.kev_used = 0;
        return QUICKED_OK;
    }, &arg);
}

QE_API quicked_status_t quicked_batch_kernel_time(quicked_batch_t* batch, double* ms_sum, int64_t* launches) {
    double ms[4]; int64_t n[4];
    const quicked_status_t st = quicked_batch_kernel_times(batch, ms, n);
    if (st < 0) return st;
    *ms_sum = ms[0] + ms[1]; *launches = n[0] + n[1];
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_scores(quicked_batch_t* batch, int32_t* scores_out, int32_t* status_out) {
    if (batch->res[batch->vis].score.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (scores_out) memcpy(scores_out, batch->res[batch->vis].score.data(), (size_t)batch->n * sizeof(int32_t));
    if (status_out) memcpy(status_out, batch->res[batch->vis].status.data(), (size_t)batch->n * sizeof(int32_t));
    return QUICKED_OK;
}

QE_API int64_t quicked_batch_cigar_bytes(quicked_batch_t* batch) { return (int64_t)batch->res[batch->vis].cigar_pool.size; }

QE_API quicked_status_t quicked_batch_cigar_view(quicked_batch_t* batch, const char** cigar_pool, const int64_t** cigar_off) {
    if (!batch || batch->res[batch->vis].cigar_off.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (cigar_pool) *cigar_pool = batch->res[batch->vis].cigar_pool.p;
    if (cigar_off) *cigar_off = batch->res[batch->vis].cigar_off.data();
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_cigars(quicked_batch_t* batch, char* cigar_pool, int64_t* cigar_off) {
    if (batch->res[batch->vis].cigar_off.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (cigar_pool && batch->res[batch->vis].cigar_pool.size) memcpy(cigar_pool, batch->res[batch->vis].cigar_pool.p, batch->res[batch->vis].cigar_pool.size);
    if (cigar_off) memcpy(cigar_off, batch->res[batch->vis].cigar_off.data(), (size_t)batch->n * sizeof(int64_t));
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_configure(quicked_batch_t* batch, int cigar_style, int check) {
    if (!batch || cigar_style < 0 || cigar_style > 2) return QUICKED_ERROR;
    if (check && batch->packed) return QUICKED_UNIMPLEMENTED;
    batch->cigar_style = cigar_style;
    batch->check = check != 0;
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_check_results(quicked_batch_t* batch, int32_t* ok_out) {
    if (!batch || batch->res[batch->vis].check_ok.size() != (size_t)batch->n) return QUICKED_ERROR;
    memcpy(ok_out, batch->res[batch->vis].check_ok.data(), (size_t)batch->n * sizeof(int32_t));
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_validate(quicked_batch_t* batch, const char* cigar_pool, int64_t pool_bytes,
                                               const int64_t* cigar_off, int32_t* ok_out) {
    struct Arg { const char* pool; int64_t bytes; const int64_t* off; int32_t* ok; } arg{cigar_pool, pool_bytes, cigar_off, ok_out};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        if (!x->off || !x->ok || (x->bytes > 0 && !x->pool)) return QUICKED_ERROR;
        if (B->packed) return QUICKED_UNIMPLEMENTED;          // the validator compares raw bytes; a packed batch has none
        tl_device = B->device;
        Context& C = ctx();
        C.sync_all();
        C.phase_u();
        const DevicePool::Mark mk = C.pool_w.mark();
        char* d_pool = C.pool_w.take<char>((size_t)x->bytes + 16);
        int64_t* d_off = C.pool_w.take<int64_t>((size_t)B->n + 1);
        int32_t* d_ok = C.pool_w.take<int32_t>((size_t)B->n + 1);
        if (x->bytes > 0) HIP_CHECK(hipMemcpyAsync(d_pool, x->pool, (size_t)x->bytes, hipMemcpyHostToDevice, C.stream));
        HIP_CHECK(hipMemsetAsync(d_pool + x->bytes, 0, 16, C.stream));      // a missing terminator cannot run off the pool
        HIP_CHECK(hipMemcpyAsync(d_off, x->off, (size_t)B->n * sizeof(int64_t), hipMemcpyHostToDevice, C.stream));
        for (int64_t i = 0; i < B->n; ++i) if (x->off[i] >= x->bytes) return QUICKED_ERROR;
        const int blocks = (int)((B->n + 63) / 64);
        hipLaunchKernelGGL(k_check_strings, dim3(blocks), dim3(64), 0, C.stream, pair_view(*B, false), (int)B->n,
                           (const char*)d_pool, (const int64_t*)d_off, d_ok);
        HIP_CHECK(hipMemcpyAsync(x->ok, d_ok, (size_t)B->n * sizeof(int32_t), hipMemcpyDeviceToHost, C.stream));
        HIP_CHECK(hipStreamSynchronize(C.stream));
        C.pool_w.release(mk);
        return QUICKED_OK;
    }, &arg);
}

QE_API quicked_status_t quicked_pool_stats(int64_t stats_out[8]) {
    for (int q = 0; q < 8; ++q) stats_out[q] = 0;
    for (const auto& bk : g_book) stats_out[1] += bk.oom_events.load();
    Context* C = tl_ctx;
    const int dev = C ? C->device : tl_device;
    if (dev >= 0 && dev < QE_MAX_DEVICES) stats_out[5] = (int64_t)g_book[dev].held.load();
    { std::lock_guard<std::mutex> lk(g_ctx_mu); for (const Context* c : g_ctx_all) { ++stats_out[6]; if (c->leased.load()) ++stats_out[7]; } }
    if (!C) return QUICKED_OK;
    stats_out[0] = (int64_t)C->held.load();
    stats_out[2] = C->last_na; stats_out[3] = C->last_sub_batches; stats_out[4] = (int64_t)C->pool_budget;
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_pool_trim(void) {
    ApiScope scope;
    try {
        Context& C = ctx();
        (void)C.release_pools(nullptr, true);
        { std::lock_guard<std::mutex> lk(g_ctx_mu); C.planned = 0; C.wanted = 0; }
        (void)release_unleased(C.device);                  // what threads that have ended left behind: pools ...
        retire_idle_streams(C.device);                      // ... and streams nobody is using
        return QUICKED_OK;
    } catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
        return QUICKED_ERROR;
    }
}

QE_API quicked_status_t quicked_early_finish_stats(int64_t stats_out[4]) {
    for (int q = 0; q < 4; ++q) stats_out[q] = qe::g_fin_stats[q].load();
    return QUICKED_OK;
}

QE_API int64_t quicked_batch_deferred_pairs(quicked_batch_t* batch) { return batch ? batch->res[batch->vis].deferred_pairs : -1; }

QE_API quicked_status_t quicked_batch_counters(quicked_batch_t* batch, int64_t counters_out[8]) {
    memcpy(counters_out, batch->res[batch->vis].counters, sizeof(batch->counters));
    return QUICKED_OK;
}

// ---- the six reference entry points ---------------------------------------
QE_API bool quicked_check_error(quicked_status_t status) { return status < 0; }    // quicked.c:380

QE_API const char* quicked_status_msg(quicked_status_t status) {                    // quicked.c:382-403
    switch (status) {
        case QUICKED_ERROR: return "ERROR: QuickEd has finished with unspecific error\n";
        case QUICKED_FAIL_NON_CONVERGENCE: return "ERROR: Hirschberg algorithm can not find a middle point of subsequence division!\n";
        case QUICKED_UNIMPLEMENTED: return "ERROR: The algorithm or parameter combination selected is not implemented\n";
        case QUICKED_UNKNOWN_ALGO: return "ERROR: Unknown algorithm selection\n";
        case QUICKED_EMPTY_SEQUENCE: return "ERROR: Tried to align an empty sequence\n";
        default: return "QuickEd finished without errors.\n";
    }
}

QE_API quicked_params_t quicked_default_params(void) {                             // quicked.c:308-321
    quicked_params_t p;
    memset(&p, 0, sizeof(p));
    p.algo = QUICKED;
    p.bandwidth = 15;
    p.window_size = 9;
    p.overlap_size = 1;
    p.hew_threshold[0] = p.hew_threshold[1] = 40;
    p.hew_percentage[0] = p.hew_percentage[1] = 15;
    return p;
}

// host timers the ABI exposes (profiler_timer.c:53-73, profiler_counter.c:46-66)
static void qe_timer_reset(profiler_timer_t* t) { memset(t, 0, sizeof(*t)); t->time_ns.min = UINT64_MAX; }
namespace qe {
static void qe_timer_start(profiler_timer_t* t) { if (!t) return; t->accumulated = 0; clock_gettime(CLOCK_REALTIME, &t->begin_timer); }
static void qe_timer_stop(profiler_timer_t* t) {
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
}  // namespace qe

// what the library hangs off aligner->mm_allocator when it owns it: the
// reference keeps its arena there (quicked.c:330-334); here it is the host
// block that owns the five timers and the last CIGAR strings.
struct AlignerState {
    mm_allocator_t shim;                  // first member: a valid mm_allocator_t* for callers that only pass it around
    profiler_timer_t timers[5];
    std::vector<char*> batch_cigars;
    std::vector<char> batch_pool;
    std::vector<char*> strings;           // every CIGAR quicked_align handed out: valid until quicked_free (quicked.c:48-50, 357-361)
    uint32_t magic;
};
static const uint32_t QE_MAGIC = 0x51CEDA11u;
// the same list for aligners that were given an external allocator (no AlignerState to hang it on)
static std::mutex g_strings_mu;
static std::unordered_map<const quicked_aligner_t*, std::vector<char*>> g_strings;
static AlignerState* own_state(const quicked_aligner_t* aligner) {
    if (aligner->mm_allocator == nullptr || aligner->params->external_allocator != nullptr) return nullptr;
    AlignerState* st = (AlignerState*)aligner->mm_allocator;
    return st->magic == QE_MAGIC ? st : nullptr;
}
static void keep_string(quicked_aligner_t* aligner, char* str) {
    if (AlignerState* st = own_state(aligner)) { st->strings.push_back(str); return; }
    std::lock_guard<std::mutex> lk(g_strings_mu);
    g_strings[aligner].push_back(str);
}
static void drop_strings(quicked_aligner_t* aligner) {
    bool listed = false;
    auto drop = [&](std::vector<char*>& v) { for (char* q : v) { listed |= q == aligner->cigar; free(q); } v.clear(); };
    if (AlignerState* st = own_state(aligner)) drop(st->strings);
    {
        std::lock_guard<std::mutex> lk(g_strings_mu);
        auto it = g_strings.find(aligner);
        if (it != g_strings.end()) { drop(it->second); g_strings.erase(it); }
    }
    if (aligner->cigar != nullptr && !listed) free(aligner->cigar);
    aligner->cigar = nullptr;
}

QE_API quicked_status_t quicked_new(quicked_aligner_t* aligner, quicked_params_t* params) {    // quicked.c:323-352
    aligner->params = params;
    aligner->score = -1;
    aligner->cigar = nullptr;
    AlignerState* st = nullptr;
    if (params->external_allocator == nullptr) {
        st = new AlignerState();
        memset(&st->shim, 0, sizeof(st->shim));
        st->magic = QE_MAGIC;
        aligner->mm_allocator = &st->shim;
    } else {
        aligner->mm_allocator = params->external_allocator;
    }
    if (params->external_timer) {
        // the caller patches the five pointers after quicked_new (benchmark_edit.c:61-65); NULL until then
        aligner->timer = aligner->timer_windowed_s = aligner->timer_windowed_l = aligner->timer_banded = aligner->timer_align = nullptr;
    } else {
        profiler_timer_t* tm = st ? st->timers : (profiler_timer_t*)calloc(5, sizeof(profiler_timer_t));
        for (int i = 0; i < 5; ++i) qe_timer_reset(&tm[i]);
        aligner->timer = &tm[0]; aligner->timer_windowed_s = &tm[1]; aligner->timer_windowed_l = &tm[2];
        aligner->timer_banded = &tm[3]; aligner->timer_align = &tm[4];
    }
    return QUICKED_WIP;
}

QE_API quicked_status_t quicked_free(quicked_aligner_t* aligner) {                             // quicked.c:354-378
    drop_strings(aligner);                 // every string quicked_align returned stays valid until here, as in the reference
    const bool own = aligner->mm_allocator != nullptr && aligner->params->external_allocator == nullptr;
    if (!aligner->params->external_timer && !own) free(aligner->timer);       // calloc'ed block of five
    if (own) {
        AlignerState* st = (AlignerState*)aligner->mm_allocator;
        if (st->magic == QE_MAGIC) delete st;
        aligner->mm_allocator = nullptr;
    }
    return QUICKED_WIP;
}

static quicked_status_t align_pairs(quicked_aligner_t* aligner, int n, const char* const* patterns, const int* plens,
                                    const char* const* texts, const int* tlens, int* scores_out, char** cigars_out,
                                    quicked_status_t* status_out, std::vector<char>* pool_keep) {
    std::vector<int64_t> po((size_t)n), to((size_t)n);
    std::vector<int32_t> pl((size_t)n), tl((size_t)n);
    size_t pb = 0, tb = 0;
    for (int i = 0; i < n; ++i) { po[i] = (int64_t)pb; pb += (size_t)plens[i]; to[i] = (int64_t)tb; tb += (size_t)tlens[i]; pl[i] = plens[i]; tl[i] = tlens[i]; }
    std::vector<char> pp(pb + 1), tp(tb + 1);
    for (int i = 0; i < n; ++i) {
        if (plens[i]) memcpy(pp.data() + po[i], patterns[i], (size_t)plens[i]);
        if (tlens[i]) memcpy(tp.data() + to[i], texts[i], (size_t)tlens[i]);
    }
    double tr_last = now_ms();
    ApiScope scope;
    // small calls (quicked_align, small quicked_align_batch) reuse one batch object per thread and device: no hipMalloc /
    // hipFree (a device-wide synchronisation) per call
    // (the object belongs to the thread's context: the next thread that takes the context over inherits it)
    const bool small = pb + tb <= ((size_t)8 << 20);
    quicked_batch_t* B = nullptr;
    if (small) {
        quicked_batch*& slot = *reinterpret_cast<quicked_batch**>(&ctx().small_batch);
        if (!slot) slot = quicked_batch_create(n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data());
        else if (quicked_batch_reload(slot, n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data()) < 0) {
            quicked_batch_destroy(slot); slot = nullptr;
        }
        B = slot;
    } else B = quicked_batch_create(n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data());
    if (!B) return QUICKED_ERROR;
    QE_TRACE_POINT("align_pairs: create");
    const quicked_params_t* p = aligner->params;
    // the five host timers are ticked around the stages they bracket in the reference
    // (quicked.c:76-78,184-193,204-235,240-275,283-294); a batch is one lap of each.
    tl_timers.windowed_s = aligner->timer_windowed_s; tl_timers.windowed_l = aligner->timer_windowed_l;
    tl_timers.banded = aligner->timer_banded; tl_timers.align = aligner->timer_align;
    qe_timer_start(aligner->timer);
    quicked_status_t st = quicked_batch_run(B, p, 1);
    qe_timer_stop(aligner->timer);
    QE_TRACE_POINT("align_pairs: run");
    tl_timers = HostTimers();
    quicked_status_t first_err = QUICKED_OK;
    bool any_err = false;
    for (int i = 0; i < n; ++i) {
        const quicked_status_t s = (st < 0 && st != QUICKED_EMPTY_SEQUENCE) ? st : (quicked_status_t)B->res[B->vis].status[(size_t)i];
        if (status_out) status_out[i] = s;
        if (s < 0 && !any_err) { any_err = true; first_err = s; }
        // a split that did not converge still has a score and a CIGAR in the reference (run_hirschberg extracts them from the
        // partial operations buffer before it returns the status, quicked.c:149-160): what the converged leaves gave
        if (scores_out && (s >= 0 || s == QUICKED_FAIL_NON_CONVERGENCE)) scores_out[i] = B->res[B->vis].score[(size_t)i];
    }
    if (cigars_out) {
        const quicked_batch::HostResults& R = B->res[B->vis];
        pool_keep->assign(R.cigar_pool.p, R.cigar_pool.p + R.cigar_pool.size);
        for (int i = 0; i < n; ++i)
            cigars_out[i] = (R.cigar_off[(size_t)i] >= 0 && !p->only_score) ? pool_keep->data() + R.cigar_off[(size_t)i] : nullptr;
    }
    if (!small) quicked_batch_destroy(B);
    QE_TRACE_POINT("align_pairs: destroy");
    if (any_err) return first_err;
    return st;
}

QE_API quicked_status_t quicked_align(quicked_aligner_t* aligner, const char* pattern, const int pattern_len,
                                      const char* text, const int text_len) {                  // quicked.c:405-437
    if (pattern_len == 0 || text_len == 0) return QUICKED_EMPTY_SEQUENCE;
    if ((unsigned)aligner->params->algo > (unsigned)HIRSCHBERG) return QUICKED_UNKNOWN_ALGO;
    int score = -1;
    char* cg = nullptr;
    std::vector<char> keep;
    quicked_status_t one = QUICKED_OK;
    const quicked_status_t st = align_pairs(aligner, 1, &pattern, &pattern_len, &text, &text_len, &score,
                                            aligner->params->only_score ? nullptr : &cg, &one, &keep);
    if (st < 0 && st != QUICKED_FAIL_NON_CONVERGENCE) return st;
    aligner->score = score;                      // also on QUICKED_FAIL_NON_CONVERGENCE, like extract_results (quicked.c:149-160)
    if (!aligner->params->only_score && cg) {
        // a previous align's string stays valid until quicked_free, as in the reference (arena allocation that the
        // next align does not release, quicked.c:48-50, 357-361)
        char* dup = strdup(cg);
        if (!dup) return QUICKED_ERROR;
        keep_string(aligner, dup);
        aligner->cigar = dup;
    }
    return st;
}

QE_API quicked_status_t quicked_align_batch(quicked_aligner_t* aligner, int n,
                                            const char* const* patterns, const int* pattern_lens,
                                            const char* const* texts, const int* text_lens,
                                            int* scores_out, char** cigars_out, quicked_status_t* status_out) {
    if (n <= 0) return QUICKED_OK;
    if ((unsigned)aligner->params->algo > (unsigned)HIRSCHBERG) {
        if (status_out) for (int i = 0; i < n; ++i) status_out[i] = QUICKED_UNKNOWN_ALGO;
        return QUICKED_UNKNOWN_ALGO;
    }
    const bool own = aligner->mm_allocator != nullptr && aligner->params->external_allocator == nullptr &&
                     ((AlignerState*)aligner->mm_allocator)->magic == QE_MAGIC;
    static thread_local std::vector<char> tl_keep;     // strings of the last batch when the aligner cannot own them
    std::vector<char>* keep = own ? &((AlignerState*)aligner->mm_allocator)->batch_pool : &tl_keep;
    return align_pairs(aligner, n, patterns, pattern_lens, texts, text_lens, scores_out, cigars_out, status_out, keep);
}
