// qe_driver.hip -- the device side of libquicked_hip.so (one translation unit with the kernels of qe_kernels.hip): launch
// helpers, the stage runners, the bound-and-align driver (run_quicked, quicked.c:163-306) as a staged batch pipeline, the
// early-finish threads, batch loading.  Device memory and contexts: qe_pool.h; the batch object: qe_batch.h; the C-ABI
// (include/quicked.h, include/quicked_batch.h): qe_capi.cpp.
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
#include "qe_batch.h"
// the kernels: this translation unit's device half.  (QE_KERNELS_HEADER: the sanitizer build of the host half on a machine
// without a GPU names tests/native/hip_stub/qe_kernels_stub.h here -- tests/test_host_sanitizers.py; never set in the product)
#ifndef QE_KERNELS_HEADER
#define QE_KERNELS_HEADER "qe_kernels.hip"
#endif
#include QE_KERNELS_HEADER


namespace qe {


// one in-stream copy as a kernel (see k_copy_multi); both buffers are padded to 16 bytes (pool / arena / stage allocations are)
// CopyBatch: the small uploads of one place in the code (the tables of a launch: 5 - 10 vectors) go in ONE launch -- every
// copy_kernel() on the batch's stream between its construction and its destruction is an entry of its table (a launch costs
// ~7 us of host time and ~5 us of the stream's whatever it moves: 15 of them per single-pair QuickEd call).  The scope ends
// before the first kernel that reads what was uploaded is queued.
struct CopyBatch;
static thread_local CopyBatch* tl_copy_batch = nullptr;
struct CopyBatch {
    CopyTable T; int64_t most = 1; hipStream_t s; CopyBatch* outer;
    explicit CopyBatch(hipStream_t stream) : s(stream), outer(tl_copy_batch) { T.n = 0; tl_copy_batch = this; }
    void flush() {
        if (T.n == 0) return;
        hipLaunchKernelGGL(k_copy_multi, dim3((unsigned)std::min<int64_t>(64, (most + 255) / 256), (unsigned)T.n), dim3(256), 0, s, T);
        T.n = 0; most = 1;
    }
    void add(void* dst, const void* src, size_t bytes) {
        T.dst[T.n] = (uint4*)dst; T.src[T.n] = (const uint4*)src; T.n_u4[T.n] = (int64_t)((bytes + 15) >> 4);
        most = std::max(most, T.n_u4[T.n]);
        if (++T.n == CopyTable::MAX) flush();
    }
    ~CopyBatch() { flush(); tl_copy_batch = outer; }
    CopyBatch(const CopyBatch&) = delete;
    CopyBatch& operator=(const CopyBatch&) = delete;
};
static void copy_kernel(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (tl_copy_batch && tl_copy_batch->s == s) { tl_copy_batch->add(dst, src, bytes); return; }
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
// FetchBatch: the small device arrays one place in the code reads back (two to five vectors and then a stream synchronisation)
// arrive in the context's pinned block through ONE copy launch; above 256 KB in total, plain asynchronous copies as before.
//   FetchBatch fb(C); fb.add(v1, d1, n); fb.add(v2, d2, n); fb.sync();      // = d2h x 2 + hipStreamSynchronize(C.stream)
struct FetchBatch {
    struct Item { void* dst; const void* src; size_t bytes, off; };
    Context& C; std::vector<Item> items; size_t top = 0;
    explicit FetchBatch(Context& c) : C(c) {}
    template <typename T> void add(std::vector<T>& dst, const T* src, size_t n) {
        dst.resize(n);
        if (n == 0) return;
        items.push_back(Item{dst.data(), src, n * sizeof(T), top});
        top += (n * sizeof(T) + 63) & ~(size_t)63;
    }
    void sync() {
        if (top <= ((size_t)256 << 10)) {
            uint8_t* st = top ? C.small_pinned(top + 256) : nullptr;
            {
                CopyBatch cb(C.stream);
                for (const Item& it : items) copy_kernel(st + it.off, it.src, it.bytes, C.stream);
            }
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(C.stream));
            for (const Item& it : items) memcpy(it.dst, st + it.off, it.bytes);
        } else {
            for (const Item& it : items) HIP_CHECK(hipMemcpyAsync(it.dst, it.src, it.bytes, hipMemcpyDeviceToHost, C.stream));
            HIP_CHECK(hipStreamSynchronize(C.stream));
        }
        items.clear(); top = 0;
    }
};
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

}  // namespace qe

#include "qe_stages.hip"

namespace qe {

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
            bool device_tried = false;
            while (!idx3.empty()) {
                if (!device_tried) {
                    // all doubling rounds in one launch where every task can have a wave (the pairs a run left: a few hundred),
                    // the convergence test on the device; what comes back flagged continues below from the cutoff it had reached
                    device_tried = true;
                    L3.pad();
                    std::vector<int32_t> dsc, dfl, dcut; std::vector<u32> dadv;
                    qe_timer_start(tl_timers.banded);
                    const bool ran = stage3_on_device(B, C, L3, dsc, dadv, dfl, dcut);
                    qe_timer_stop(tl_timers.banded);
                    if (ran) {
                        TaskList Ln; std::vector<size_t> idxn;
                        for (size_t k = 0; k < idx3.size(); ++k) {
                            const size_t t = idx3[k];
                            B.counters[0] += (int64_t)dadv[k];
                            B.note_pair(L3.pair[k], 0, (int64_t)dadv[k]);
                            if (dfl[k]) { Ln.push(L.pair[t], 0, L.m[t], 0, L.n[t], dcut[k], L.n[t]); idxn.push_back(t); }
                            else bound[t] = dsc[k];
                        }
                        L3 = Ln; idx3 = idxn;
                        continue;
                    }
                }
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
    // only_score, a run the caller waits for: the score is the end value of the fill's cells; nothing is stored, walked or
    // formatted (run_fill_score).  Not where a pair holds symbols whose raw bytes differ from their codes (lower case, IUPAC:
    // the reference's traceback compares the raw bytes, bpm_banded.c:1012, so its edit count is not the matrix's end
    // value there) -- the pack flags say so; they are this run's (stage 1 has been fetched: the pack is over).
    auto all_canonical = [&]() {
        std::vector<u32> fl;
        d2h(fl, (const u32*)B.d_flags[B.parity], (size_t)B.n, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        for (u32 f : fl) if (f & FLAG_NONCANON) return false;
        return true;
    };
    // Not where a pair's align step would split either (bpm_hirschberg.c:63-65): the levels' half passes are two launches of
    // half the chain each, side by side, and one pass over a 100 kb read's band is one wave's chain of 0.4 s -- 64 / 1 000 /
    // 10 000 pairs of 100 kb alone: 54 / 77 / 262 ms through the levels, 417 / 422 / 413-518 ms in one pass
    // (profiles/r06_y_probe_long_reads.txt)
    auto none_splits = [&]() {
        const uint64_t split = split_threshold();
        for (size_t t = 0; t < L.pair.size(); ++t)
            if (L.pair[t] >= 0 && (uint64_t)host_geometry(L.m[t], L.n[t], bound[t]).ebb * (uint64_t)L.n[t] * 16u > split) return false;
        return true;
    };
    TaskList LS;
    if (p.algo == QUICKED && p.only_score && whole_batch && fetch && quicked_score_pass_wanted() && none_splits()) {
        for (size_t t = 0; t < L.pair.size(); ++t)
            if (L.pair[t] >= 0) LS.push(L.pair[t], 0, L.m[t], 0, L.n[t], bound[t], L.n[t]);
        LS.pad();
    }
    if (!LS.pair.empty() && quicked_score_pass_fits(LS, fetch) && all_canonical()) {
        enter_a();
        qe_timer_start(tl_timers.align);
        StageResult RS;
        run_fill_score(B, C, LS, &RS, true, &B.d_score);
        qe_timer_stop(tl_timers.align);
        scatter_scores(B, LS, RS.score, QUICKED_WIP);
        B.counters[1] += (int64_t)sum_u32(RS.adv);
        QE_TRACE_POINT("score pass + fetch");
        return;
    }
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
                                std::vector<int32_t>& est, bool fetch) {
    if (p.algo != QUICKED || !quicked_fast_enabled(C) || B.est_bound <= 0) return false;   // the first run of a batch is a classic one
    if (p.only_score && fetch && env_int("QE_QUICKED_SCORE_PASS_FAST", 1) == 0 && quicked_score_pass_wanted()) return false;   // the classic flow ends in the score pass
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
    { FetchBatch fb(C); fb.add(cut, d_cut, nt); fb.add(skip, d_skip, nt); fb.add(steps, d_steps, nt); fb.sync(); }
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
    }
    if (F.fast) { stash_add(items, F.d_cut, nq * 4); stash_add(items, F.d_skip, nq * 4); stash_add(items, F.d_stage_steps, nq * 4); }
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
    const int64_t groups = std::max<int64_t>(1, (n + 63) / 64), slots = (int64_t)chip(tl_device).slots2();
    return (int)std::max<int64_t>(floor_sets, std::min<int64_t>(Context::NA, (slots + groups - 1) / groups));
}

static void finisher_submit(quicked_batch& B, const std::shared_ptr<void>& pf);

quicked_status_t run_batch(quicked_batch& B, const quicked_params_t& p, bool fetch) {
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
    // The depth of a batch's rotation is decided by its first queued run and does not grow afterwards: what this thread may
    // hold moves with what the other threads are doing at the instant of the plan, and a rotation that widened whenever a
    // neighbour paused allocated another 25 GB pool set in the middle of a stream (two threads of 150 k pairs: the four
    // timed runs took 1.9 instead of 0.18 s in three of eight processes, one thread ending up with five sets' pools)
    if (!fetch && !C.memory_tight) { if (B.na_cap > 0) na = std::min(na, B.na_cap); else B.na_cap = na; }
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
        if (B.ev_done_set[par]) HIP_CHECK(B.done_wait_on(C.sa(), par));
    }
    else {
        C.phase_w();
        C.pw().reset();
        if (C.decided_set[C.ai]) { HIP_CHECK(hipStreamWaitEvent(C.sw(), C.ev_decided[C.ai], 0)); C.decided_set[C.ai] = false; }
        if (B.ev_done_set[par]) HIP_CHECK(B.done_wait_on(C.sw(), par));
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
        B.ev_unpacked_tag = C.tag_a2[C.ai];
        B.unpack_pending = false; B.unpack_event_set = true;
    } else if (B.packed && B.unpack_event_set) {
        { std::shared_lock<std::shared_mutex> life(g_stream_life); if (!stream_gone(B.ev_unpacked_tag)) HIP_CHECK(hipStreamWaitEvent(C.stream, B.ev_unpacked, 0)); }
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
        if (quicked_fast_wanted(B, C, p, L, est, fetch)) {
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
            TaskList LA;
            for (size_t t = 0; t < nt; ++t) {
                if (L.pair[t] < 0) continue;
                LA.push(L.pair[t], 0, L.m[t], 0, L.n[t], est[t], L.n[t]);
            }
            // only_score: the fill's end value from a score-only pass instead of fill + traceback + edit count (run_fill_score)
            bool score_pass = p.only_score && quicked_score_pass_wanted();
            if (score_pass && (score_pass = quicked_score_pass_fits(LA, fetch))) LA.pad();      // (run_align takes its roots unpadded)
            sa.flags = score_pass ? B.d_flags[B.parity] : nullptr;
            hipLaunchKernelGGL(k_stage1_decide, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, C.stream, sa);
            HIP_CHECK(hipEventRecord(C.ev_decided[C.ai], C.stream));    // the stage's outputs live in the set's W pool, which the set's next run recycles
            C.decided_set[C.ai] = true;
            QE_TRACE_POINT("fast: decide queued");
            qe_timer_start(tl_timers.align);
            AlignStats AS;
            if (score_pass) {
                StageResult RS;
                run_fill_score(B, C, LA, &RS, fetch, &B.d_score, pf, d_cut, d_skip);
                if (fetch) { scatter_scores(B, LA, RS.score, QUICKED_WIP); AS.fill_adv = sum_u32(RS.adv); }      // (the tasks that left the list: -1, 0 -- the classic flow below)
            } else
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
        if (B.last_parity >= 0 && B.last_parity != par && B.ev_done_set[B.last_parity]) HIP_CHECK(B.done_wait_on(C.stream, B.last_parity));
        stash_results(B, C, *pf);
    }
    HIP_CHECK(hipEventRecord(C.ev1, C.stream));
    HIP_CHECK(hipEventRecord(C.ev_last, C.sa()));
    std::atomic_store(&C.ev_last_tag, C.tag_a2[C.ai]);
    HIP_CHECK(hipEventRecord(B.ev_done[par], C.sa()));
    std::atomic_store(&B.ev_done_tag[par], C.tag_a2[C.ai]);
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
    HIP_CHECK(B.done_sync(F.parity));
    C.phase_u();
    reset_host_results(B);
    for (int q = 0; q < 8; ++q) B.counters[q] = F.counters[q];
    if (F.kind == 1) {
        const size_t nt = F.task_pair.size();
        std::vector<int32_t> sc, ab; std::vector<u32> w;
        {
            FetchBatch fb(C);
            fb.add(sc, F.d_score, nt);
            if (F.d_adv) fb.add(w, F.d_adv, nt); else if (F.d_steps) fb.add(w, F.d_steps, nt);
            if (F.d_abort) fb.add(ab, F.d_abort, nt);
            fb.sync();
        }
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
    }
    if (F.fast && left) {
        if (fast_finish_collect(B, C, F.L, F.d_cut, F.d_skip, F.d_stage_steps, *left)) return QUICKED_OK;      // the caller goes on
    } else if (F.fast) quicked_fast_finish(B, C, F.params, F.L, F.d_cut, F.d_skip, F.d_stage_steps, F.matrix_budget, F.parity);
    fetch_finalize(B, F.quicked && (F.kind != 1 || F.fast));
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
// device / cigar_style / check: the batch's as they were when the run was queued (under its fin_mu).  A finisher looks at
// them BEFORE it takes the batch's fin_mu -- it waits for the run's event and picks the queued jobs that fit its flow first --
// while the batch's owner may be inside quicked_batch_reload / _configure, which write those fields: found by the
// ThreadSanitizer build under load (one run in ~100; tests/test_host_sanitizers.py)
struct FinishJob { quicked_batch* B; std::shared_ptr<void> pf; int device = 0, cigar_style = 0; bool check = false; };
static std::mutex& g_fin_mu = *new std::mutex;                      // never destroyed: detached threads wait on them at exit
static std::condition_variable& g_fin_cv = *new std::condition_variable;
static std::deque<FinishJob>& g_fin_q = *new std::deque<FinishJob>;
static int g_fin_threads = 0, g_fin_idle = 0;
static bool g_fin_retire = false;                    // quicked_pool_trim is joining the idle pool (under g_fin_mu)
static std::mutex& g_fin_pool_mu = *new std::mutex;  // the pool's threads: created, retired and joined under this
static std::vector<std::thread>& g_fin_pool = *new std::vector<std::thread>;
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
    {
        CopyBatch cb(C.stream);
        h2d(V->d_p_off, V->p_off, C.stream); h2d(V->d_t_off, V->t_off, C.stream);
        h2d(V->d_plp_off, V->plp_off, C.stream); h2d(V->d_plt_off, V->plt_off, C.stream);
        h2d(V->d_p_len, V->p_len, C.stream); h2d(V->d_t_len, V->t_len, C.stream);
        h2d(d_fl, flags, C.stream); h2d(d_gsrc, g_src, C.stream); h2d(d_gdst, g_dst, C.stream); h2d(d_gnw, g_nw, C.stream);
    }
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
    HIP_CHECK(hipSetDevice(job.device));
    tl_bound_device = job.device;
    for (;;) {
        const hipError_t e = B0.done_query(F0.parity);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) throw HipError{e, "hipEventQuery(B.ev_done[F.parity])", __LINE__};
        if (g_fin_stop.load()) return;                               // the process is exiting
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    // other queued jobs whose runs are over too: the same flow serves their pairs (merged_finish)
    const int merge_max = std::max(1, env_int("QE_FINISH_MERGE", 4));
    std::vector<FinishJob> group{job};
    {
        std::lock_guard<std::mutex> lk(g_fin_mu);
        for (auto it = g_fin_q.begin(); it != g_fin_q.end() && (int)group.size() < merge_max;) {
            const PendingFetch& F = *static_cast<const PendingFetch*>(it->pf.get());
            const bool fits = it->device == job.device && it->B != &B0 && same_flow(F0, F) && it->cigar_style == job.cigar_style &&
                              !it->check && !job.check && it->B->done_query(F.parity) == hipSuccess;
            if (fits) { group.push_back(*it); taken.push_back(*it); it = g_fin_q.erase(it); }
            else ++it;
        }
    }
    (void)hipGetLastError();
    // the job's own batch first, THEN a context -- the order every API call takes them in (guard(): fin_mu, then the context's
    // busy mutex).  A finisher that held a context while it waited here for a caller's long call on this batch kept its pools
    // out of that caller's reach (reclaim level 3 skips busy contexts) and counted as at work in the book
    std::unique_lock<std::mutex> lk0(B0.fin_mu);
    if (B0.pending_fetch.get() != job.pf.get() && group.size() == 1) {
        if (trace_on()) fprintf(stderr, "[qe] early finish: batch %p already fetched / superseded\n", (void*)&B0);
        return;
    }
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
        if (x == 0) lk = std::move(lk0); else if (!lk.try_lock()) continue;
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
    std::vector<MergeItem> left;                          // declared out here: a failure below still finds every lock and batch
    try {
        if (items.size() == 1) {
            quicked_batch& B = *items[0].B;
            B.fin_status = fetch_pending(B);
            B.shadow_ready = true;
        } else {
            // `left`: the batches whose runs did leave pairs (the skip flags said so; the collect step agrees)
            for (MergeItem& it : items) {
                it.B->fin_status = fetch_pending(*it.B, &it.W);
                if (!it.W.Ls.pair.empty()) left.push_back(std::move(it)); else { it.B->shadow_ready = true; it.B->wr = &it.B->res[it.B->vis]; it.lk.unlock(); }
            }
            items.swap(left);
            // one flow for the batches that left FEW pairs -- there the flow's duration is launch latency, whatever the
            // number of pairs (12.5 k-pair batches with 1 % hard pairs: 0.54 -> 1.08 M alignments/s) -- a flow of its own
            // for a batch that left thousands (20 k indel-heavy pairs each: merged, three of them ran 5 x slower than apart)
            std::sort(items.begin(), items.end(), [](const MergeItem& a, const MergeItem& b) { return a.W.Ls.pair.size() < b.W.Ls.pair.size(); });
            const size_t merge_pairs = (size_t)std::max(0, env_int("QE_FINISH_MERGE_PAIRS", 8192));
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
    catch (...) {
        // e.g. out of memory next to the other threads' pools (a HipError), or std::bad_alloc in one of the host vectors:
        // nothing is lost -- the runs' results are still in the batches' result arenas, and the callers' fetches do the same
        // work in their own contexts
        (void)hipGetLastError();
        for (std::vector<MergeItem>* set : {&items, &left}) for (MergeItem& it : *set) {      // (an entry moved from one to the other owns no lock)
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
static int g_fin_exited = 0, g_fin_exited_joined = 0;   // threads that have left finisher_loop / of those, joined by a retire (under g_fin_mu)
static std::condition_variable& g_fin_exit_cv = *new std::condition_variable;
static void finisher_loop(int index);
static void finisher_main(int index) {
    finisher_loop(index);
    std::lock_guard<std::mutex> lk(g_fin_mu);
    ++g_fin_exited;
    g_fin_exit_cv.notify_all();
}
static void finisher_loop(int index) {
    for (;;) {
        FinishJob job;
        {
            std::unique_lock<std::mutex> lk(g_fin_mu);
            ++g_fin_idle;
            g_fin_cv.wait(lk, [index] { return g_fin_stop.load() || g_fin_retire || (!g_fin_q.empty() && index < g_fin_limit); });
            --g_fin_idle;
            if (g_fin_stop.load() || g_fin_retire) return;         // the pool is being joined (process exit, quicked_pool_trim)
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
        if (g_fin_stop.load()) { --g_fin_busy; return; }      // exiting: the batch objects may be gone
        taken.push_back(job);
        for (const FinishJob& j : taken) {
            // notified under the lock: quicked_batch_destroy, which waits for fin_jobs == 0 under fin_mu, must not be able to
            // delete the batch between the decrement and the notify (its condition variable with it)
            std::lock_guard<std::mutex> lk(j.B->fin_mu);
            --j.B->fin_jobs;
            j.B->fin_cv.notify_all();
        }
        // busy until the bookkeeping above is done: it takes batches' fin_mu, and a caller that holds one of those may be
        // waiting for g_fin_pool_mu in finisher_submit -- finisher_retire, which joins this thread with that mutex held, must
        // therefore not start while this thread can still block on a fin_mu (it backs off while anything is busy)
        --g_fin_busy;
    }
}
// at process exit (this library's destructors run before the HIP runtime's, which it depends on): no new early-finish work,
// and a job in progress gets a few seconds to leave the runtime alone
// The early-finish threads are an OWNED pool (g_fin_pool: joinable, never detached).  At process exit (this library's
// destructors run before the HIP runtime's, which it depends on) no new work is taken, a job in progress notices the stop
// flag at its next poll, and every thread is joined.
__attribute__((destructor)) static void finisher_shutdown() {
    g_fin_stop.store(true);
    // the pool is taken out under its mutex and joined without it (a caller blocked in finisher_submit behind a join would
    // keep a batch's fin_mu from a thread that is being joined); a thread that does not come back within 5 s -- stuck in a
    // runtime that is itself shutting down -- is detached rather than waited for
    std::vector<std::thread> pool;
    { std::lock_guard<std::mutex> pl(g_fin_pool_mu); pool.swap(g_fin_pool); }
    { std::lock_guard<std::mutex> lk(g_fin_mu); g_fin_q.clear(); }
    g_fin_cv.notify_all();
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(5);
    {
        std::unique_lock<std::mutex> lk(g_fin_mu);
        g_fin_exit_cv.wait_until(lk, t_end, [&] { return g_fin_exited >= (int)pool.size() + g_fin_exited_joined; });
    }
    const bool all_out = [&] { std::lock_guard<std::mutex> lk(g_fin_mu); return g_fin_exited >= (int)pool.size() + g_fin_exited_joined; }();
    for (std::thread& t : pool) { if (!t.joinable()) continue; if (all_out) t.join(); else t.detach(); }
}
// quicked_pool_trim(): a process that is done with its batches gives the threads back too -- only while none of them has
// work (they start again on demand)
void finisher_retire() {
    std::lock_guard<std::mutex> pl(g_fin_pool_mu);
    {
        std::lock_guard<std::mutex> lk(g_fin_mu);
        if (g_fin_pool.empty() || !g_fin_q.empty() || g_fin_busy.load() > 0) return;
        g_fin_retire = true;
    }
    g_fin_cv.notify_all();
    // every thread is idle or on its way out (nothing queued, nothing busy -- and busy covers a thread's fin_mu bookkeeping):
    // none of them can be waiting for a mutex a caller behind g_fin_pool_mu holds, so joining here cannot cycle
    const int joined = (int)g_fin_pool.size();
    for (std::thread& t : g_fin_pool) if (t.joinable()) t.join();
    g_fin_pool.clear();
    std::lock_guard<std::mutex> lk(g_fin_mu);
    g_fin_retire = false; g_fin_threads = 0; g_fin_idle = 0; g_fin_exited_joined += joined;
}

// called by run_batch with B.fin_mu held
static void finisher_submit(quicked_batch& B, const std::shared_ptr<void>& pf) {
    const int max_threads = env_int("QE_FINISHERS", 3);           // read per call: tests switch it
    if (max_threads <= 0) return;
    if (g_fin_stop.load()) return;
    ++B.fin_jobs;
    std::lock_guard<std::mutex> pl(g_fin_pool_mu);                // (order: a batch's fin_mu, the pool, the queue)
    std::lock_guard<std::mutex> lk(g_fin_mu);
    g_fin_limit = max_threads;
    g_fin_q.push_back(FinishJob{&B, pf, B.device, B.cigar_style, B.check});
    if (g_fin_idle == 0 && g_fin_threads < max_threads) { g_fin_pool.emplace_back(finisher_main, g_fin_threads); ++g_fin_threads; }
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

// (re)loads a batch object with n pairs: host-side layout, arena (kept when it is large enough), H2D.  The caller has
// made sure no run of the batch is still on the device.
namespace qe {
static void batch_reset_state(quicked_batch* B) {
    B->pl_p_words = 0; B->pl_t_words = 0;
    for (bool& h : B->have_rev) h = false;
    for (bool& e : B->ev_done_set) e = false;
    B->parity = 0; B->pending = false; B->pending_fetch.reset(); B->d_score = nullptr; B->na_cap = 0;
    if (B->est_bound < 0) B->est_bound = 0;          // other pairs: QuickEd's sizing decision is taken again (a streamed batch keeps its estimate)
    B->res[0].clear(); B->res[1].clear(); B->vis = 0; B->wr = &B->res[0]; B->shadow_ready = false; B->last_parity = -1;
}
static void batch_arena(quicked_batch* B, size_t need) {
    if (B->arena && B->arena_bytes >= need) return;
    if (B->arena) { HIP_CHECK(hipDeviceSynchronize()); device_free(B->arena, B->device); B->arena = nullptr; B->arena_bytes = 0; }
    device_malloc((void**)&B->arena, need, B->device, nullptr, "hipMalloc(batch arena)", __LINE__);
    B->arena_bytes = need;
}
void batch_load(quicked_batch* B, Context& C, int64_t n,
                       const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                       const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    batch_reset_state(B);
    B->n = n; B->packed = false; B->wire = 0;
    if (B->device != C.device) B->device = C.device;      // (a batch stays on its device: no write for a finisher's unlocked reads to race with)
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
    // a small batch (single quicked_align calls): strings and tables through the context's pinned block, one copy launch
    const size_t small_total = p_bytes + t_bytes + (size_t)n * 40 + 8 * 64;
    if (small_total <= ((size_t)512 << 10)) {
        uint8_t* st = C.small_pinned(small_total + 256);
        size_t top = 0;
        CopyBatch cb(C.stream);
        auto put = [&](void* dst, const void* src, size_t bytes) {
            if (bytes == 0) return;
            memcpy(st + top, src, bytes);
            copy_kernel(dst, st + top, bytes, C.stream);
            top += (bytes + 63) & ~(size_t)63;
        };
        auto put_pool = [&](uint8_t* dst, const char* pool, const int64_t* off, const int32_t* len, const std::vector<int64_t>& doff,
                            bool dense, int64_t lo, size_t bytes) {
            if (bytes == 0) return;
            if (dense) { put(dst, pool + lo, bytes); return; }
            uint8_t* h = st + top;
            memset(h, 0, bytes);
            for (int64_t i = 0; i < n; ++i) if (len[i]) memcpy(h + doff[(size_t)i], pool + off[i], (size_t)len[i]);
            copy_kernel(dst, h, bytes, C.stream);
            top += (bytes + 63) & ~(size_t)63;
        };
        put_pool(B->d_asc_p, pattern_pool, pattern_off, pattern_len, B->p_off, p_dense, p_lo, p_bytes);
        put_pool(B->d_asc_t, text_pool, text_off, text_len, B->t_off, t_dense, t_lo, t_bytes);
        put(B->d_p_off, B->p_off.data(), (size_t)n * 8); put(B->d_t_off, B->t_off.data(), (size_t)n * 8);
        put(B->d_plp_off, B->plp_off.data(), (size_t)n * 8); put(B->d_plt_off, B->plt_off.data(), (size_t)n * 8);
        put(B->d_p_len, B->p_len.data(), (size_t)n * 4); put(B->d_t_len, B->t_len.data(), (size_t)n * 4);
        cb.flush();
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(C.stream));
        return;
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
    {
        CopyBatch cb(C.stream);
        h2d(B->d_p_off, B->p_off, C.stream); h2d(B->d_t_off, B->t_off, C.stream);
        h2d(B->d_plp_off, B->plp_off, C.stream); h2d(B->d_plt_off, B->plt_off, C.stream);
        h2d(B->d_p_len, B->p_len, C.stream); h2d(B->d_t_len, B->t_len, C.stream);
    }
    HIP_CHECK(hipStreamSynchronize(C.stream));
}

// every run of the batch that is still on the device (queued by any thread) is over
void batch_quiesce(quicked_batch* B) {
    for (int q = 0; q < quicked_batch::NP; ++q)
        if (B->ev_done[q] && B->ev_done_set[q]) HIP_CHECK(B->done_sync(q));
}
}  // namespace qe

namespace qe {
void batch_load_packed(quicked_batch* B, Context& C, int64_t n, int wire,
                              const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                              const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    batch_reset_state(B);
    B->n = n; B->packed = true; B->wire = wire;
    if (B->device != C.device) B->device = C.device;
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
    {
        CopyBatch cb(C.stream);
        h2d(d_pwo, pwo, C.stream); h2d(d_two, two, C.stream);
        h2d(B->d_p_off, B->p_off, C.stream); h2d(B->d_t_off, B->t_off, C.stream);
        h2d(B->d_plp_off, B->plp_off, C.stream); h2d(B->d_plt_off, B->plt_off, C.stream);
        h2d(B->d_p_len, B->p_len, C.stream); h2d(B->d_t_len, B->t_len, C.stream);
    }
    HIP_CHECK(hipStreamSynchronize(C.stream));
    QE_TRACE_POINT("load_packed: arrays");
    B->d_wire_p = d_pw; B->d_wire_t = d_tw; B->d_wire_p_off = d_pwo; B->d_wire_t_off = d_two;
    B->unpack_pending = n > 0; B->unpack_event_set = false;
    if (!B->ev_unpacked) HIP_CHECK(hipEventCreateWithFlags(&B->ev_unpacked, hipEventDisableTiming));
    HIP_CHECK(hipStreamSynchronize(C.stream));
}
}  // namespace qe

namespace qe {
quicked_status_t fetch_results(quicked_batch& B) { return fetch_pending(B); }
void early_finish_stats(int64_t stats_out[4]) { for (int q = 0; q < 4; ++q) stats_out[q] = g_fin_stats[q].load(); }

quicked_status_t batch_validate(quicked_batch* B, Context& C, const char* cigar_pool, int64_t pool_bytes, const int64_t* cigar_off, int32_t* ok_out) {
    C.sync_all();
    C.phase_u();
    const DevicePool::Mark mk = C.pool_w.mark();
    char* d_pool = C.pool_w.take<char>((size_t)pool_bytes + 16);
    int64_t* d_off = C.pool_w.take<int64_t>((size_t)B->n + 1);
    int32_t* d_ok = C.pool_w.take<int32_t>((size_t)B->n + 1);
    if (pool_bytes > 0) HIP_CHECK(hipMemcpyAsync(d_pool, cigar_pool, (size_t)pool_bytes, hipMemcpyHostToDevice, C.stream));
    HIP_CHECK(hipMemsetAsync(d_pool + pool_bytes, 0, 16, C.stream));      // a missing terminator cannot run off the pool
    HIP_CHECK(hipMemcpyAsync(d_off, cigar_off, (size_t)B->n * sizeof(int64_t), hipMemcpyHostToDevice, C.stream));
    for (int64_t i = 0; i < B->n; ++i) if (cigar_off[i] >= pool_bytes) { C.pool_w.release(mk); return QUICKED_ERROR; }
    const int blocks = (int)((B->n + 63) / 64);
    hipLaunchKernelGGL(k_check_strings, dim3(blocks), dim3(64), 0, C.stream, pair_view(*B, false), (int)B->n,
                       (const char*)d_pool, (const int64_t*)d_off, d_ok);
    HIP_CHECK(hipMemcpyAsync(ok_out, d_ok, (size_t)B->n * sizeof(int32_t), hipMemcpyDeviceToHost, C.stream));
    HIP_CHECK(hipStreamSynchronize(C.stream));
    C.pool_w.release(mk);
    return QUICKED_OK;
}
}  // namespace qe
