// qe_stages.hip -- included by qe_driver.hip (same translation unit as the kernels): how a stage gets onto the device.
// Launch shape of the lane-per-alignment kernels, task lists and per-group workspace layouts, and the stage runners --
// BandEd score-only in its three forms (one lane / G lanes / one wave per alignment), WindowEd, the CIGAR formatter, and
// the align step (bpm_compute_matrix_hirschberg, bpm_hirschberg.c:33-270) as a level-by-level work list.  Every runner
// returns with its kernels enqueued on C.stream.
#pragma once

namespace qe {

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
// The cooperative forms are launches of few waves, each a serial chain; next to a chip-filling launch of another run a lone
// wave gets a third of its SIMD's issue slots.  s_setprio 3 in those kernels (QE_WAVE_PRIO=0: off) puts them first in line.
static int wave_prio() { return env_int("QE_WAVE_PRIO", 1); }
template <class Args> static Args with_prio(Args a) { a.prio = wave_prio(); return a; }

template <typename Kernel, typename Args>
static void launch_groups(Context& C, Kernel kernel, const Args& args, size_t ngroups, int max_waves, size_t lds_per_wave, bool chain = false,
                          size_t pin_override = 0) {
    if (ngroups == 0) return;
    const int wpb = std::min(4, max_waves);
    const unsigned blocks = (unsigned)((ngroups + wpb - 1) / wpb);
    // 54 KB: 3 x 54 KB > 160 KB >= 2 x 54 KB, two workgroups per CU.  When the launches in flight have fewer workgroups than
    // the chip has CUs, 84 KB (one per CU): the dispatcher packs the workgroups of CONCURRENT small kernels two to a CU
    // while other CUs idle (three 49-workgroup launches in flight: 16.5 ms each at 54 KB, 11.7 ms at 84 KB, 11.4 ms alone)
    const Chip& chp = chip(C.device);
    size_t pin = ((size_t)blocks * (size_t)std::max(1, C.in_flight) > (size_t)chp.cus) ? (size_t)54 * 1024 : (size_t)84 * 1024;
    // chain: a launch of few waves whose duration is one wave's serial chain (WindowEd on a few thousand long reads: 157
    // waves of 1563 windows each).  108 KB: no 54 KB workgroup fits beside it, so its waves have their SIMDs to themselves
    // instead of sharing them with the fill of the run before (config 4: the stage took 59 ms beside that fill, 36 alone)
    if (chain && (size_t)blocks * (size_t)std::max(1, C.in_flight) <= (size_t)chp.cus / 2) pin = (size_t)108 * 1024;
    if (pin_override) pin = pin_override;
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
    const size_t ntp = (nt + 63) & ~(size_t)63;
    o.adv = C.scratch_p->take<u32>(2 * ntp); o.steps = o.adv + ntp;         // one block: one memset
    o.str_off = C.scratch_p->take<int64_t>(nt + 1);
    // the work counters are summed over every slot of the list: padding slots (and tasks a kernel skips) count 0
    HIP_CHECK(hipMemsetAsync(o.adv, 0, 2 * ntp * sizeof(u32), C.stream));
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

// d_cut / d_skip: the cutoffs are still being computed on the device (QuickEd's fast flow): the list's are the estimates the
// buffers are sized for; k_apply_cutoffs puts the real ones in and takes the tasks out that the host finishes afterwards
static ScoreLaunch launch_banded_score(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int timed, bool fill_geom = false,
                                       const int32_t* d_cut = nullptr, const int32_t* d_skip = nullptr) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    BandLayout lay = band_layout(L, fill_geom, false);
    lay.mat_u4 = 0;                                   // (the fill's geometry, not its checkpoints)
    S.T = upload_tasks(L, C);
    S.D = upload_layout(lay, C);
    S.O = take_out(C, S.nt);
    if (d_cut) {
        HIP_CHECK(hipMemsetAsync(S.O.score, 0xFF, S.nt * sizeof(int32_t), C.stream));      // a task taken out of the list has no score (-1)
        hipLaunchKernelGGL(k_apply_cutoffs, dim3((unsigned)((S.nt + 255) / 256)), dim3(256), 0, C.stream, (int)S.nt, S.T.cutoff, S.T.pair, d_cut, d_skip);
    }
    BandedArgs a;
    a.P = pair_view(B, reversed); a.T = S.T.v;
    a.ws = S.D.ws; a.g_ws_off = S.D.ws_off; a.g_nslots = S.D.nslots; a.g_nrows = S.D.nrows; a.g_nch = S.D.nch;
    a.mat = nullptr; a.g_mat_off = S.D.mat_off;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len;
    a.only_if = nullptr;
    a.lane_rel = env_int("QE_LANE_REL", 1);
    a.fill_geom = fill_geom ? 1 : 0;
    auto* ke = timed ? C.kernel_events(timed - 1) : nullptr;       // timed = kind + 1 (Context::kernel_events)
    if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
    // QE_SCORE_WAVES = 3: a 52 KB pin, three workgroups per CU = three waves per SIMD (the kernel's 158 VGPRs allow it)
    launch_groups(C, k_banded<false>, a, L.ngroups(), 8, 0, false, env_int("QE_SCORE_WAVES", 2) == 3 ? (size_t)52 * 1024 : 0);
    if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
    return S;
}

// lanes per alignment for the cooperative score-only kernel: enough waves to fill the chip
// (>= ~4 per SIMD) while every lane keeps >= 2 band slots; QE_COOP_G overrides (0 / 1 = off)
static int coop_lanes(const TaskList& L, int in_flight = 1, bool fill = false) {
    const char* const e_name = fill ? "QE_COOP_FILL_G" : "QE_COOP_G";
    const bool e = env_set(e_name);
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
    const Chip& chp = chip(tl_device);
    const size_t target = std::min<size_t>(2 * chp.slots2(), chp.frac2(0.34) * (size_t)std::max(1, n_max / 10000));      // ~700 waves of 2 048 slots per 10 kb of read
    if (e) G = env_int(e_name, 1);
    else
        while (G < 64 && ((live * G) / 64) * (size_t)std::max(1, in_flight) < target) G *= 2;      // runs in flight fill the chip together
    // the band-height test first + 2 < last must stay decidable G-2 chunks early: keep the band >= 3 G + 4 slots
    // the band-height test first + 2 < last must stay decidable G - 2 chunks early: a band of >= 3 G + 4 slots always is;
    // with 2 G + 4 a task whose band comes within G slots of its minimum height is flagged and recomputed by the one-lane
    // kernel -- rare, and worth it where the launch is short of waves anyway (4 000 pairs of 10 kb: 4.2 -> 2.8 ms with G = 8)
    while (G > 1 && min_nsl < 3 * G + 4) G /= 2;
    if (!e)
        while (G < 64 && min_nsl >= 2 * (2 * G) + 4 && ((live * G) / 64) * (size_t)std::max(1, in_flight) < chp.slots2() / 4) G *= 2;
    return G < 2 ? 1 : G;
}

// k_banded_coop over the list, then k_banded<false> over the tasks it flagged
// the band state of a wave's 64 / G tasks in LDS (k_banded_coop_lds): bytes per wave for bands of ns slots
static size_t coop_lds_bytes(int ns, int G) {
    const int NA = 64 / G, rr = ns + G + 4, cr = std::max(16, 4 * G);
    const size_t bytes = (size_t)2 * (ns + 1) * NA * 8 + (size_t)2 * rr * NA * 4 + (size_t)2 * cr * NA * 2 + (size_t)2 * NA * 4;
    return (bytes + 63) & ~(size_t)63;
}
// fill_geom: score-only over the FILL's cells (BandedArgs::fill_geom) -- the LDS form only; d_cut / d_skip as launch_banded_score
static ScoreLaunch launch_banded_coop(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int G, int timed, bool fill_geom = false,
                                      const int32_t* d_cut = nullptr, const int32_t* d_skip = nullptr) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    const int NA = 64 / G;
    const size_t nwaves = S.nt / NA;
    std::vector<int64_t> w_off(nwaves);
    std::vector<int32_t> w_ns(nwaves), w_nr(nwaves), w_nch(nwaves);
    size_t ws_bytes = 0;
    int ns_max = 3;
    for (size_t w = 0; w < nwaves; ++w) {
        int ns = 3, nr = 4, nch = 2;
        for (int q = 0; q < NA; ++q) {
            const size_t t = w * NA + q;
            if (L.pair[t] < 0) continue;
            const HGeom Gm = host_geometry(L.m[t], L.n[t], L.cutoff[t]);
            const int nsl = fill_geom ? Gm.ebb : Gm.ebb_local;
            ns = std::max(ns, nsl);
            nr = std::max(nr, (L.m[t] + 63) / 64 + nsl + 4);
            nch = std::max(nch, L.n[t] / 64 + 3);
        }
        w_ns[w] = ns; w_nr[w] = nr; w_nch[w] = nch;
        ns_max = std::max(ns_max, ns);
        w_off[w] = (int64_t)ws_bytes;
        const size_t bytes = (size_t)2 * (ns + 1) * NA * 8 + (size_t)2 * nr * NA * 4 + (size_t)2 * nch * NA * 2 + (size_t)2 * NA * 4;
        ws_bytes += (bytes + 255) & ~(size_t)255;
    }
    if (fill_geom && (env_int("QE_COOP_LDS", 1) == 0 || coop_lds_bytes(ns_max, G) > (size_t)38 * 1024))
        return launch_banded_score(B, C, L, reversed, timed, true, d_cut, d_skip);      // no room on chip: one lane per task
    S.T = upload_tasks(L, C);
    S.O = take_out(C, S.nt);
    if (d_cut) {
        HIP_CHECK(hipMemsetAsync(S.O.score, 0xFF, S.nt * sizeof(int32_t), C.stream));
        hipLaunchKernelGGL(k_apply_cutoffs, dim3((unsigned)((S.nt + 255) / 256)), dim3(256), 0, C.stream, (int)S.nt, S.T.cutoff, S.T.pair, d_cut, d_skip);
    }
    uint8_t* ws = C.scratch_p->take<uint8_t>(ws_bytes + 256);
    int64_t* d_off = C.scratch_p->take<int64_t>(nwaves); int32_t* d_ns = C.scratch_p->take<int32_t>(nwaves);
    int32_t* d_nr = C.scratch_p->take<int32_t>(nwaves); int32_t* d_nch = C.scratch_p->take<int32_t>(nwaves);
    { CopyBatch cb(C.stream); h2d(d_off, w_off, C.stream); h2d(d_ns, w_ns, C.stream); h2d(d_nr, w_nr, C.stream); h2d(d_nch, w_nch, C.stream); }
    S.G = G; S.cws = ws; S.c_off = d_off; S.c_ns = d_ns; S.c_nr = d_nr; S.c_nch = d_nch;
    CoopArgs a;
    a.P = pair_view(B, reversed); a.T = S.T.v; a.G = G;
    a.ws = ws; a.w_ws_off = d_off; a.w_nslots = d_ns; a.w_nrows = d_nr; a.w_nch = d_nch;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len; a.o_abort = S.O.hew;
    a.fill_geom = fill_geom ? 1 : 0;
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
    BandLayout lay = band_layout(L, fill_geom, false);
    lay.mat_u4 = 0;
    S.D = upload_layout(lay, C);
    BandedArgs b;
    b.fill_geom = fill_geom ? 1 : 0;
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
    return live > 0 && (force == 1 || (live <= (size_t)chip(tl_device).simds && n_max >= 4096));      // at most a wave per SIMD
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

// k_banded_sys<.., false> over the list (16 lanes per task for bands of <= 15 slots, else a wave per task), then
// k_banded<false> over the tasks it flagged (N, a taller band)
static ScoreLaunch launch_banded_sys(quicked_batch& B, Context& C, const TaskList& L, bool reversed, int lg, int timed, bool fill_geom = false,
                                     const int32_t* d_cut = nullptr, const int32_t* d_skip = nullptr) {
    ScoreLaunch S;
    S.nt = L.pair.size();
    BandLayout lay = band_layout(L, fill_geom, false);
    lay.mat_u4 = 0;
    int max_nsl = 0;
    for (int32_t v : lay.nslots) max_nsl = std::max(max_nsl, (int)v);
    S.T = upload_tasks(L, C);
    S.D = upload_layout(lay, C);
    S.O = take_out(C, S.nt);
    if (d_cut) {       // as launch_banded_score
        HIP_CHECK(hipMemsetAsync(S.O.score, 0xFF, S.nt * sizeof(int32_t), C.stream));
        hipLaunchKernelGGL(k_apply_cutoffs, dim3((unsigned)((S.nt + 255) / 256)), dim3(256), 0, C.stream, (int)S.nt, S.T.cutoff, S.T.pair, d_cut, d_skip);
    }
    BandedArgs a;
    a.fill_geom = fill_geom ? 1 : 0;
    a.P = pair_view(B, reversed); a.T = S.T.v;
    a.ws = S.D.ws; a.g_ws_off = S.D.ws_off; a.g_nslots = S.D.nslots; a.g_nrows = S.D.nrows; a.g_nch = S.D.nch;
    a.mat = nullptr; a.g_mat_off = S.D.mat_off;
    a.o_score = S.O.score; a.o_first = S.O.first; a.o_last = S.O.last; a.o_posv = S.O.posv; a.o_adv = S.O.adv;
    a.o_maxrow = S.O.len;
    a.only_if = nullptr; a.o_abort = S.O.hew;
    a.lane_rel = env_int("QE_LANE_REL", 1);
    auto* ke = timed ? C.kernel_events(timed - 1) : nullptr;
    if (ke) HIP_CHECK(hipEventRecord(ke->first, C.stream));
    if (lg == 4) launch_groups(C, k_banded_sys<4, false>, with_prio(a), S.nt / 4, 4, 0, false, (size_t)40 * 1024);
    else launch_groups(C, k_banded_sys<6, false>, with_prio(a), S.nt, 4, 0, false, (size_t)40 * 1024);
    a.only_if = S.O.hew;
    if (lg == 6 && max_nsl > 63) launch_groups(C, k_banded_sys2<false>, with_prio(a), S.nt, 4, 0, false, (size_t)40 * 1024);      // bands of 64 .. 127 slots
    launch_groups(C, k_banded<false>, a, L.ngroups(), 8, 0);
    if (ke) HIP_CHECK(hipEventRecord(ke->second, C.stream));
    return S;
}
// whole-text passes on few tasks (QuickEd's stage-3 doubling rounds on the pairs a run left, a BandEd score-only call on one
// pair or a few hundred): 0 = no, else log2 of the lanes per task.  QE_SCORE_SYS = 0 / 1: never / wherever eligible (tests)
static int sys_score_lanes(const TaskList& L, int in_flight, bool fill_geom = false) {
    const int env = env_int("QE_SCORE_SYS", -1);
    if (env == 0) return 0;
    size_t live = 0;
    int max_nsl = 0;
    for (size_t t = 0; t < L.pair.size(); ++t) {
        if (L.pair[t] < 0) continue;
        ++live;
        if (L.tfin[t] != L.n[t]) return 0;                  // a stopped band is exported in k_banded's layout (Hirschberg half passes)
        const HGeom hg = host_geometry(L.m[t], L.n[t], L.cutoff[t]);
        max_nsl = std::max(max_nsl, fill_geom ? hg.ebb : hg.ebb_local);
    }
    if (live == 0 || max_nsl > 127) return 0;
    const size_t nt = L.pair.size(), fl = (size_t)std::max(1, in_flight);
    const Chip& chp = chip(tl_device);
    // one round of waves; two for the pass over the fill's cells, as for the fill itself (run_align's sys_fill: 12.5 k tasks, 3 125 waves)
    if (max_nsl <= 15) return (env == 1 || nt / 4 * fl <= (fill_geom ? 2 : 1) * chp.slots2()) ? 4 : 0;
    return (env == 1 || nt * fl <= chp.frac2(0.54)) ? 6 : 0;                            // a wave per task: about a wave per SIMD
}

// QuickEd's stage 3 (quicked.c:248-278) in ONE launch: k_banded_sys<.., false> with the band doubling on the device.  Every
// task comes back either converged (score = the exact distance within its last cutoff) or flagged with the cutoff it was about
// to run (its band outgrew the group's lanes, N symbols): the host's rounds take those from there.  false: not tried
// (too many tasks for a wave each).  QE_STAGE3_DEVICE = 0 / 1: never / whenever the list is not empty (tests)
static bool stage3_on_device(quicked_batch& B, Context& C, const TaskList& L, std::vector<int32_t>& score, std::vector<u32>& adv,
                             std::vector<int32_t>& flagged, std::vector<int32_t>& cutoff) {
    const int env = env_int("QE_STAGE3_DEVICE", -1);
    if (env == 0) return false;
    size_t live = 0;
    for (int32_t pr : L.pair) live += pr >= 0;
    if (live == 0 || (env != 1 && L.pair.size() > chip(C.device).frac2(0.54))) return false;
    const size_t nt = L.pair.size();
    const DevTasks T = upload_tasks(L, C);
    const TaskOut O = take_out(C, nt);
    int32_t* d_cut = C.scratch_p->take<int32_t>(nt);
    HIP_CHECK(hipMemsetAsync(O.hew, 0, nt * sizeof(int32_t), C.stream));
    BandedArgs a;
    a.P = pair_view(B, false); a.T = T.v;
    a.ws = nullptr; a.g_ws_off = nullptr; a.g_nslots = nullptr; a.g_nrows = nullptr; a.g_nch = nullptr;
    a.mat = nullptr; a.g_mat_off = nullptr;
    a.o_score = O.score; a.o_first = O.first; a.o_last = O.last; a.o_posv = O.posv; a.o_adv = O.adv; a.o_maxrow = O.len;
    a.only_if = nullptr; a.o_abort = O.hew; a.doubling = 1; a.o_cutoff = d_cut;
    launch_groups(C, k_banded_sys<6, false>, with_prio(a), nt, 4, 0, false, (size_t)40 * 1024);
    d2h(score, O.score, nt, C.stream); d2h(adv, O.adv, nt, C.stream); d2h(flagged, O.hew, nt, C.stream); d2h(cutoff, d_cut, nt, C.stream);
    HIP_CHECK(hipStreamSynchronize(C.stream));
    return true;
}

static void run_banded_score(quicked_batch& B, Context& C, const TaskList& L, bool reversed, StageResult* R,
                             bool fetch, int32_t** d_score_out, PendingFetch* pf = nullptr) {
    // one wavefront per alignment only where the cooperative on-chip form has no room (a band of fewer than 8 slots): with
    // G = 8 lanes per alignment and 4-slot passes that form does a 10 kb pair in 2.7 ms, the wave form in 4.3
    const bool forced = env_set("QE_COOP_G") || env_int("QE_WAVE", -1) == 1;      // tests of the other forms
    const int lg = forced ? 0 : sys_score_lanes(L, fetch ? 1 : C.in_flight);
    const int G0 = lg ? 1 : coop_lanes(L, fetch ? 1 : C.in_flight);
    const bool wave = !lg && wave_form_wanted(L) && (G0 < 2 || env_int("QE_WAVE", -1) == 1);
    const int G = wave ? 1 : G0;
    const ScoreLaunch S = lg ? launch_banded_sys(B, C, L, reversed, lg, 1) :
                          wave ? launch_banded_wave(B, C, L, reversed, 1)
                               : ((G >= 2) ? launch_banded_coop(B, C, L, reversed, G, 1) : launch_banded_score(B, C, L, reversed, 1));
    if (d_score_out) *d_score_out = S.O.score;
    if (pf && !fetch) {
        pf->kind = 1; pf->task_pair = L.pair; pf->d_score = S.O.score; pf->d_adv = S.O.adv; pf->counter_slot = 0;
        pf->d_abort = (G >= 2) ? S.O.hew : nullptr;
    }
    if (fetch && R) {
        FetchBatch fb(C);
        if (G >= 2) fb.add(R->hew, S.O.hew, S.nt);              // abort flags (diagnostics)
        fb.add(R->score, S.O.score, S.nt); fb.add(R->adv, S.O.adv, S.nt);
        fb.sync();
    }
}

// QuickEd with only_score (quicked.c:283-294 aligns and extract_results counts the alignment's edits, quicked.c:34-56): the
// bound is an upper bound of the distance (it is the cost of a real path), so the banded alignment with that cutoff is an
// optimal one and its edit count is the value of the fill's end cell -- which a pass over the FILL's cells (its band
// geometry, its bookkeeping: BandedArgs::fill_geom) computes without storing a checkpoint, walking a path or formatting a
// run.  A read whose align step would split (bpm_hirschberg.c:63-65) keeps it -- the children's distances add up to the same
// end value, but the levels' half passes are the faster way through a band that long (quicked_classic).  One lane per alignment,
// or the systolic forms where the launch is short of waves (run_fill_score).
// Measured on 10 kb pairs (align step / score pass, profiles/r06_t_probe_score_pass*.txt): one run alone 1 pair 3.9 / 2.9 ms,
// 1 k pairs 4.7 / 3.2, 4 k 4.8 / 3.2, 8 k 6.1 / 3.8, 12.5 k 8.2 / 5.0, 25 k 13.5 / 8.6, 100 k 21.9 / 14.3; a stream of queued runs
// 1 k pairs 0.44 / 1.5 M alignments/s, 4 k 2.9 / 4.8, 12.5 k 5.6 / 9.1, 25 k 6.3 / 10.4, 100 k 7.05 / 11.5 -- every size of such reads (taller bands: quicked_score_pass_fits).
// In the fast flow the pass reads its cutoffs from the device like the align step does (k_apply_cutoffs); pairs with lower-case /
// IUPAC symbols leave the flow there (Stage1Args::flags) and in the host-driven flow keep the whole batch on the align step.
// QE_QUICKED_SCORE_PASS = 0: never (the align step: tests), 1: wherever the results allow it and no read splits; QE_QUICKED_SCORE_PASS_FAST = 0: synchronous runs take the pass at the
// end of the host-driven flow only.
static bool quicked_score_pass_wanted() { return env_int("QE_QUICKED_SCORE_PASS", 1) != 0; }
// ... and for THIS list (cutoffs: the bounds, or the fast flow's estimates): a run the caller waits for, of a few thousand
// tasks whose bands are too tall for the systolic forms, is a handful of one-lane waves with one wave's chain each, and the
// align step's cooperative fill is ahead there (2 000 pairs of 20 kb at 5 %: 11 ms against 15; of 10 kb at 10 %: 7.8 / 7.8;
// from 8 000 pairs on the pass wins: 20 / 16 and 12.6 / 7.9 ms).  Queued runs fill the chip together: always.
static bool quicked_score_pass_fits(const TaskList& L, bool fetch) {
    if (env_int("QE_QUICKED_SCORE_PASS", -1) == 1 || !fetch) return true;
    if (sys_score_lanes(L, 1, true) != 0) return true;
    size_t live = 0;
    for (int32_t pr : L.pair) live += pr >= 0;
    return live >= (size_t)chip(tl_device).simds * 4;
}
// fetch: the scores and block-advance counts to the host (R).  pf: a queued run's (kind 1; the fast flow's fields are the caller's)
static void run_fill_score(quicked_batch& B, Context& C, const TaskList& L, StageResult* R, bool fetch, int32_t** d_score_out,
                           PendingFetch* pf = nullptr, const int32_t* d_cut = nullptr, const int32_t* d_skip = nullptr) {
    // launches of few waves: the systolic forms (16 lanes or a wave per task), as the score-only passes and the fills take them
    // (the cooperative LDS form can run the pass too -- launch_banded_coop(.., fill_geom) -- but loses to the one-lane kernel
    // wherever it was tried: 2 000 / 8 000 / 30 000 pairs of 20 kb alone 21 / 21 / 30 ms against 15 / 16 / 21, its bands of ~20
    // slots leave G = 8 lanes 2 G + 4 slots and most tasks to the fallback pass: profiles/r06_y_probe_coop_form.txt)
    const int lg = sys_score_lanes(L, fetch ? 1 : C.in_flight, true);
    const int Gc = lg ? 0 : env_int("QE_SCORE_PASS_COOP_G", 0);       // tests / probes: that many lanes per task in the cooperative LDS form
    const ScoreLaunch S = lg ? launch_banded_sys(B, C, L, false, lg, 2 /* timed as a fill */, true, d_cut, d_skip) :
                          Gc >= 2 ? launch_banded_coop(B, C, L, false, Gc, 2, true, d_cut, d_skip)
                                  : launch_banded_score(B, C, L, false, 2, true, d_cut, d_skip);
    if (d_score_out) *d_score_out = S.O.score;
    if (pf && !fetch) {
        pf->kind = 1; pf->task_pair = L.pair; pf->d_score = S.O.score; pf->d_adv = S.O.adv; pf->counter_slot = 1;
        pf->ok_status = QUICKED_WIP;
    }
    if (fetch && R) {
        FetchBatch fb(C);
        fb.add(R->score, S.O.score, S.nt); fb.add(R->adv, S.O.adv, S.nt);
        fb.sync();
    }
}

// One wave per alignment wherever the strings are more than a few runs long, else one lane per alignment.  Decided before the
// traceback runs: the wave form wants every task's runs in a stretch of their own (TraceArgs::runs_by_task).  Round 5: the
// wave form used to be kept for few alignments or very long reads; measured per read length and batch size
// (tools/probe_format_wave.sh) it is never behind -- streams of 12.5 k pairs of 10 kb 5.1 -> 5.8 M alignments/s, one such
// batch alone 9.8 -> 8.8 ms, 100 k pairs 6.7 -> 6.95 M, 100 k pairs of 3 kb 19.1 -> 21.9 M, of 300 bases 20.2 -> 21.8 M
static bool wave_formatter_wanted(const quicked_batch& B, const SegList& SL, bool want_strings) {
    const size_t nr = SL.root_pair.size(), nseg = SL.kind.size();
    size_t pool_bytes = 0;
    if (want_strings) for (size_t b : SL.bound) pool_bytes += b;
    const int wave_env = env_int("QE_FORMAT_WAVE", -1);      // tests force either form
    return B.cigar_style != 2 && nr > 0 && (wave_env >= 0 ? wave_env != 0 : (nseg > 0 && pool_bytes / nr >= 512));
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
    {
        CopyBatch cb(C.stream);
        h2d(d_off, SL.off, C.stream); h2d(d_kind, SL.kind, C.stream); h2d(d_a, SL.a, C.stream); h2d(d_b, SL.b, C.stream);
        h2d(d_rootpair, SL.root_pair, C.stream);
    }
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
    std::vector<int32_t> okv;
    const size_t nr = A.nroots;
    const bool strings = want_strings && A.pool != nullptr && A.pool_bytes > 0;
    const size_t small_bytes = 6 * (((nr * 8) + 63) & ~(size_t)63) + (strings ? A.pool_bytes + 64 : 0);
    const size_t base = B.wr->cigar_pool.size;
    if (small_bytes <= ((size_t)256 << 10)) {
        // few alignments (a single quicked_align call): the per-root arrays and the string pool up to its bound arrive in the
        // context's pinned block through ONE copy launch and ONE synchronisation (five copies, the strings' own round trip
        // and two synchronisations otherwise: ~0.1 ms of a call)
        uint8_t* st = C.small_pinned(small_bytes + 256);
        size_t top = 0;
        auto get = [&](const void* src, size_t bytes) { uint8_t* p = st + top; copy_kernel(p, src, bytes, C.stream); top += (bytes + 63) & ~(size_t)63; return p; };
        const uint8_t *h_len, *h_edits, *h_nops, *h_off = nullptr, *h_ok = nullptr, *h_pool = nullptr;
        {
            CopyBatch cb(C.stream);
            h_len = get(A.len, nr * 4); h_edits = get(A.edits, nr * 4); h_nops = get(A.nops, nr * 4);
            if (want_strings) h_off = get(A.str_off, nr * 8);
            if (A.ok) h_ok = get(A.ok, nr * 4);
            if (strings) h_pool = get(A.pool, A.pool_bytes);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(C.stream));
        len.assign((const int32_t*)h_len, (const int32_t*)h_len + nr); edits.assign((const int32_t*)h_edits, (const int32_t*)h_edits + nr);
        nops.assign((const int32_t*)h_nops, (const int32_t*)h_nops + nr);
        if (h_off) off.assign((const int64_t*)h_off, (const int64_t*)h_off + nr);
        if (h_ok) okv.assign((const int32_t*)h_ok, (const int32_t*)h_ok + nr);
        int64_t total = 0;
        if (want_strings) for (size_t i = 0; i < nr; ++i) total = std::max<int64_t>(total, off[i] + len[i] + 1);
        if (total && h_pool) {
            if ((size_t)total > A.pool_bytes) throw HipError{hipErrorUnknown, "CIGAR strings beyond their pool's bound", __LINE__};
            B.wr->cigar_pool.reserve(base + (size_t)total);
            memcpy(B.wr->cigar_pool.p + base, h_pool, (size_t)total);
            B.wr->cigar_pool.size = base + (size_t)total;
        }
    } else {
        d2h(len, A.len, A.nroots, C.stream); d2h(edits, A.edits, A.nroots, C.stream); d2h(nops, A.nops, A.nroots, C.stream);
        if (want_strings) d2h(off, A.str_off, A.nroots, C.stream);
        if (A.ok) d2h(okv, A.ok, A.nroots, C.stream);
        HIP_CHECK(hipStreamSynchronize(C.stream));
        int64_t total = 0;
        if (want_strings) for (size_t i = 0; i < A.nroots; ++i) total = std::max<int64_t>(total, off[i] + len[i] + 1);
        if (total) {
            B.wr->cigar_pool.reserve(base + (size_t)total);
            HIP_CHECK(hipMemcpyAsync(B.wr->cigar_pool.p + base, A.pool, (size_t)total, hipMemcpyDeviceToHost, C.stream));
            HIP_CHECK(hipStreamSynchronize(C.stream));
            B.wr->cigar_pool.size = base + (size_t)total;
        }
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
    if (W == 2 && O_ == 1) {
        // few waves: four lanes per alignment run the chain of full windows first (k_windowed_quad, a third of the one-lane
        // chain's latency for 1.9 x its instructions), the one-lane kernel then only has every task's clamped last windows
        // left.  Worth it while the launches in flight leave SIMDs idle; QE_WINDOWED_QUAD = 0 / 1: never / always (tests)
        const int quad = env_int("QE_WINDOWED_QUAD", -1);
        const size_t waves = (size_t)ng * 4 * (size_t)std::max(1, fetch ? 1 : C.in_flight);
        if (score_only && (quad == 1 || (quad != 0 && waves <= chip(C.device).slots2()))) {
            a.state = C.scratch_p->take<int32_t>(5 * nt);
            launch_groups(C, k_windowed_quad, with_prio(a), nt / 16, 4, (size_t)QE_WQ_LDS_PER_WAVE, /* chain */ true);
        }
        launch_groups(C, k_windowed, a, (size_t)ng, 8, 8192, /* chain */ true);
    } else {
        // few waves: sixteen lanes per alignment (k_windowed_sys: the window's block rows as a systolic array, sixteen traceback
        // tiles rebuilt at a time); what it flags (N, non-canonical symbols) stays with the one-lane kernel.
        // QE_WINDOWED_SYS = 0 / 1: never / wherever eligible (tests)
        const int wsys = env_int("QE_WINDOWED_SYS", -1);
        const size_t waves = (size_t)ng * 16 * (size_t)std::max(1, fetch ? 1 : C.in_flight);
        // not for W == 2 with the x86 SSE semantics (bpm_windowed.c:577: the SSE window kernel runs whenever window_size == 2
        // and force_scalar is off, whatever the overlap): k_windowed_sys computes the scalar kernel's windows, the one-lane
        // kernel's history path has the SSE boundary pattern
        if (score_only && W <= 15 && !(sse && W == 2) && a.cp_path != 0 && (wsys == 1 || (wsys != 0 && waves <= 2 * chip(C.device).slots2()))) {
            a.o_abort = C.scratch_p->take<int32_t>(nt);
            launch_groups(C, k_windowed_sys, with_prio(a), nt / 4, 4, 0, /* chain */ false, (size_t)40 * 1024);
            a.only_if = a.o_abort;
        }
        launch_groups(C, k_windowed_cp, a, (size_t)ng, 8, 8192, /* chain */ true);
    }
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
        { FetchBatch fb(C); fb.add(R->score, O.score, nt); fb.add(R->hew, O.hew, nt); fb.add(R->steps, O.steps, nt); fb.sync(); }
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
// BUFFER_SIZE_16M of bpm_hirschberg.c:65; QE_SPLIT_BYTES lowers it so tests can force many split levels on small inputs
static uint64_t split_threshold() {
    return (uint64_t)env_ll("QE_SPLIT_BYTES", (long long)1 << 24);
}
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
        { CopyBatch cb(C.stream); h2d(dm, hm, C.stream); h2d(dn1, hn1, C.stream); h2d(dn2, hn2, C.stream); }
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
    {
        CopyBatch cb(C.stream);
        h2d(d_ws_off, ws_off, C.stream); h2d(d_mat_off, mat_off, C.stream); h2d(d_runs_off, lay.runs_off, C.stream);
        h2d(d_nslots, lay.nslots, C.stream); h2d(d_nrows, lay.nrows, C.stream); h2d(d_nch, lay.nch, C.stream);
        h2d(d_runs_cap, lay.runs_cap, C.stream);
    }
    u32* d_runs = C.scratch_p->take<u32>(lay.runs_u32 + 64);
    const bool wave_fmt = wave_formatter_wanted(B, SL, want_cigar);       // also the layout the traceback leaves its runs in
    // Lanes per leaf for the fill: 1 where the leaves fill the chip -- and wherever the cutoff is a tight bound of the distance
    // (QuickEd's bound, Hirschberg's exact child distances): such a band is pruned down to a few slots, its height test
    // (first + 2 < last) cannot be decided chunks ahead, and the cooperative protocol hands most leaves back to the
    // one-lane kernel (config 4's leaves: both kernels ran, 45 + 36 ms instead of 38).  A user bandwidth leaves the band
    // tall: few long BandEd alignments with CIGAR fill with G lanes each.  QE_COOP_FILL_G forces a width (tests).
    const bool fill_forced = env_set("QE_COOP_FILL_G");
    int Gfill = (env_int("QE_COOP_LDS", 1) == 0 || (tight_runs && !fill_forced)) ? 1 : coop_lanes(LL, fetch ? 1 : C.in_flight, true);
    // Round 4: ... except where the bound is LARGE.  A pair with large indels has a bound of thousands, its band is 40-60
    // slots tall for most of its length (the edge pruning only bites as the score nears the cutoff), and a launch of such
    // leaves is a few waves of one lane's chain each (the pairs a QuickEd run left for the host-driven flow: 25-35 ms for
    // thirteen waves).  There the leaves whose bands are tall enough for G >= 4 lanes (>= 3 G + 4 slots) fill cooperatively
    // and the others arrive flagged and stay with the one-lane kernel -- one list may hold both kinds.  Not where the
    // cutoffs are still on the device (the fast flow's chip-filling runs), and only for launches of at most 64 one-lane
    // waves: 20 k such pairs (313 waves one-lane, 3 750 cooperative) fill faster one lane each (76 against 85 ms per run),
    // the 813 pairs a 100 k-pair run leaves gain (50 -> 47 ms per batch of the mixed stream).
    // tight bounds in a launch of few waves: the systolic fill (k_banded_sys, below) -- 16 lanes per leaf for bands of <= 15
    // slots, a wave per leaf for bands of <= 63 (the bounds of pairs with large indels), the one-lane kernel for what is left.
    // QE_FILL_SYS = 0 / 1: never / wherever the bound is tight (tests)
    const int sys_env = env_int("QE_FILL_SYS", -1);
    const size_t in_fl = (size_t)std::max(1, fetch ? 1 : C.in_flight);
    const Chip& chp = chip(C.device);
    const bool sys_fill = Gfill < 2 && tight_runs && (sys_env == 1 || (sys_env != 0 && (size_t)ng * 16 * in_fl <= 2 * chp.slots2()));
    std::vector<int32_t> hew_init;
    if (!sys_fill && Gfill < 2 && tight_runs && !fill_forced && !d_cut && env_int("QE_COOP_LDS", 1) != 0 && env_int("QE_COOP_TALL_FILL", 1) != 0 &&
        (size_t)ng * (size_t)std::max(1, fetch ? 1 : C.in_flight) <= 64) {
        size_t live = 0;
        std::vector<int> ebb(nt, 0);
        for (size_t t = 0; t < nt; ++t) if (LL.pair[t] >= 0) { ebb[t] = host_geometry(LL.m[t], LL.n[t], LL.cutoff[t]).ebb; ++live; }
        for (int G : {16, 8, 4}) {
            size_t ok = 0;
            for (size_t t = 0; t < nt; ++t) ok += LL.pair[t] >= 0 && ebb[t] >= 3 * G + 4;
            if (ok >= 16 && ok * 4 >= live) {              // a quarter of the leaves or more: the launch's duration is theirs
                Gfill = G;
                hew_init.assign(nt, 1);
                for (size_t t = 0; t < nt; ++t) if (LL.pair[t] >= 0 && ebb[t] >= 3 * G + 4) hew_init[t] = 0;
                break;
            }
        }
    }
    int32_t* d_hew_init = nullptr;
    if (!hew_init.empty()) { d_hew_init = C.scratch_p->take<int32_t>(nt); h2d(d_hew_init, hew_init, C.stream); }
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
                if (d_hew_init) copy_kernel(O.hew + o, d_hew_init + o, (size_t)(g1 - g0) * 64 * sizeof(int32_t), C.stream);
                else HIP_CHECK(hipMemsetAsync(O.hew + o, 0, (size_t)(g1 - g0) * 64 * sizeof(int32_t), C.stream));
                launch_groups(C, k_banded_coop_lds<true>, x, (size_t)(g1 - g0) * 64 / NAf, 8, (size_t)x.lds_per_wave);
                a.only_if = O.hew + o;
            }
        }
        // the cooperative kernels need few registers and no LDS: four workgroups per CU (40 KB each), so that a launch of up
        // to 4 096 waves is resident at once instead of running in two rounds of 2 048 (12.5 k leaves x 16 lanes = 3 125 waves)
        const size_t sys_pin = (size_t)40 * 1024;
        if (sys_fill && a.only_if == nullptr) {
            int maxns = 0;
            for (int g = g0; g < g1; ++g) maxns = std::max(maxns, (int)lay.nslots[g]);
            a.o_abort = O.hew + o;
            launch_groups(C, k_banded_sys<4, true>, with_prio(a), (size_t)(g1 - g0) * 16, 4, 0, false, sys_pin);
            a.only_if = O.hew + o;
            // what it flagged for its height: one wave per leaf while the sub-batch is small enough for that
            if (maxns > 15 && (sys_env == 1 || (size_t)(g1 - g0) * 64 * in_fl <= 2 * chp.slots2())) {
                launch_groups(C, k_banded_sys<6, true>, with_prio(a), (size_t)(g1 - g0) * 64, 4, 0, false, sys_pin);
                // ... and what THAT flagged for its height (64 .. 127 slots): two rows per lane, two sweeps per chunk
                if (maxns > 63) launch_groups(C, k_banded_sys2<true>, with_prio(a), (size_t)(g1 - g0) * 64, 4, 0, false, sys_pin);
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
        // a launch of few waves: sixteen lanes per leaf rebuild the tiles along the path's diagonal together and hand the
        // walk from tile to tile (k_traceback_sys); what it flags (N, non-canonical symbols) stays with the one-lane kernel.
        // QE_TRACE_SYS = 0 / 1: never / always (tests)
        // lanes per leaf: 16 while the launch is one round of waves (two per SIMD at 246 VGPRs: ~8 k leaves), 8 up to ~3 300
        // waves (the walk runs in all lanes of a group at once in both: one batch of 12.5 k leaves alone 8.1 ms with 8 lanes,
        // 8.6 with 16, 10.3 with the tile-by-tile walk of the round's first half), 4 up to ~1 700 waves
        const int tsys = env_int("QE_TRACE_SYS", -1);
        const size_t gw = (size_t)(g1 - g0) * (size_t)std::max(1, fetch ? 1 : C.in_flight);      // one-lane waves in flight
        int tlg = 0;
        if (tsys > 1) tlg = tsys == 4 ? 2 : (tsys == 8 ? 3 : 4);
        // (2 100 / 3 300 / 1 700 waves on the 1 024 SIMDs of an MI355X: one round at two waves per SIMD, 1.6 rounds, 0.8)
        else if (tsys != 0) tlg = (gw * 16 <= chp.frac2(1.03) || tsys == 1) ? 4 : (gw * 8 <= chp.frac2(1.61) ? 3 : (gw * 4 <= chp.frac2(0.83) ? 2 : 0));
        if (tlg) {
            tr.o_abort = C.scratch_p->take<int32_t>((size_t)(g1 - g0) * 64);
            const size_t nwv = (size_t)(g1 - g0) << tlg;
            if (tlg == 4) launch_groups(C, k_traceback_sys<4>, with_prio(tr), nwv, 4, 0, false, sys_pin);
            else if (tlg == 3) launch_groups(C, k_traceback_sys<3>, with_prio(tr), nwv, 4, 0, false, sys_pin);
            else launch_groups(C, k_traceback_sys<2>, with_prio(tr), nwv, 4, 0, false, sys_pin);
            tr.only_if = tr.o_abort;
        }
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
            { FetchBatch fb(C); fb.add(adv, O.adv, nt); fb.add(steps, O.steps, nt); fb.sync(); }
            for (size_t t = 0; t < nt; ++t) if (LL.pair[t] >= 0) { stats->fill_adv += adv[t]; stats->tb_steps += steps[t]; B.note_pair(LL.pair[t], 1, adv[t]); B.note_pair(LL.pair[t], 3, steps[t]); }
        }
        fetch_alignments(B, C, SL, AO, want_cigar, ok_status, &root_status);
    }
}

}  // namespace qe
