// Host stand-ins for the kernels of quicked_amd/csrc/qe_kernels.hip, for the sanitizer build of the library's HOST layer
// (see hip/hip_runtime.h next to this file).  The kernels the host logic depends on for its control flow are real code here
// (in-stream copies, the offsets scan, the stage-1 rule, the cut-off hand-over); the alignment kernels leave plausible
// zeros -- a bound of `QE_STUB_BOUND` and, for every `QE_STUB_SKIP_EVERY`-th task, the "goes on to stage 2" flag, so that
// the fast flow's overflow path, the early-finish threads and the merged flows all get work.  TEST INFRASTRUCTURE.
#pragma once
#include <algorithm>
#include "qe_types.h"

namespace qe {

struct CopyTable {
    enum { MAX = 16 };
    int32_t n;
    uint4* dst[MAX]; const uint4* src[MAX]; int64_t n_u4[MAX];
};
static inline int stub_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static void k_copy_multi(CopyTable T) { for (int i = 0; i < T.n; ++i) if (T.n_u4[i] > 0) memmove(T.dst[i], T.src[i], (size_t)T.n_u4[i] * 16); }
static void k_gather_words(int nspans, const int64_t* src_addr, const int64_t* dst_off, const int32_t* nwords, u64* dst) {
    for (int e = 0; e < nspans; ++e) memmove(dst + dst_off[e], reinterpret_cast<const u64*>(src_addr[e]), (size_t)nwords[e] * 8);
}
static void k_copy_total(uint4* dst, const uint4* src, const int64_t* total, int64_t cap_u4) {
    const int64_t n = std::min((*total + 15) >> 4, cap_u4);
    if (n > 0) memmove(dst, src, (size_t)n * 16);
}
static void k_scan_offsets(const int32_t* len, const int32_t* pair, int64_t* off, int64_t* total, int n) {
    int64_t carry = 0;
    for (int i = 0; i < n; ++i) { off[i] = carry; if (pair[i] >= 0) carry += (int64_t)len[i] + 1; }
    *total = carry;
}
static void k_stage1_decide(Stage1Args A) {
    const int every = stub_env("QE_STUB_SKIP_EVERY", 0);
    for (int t = 0; t < A.nt; ++t) {
        int cut = 0, skip = 1;
        u32 steps = 0;
        if (A.pair[t] >= 0) {
            cut = A.score[t];
            skip = ((every > 0 && t % every == 0) ? 1 : 0) | ((cut > A.est[t]) ? 2 : 0);
            steps = A.steps[t];
        }
        A.o_cut[t] = cut; A.o_skip[t] = skip; A.o_steps[t] = steps;
    }
}
static void k_apply_cutoffs(int nt, int32_t* cutoff, int32_t* pair, const int32_t* cut, const int32_t* skip) {
    for (int t = 0; t < nt; ++t) { cutoff[t] = cut[t]; if (skip[t] != 0) pair[t] = -1; }
}
// WindowEd: a bound per task (what stage 1 hands the align step), no high-error windows
static void stub_windowed(const WindowArgs& A) {
    const int bound = stub_env("QE_STUB_BOUND", 70);
    for (int t = 0; t < A.T.ntasks; ++t) {
        if (A.T.pair[t] < 0 || (A.only_if && A.only_if[t] == 0)) continue;
        A.o_score[t] = bound; A.o_hew[t] = 0; A.o_steps[t] = 1;
        if (!A.score_only) { A.o_nruns[t] = 0; A.o_nops[t] = 0; A.o_edits[t] = 0; }
    }
}
static void k_windowed(WindowArgs A) { stub_windowed(A); }
static void k_windowed_cp(WindowArgs A) { stub_windowed(A); }
static void k_windowed_sys(WindowArgs A) {
    for (int t = 0; t < A.T.ntasks; ++t) if (A.T.pair[t] >= 0 && A.o_abort) A.o_abort[t] = 0;
    stub_windowed(A);
}
static void k_windowed_quad(WindowArgs A) { if (A.state) memset(A.state, 0, (size_t)5 * A.T.ntasks * sizeof(int32_t)); }
// BandEd: a score per task, nothing flagged
static void stub_banded(const BandedArgs& A) {
    for (int t = 0; t < A.T.ntasks; ++t) {
        if (A.T.pair[t] < 0 || (A.only_if && A.only_if[t] == 0)) continue;
        A.o_score[t] = stub_env("QE_STUB_BOUND", 70) / 2; A.o_first[t] = 0; A.o_last[t] = 0; A.o_posv[t] = 0; A.o_adv[t] = 1; A.o_maxrow[t] = 1 << 20;
    }
}
template <bool FILL> static void k_banded(BandedArgs A) { stub_banded(A); }
static void k_banded_wave(BandedArgs A) { stub_banded(A); }
template <int LG, bool FILL> static void k_banded_sys(BandedArgs A) {
    for (int t = 0; t < A.T.ntasks; ++t) if (A.T.pair[t] >= 0 && A.o_abort) A.o_abort[t] = 0;
    stub_banded(A);
}
template <bool FILL> static void k_banded_sys2(BandedArgs A) {
    for (int t = 0; t < A.T.ntasks; ++t) if (A.T.pair[t] >= 0 && A.o_abort && !(A.only_if && A.only_if[t] == 0)) A.o_abort[t] = 0;
    stub_banded(A);
}
static void k_banded_coop(CoopArgs A) {
    for (int t = 0; t < A.T.ntasks; ++t) {
        if (A.T.pair[t] < 0) continue;
        A.o_score[t] = stub_env("QE_STUB_BOUND", 70) / 2; A.o_first[t] = 0; A.o_last[t] = 0; A.o_posv[t] = 0; A.o_adv[t] = 1; A.o_maxrow[t] = 1 << 20;
    }
}
template <bool FILL> static void k_banded_coop_lds(CoopLdsArgs X) { k_banded_coop(X.A); }
static void stub_trace(const TraceArgs& A) {
    for (int t = 0; t < A.T.ntasks; ++t) {
        if (A.T.pair[t] < 0 || (A.only_if && A.only_if[t] == 0)) continue;
        A.o_nruns[t] = 0; A.o_nops[t] = 0; A.o_edits[t] = 0; A.o_steps[t] = 1;
    }
}
static void k_traceback(TraceArgs A) { stub_trace(A); }
template <int LG> static void k_traceback_sys(TraceArgs A) {
    for (int t = 0; t < A.T.ntasks; ++t) if (A.T.pair[t] >= 0 && A.o_abort) A.o_abort[t] = 0;
    stub_trace(A);
}
static void k_join(JoinArgs A) { for (int j = 0; j < A.nnodes; ++j) { A.o_best[j] = A.m[j] / 2; A.o_score_l[j] = 1; A.o_score_r[j] = 1; A.o_ok[j] = 1; } }
// CIGAR strings: one "1M" per entry, so that string pools, offsets and the D2H of the strings are exercised
template <bool WRITE> static void k_format_segs(SegFormatArgs A) {
    for (int i = 0; i < A.npairs; ++i) {
        if (!WRITE) { A.o_len[i] = 2; A.o_edits[i] = 0; A.o_nops[i] = 1; }
        else if (A.pool && A.str_off) { char* s = A.pool + A.str_off[i]; s[0] = '1'; s[1] = 'M'; s[2] = 0; }
    }
}
template <bool WRITE> static void k_format_segs_wave(SegFormatArgs A) { k_format_segs<WRITE>(A); }
static void k_check_segs(SegCheckArgs C) { for (int i = 0; i < C.F.npairs; ++i) C.o_ok[i] = 1; }
static void k_check_strings(PairView, int npairs, const char*, const int64_t*, int32_t* o_ok) { for (int i = 0; i < npairs; ++i) o_ok[i] = 1; }
static void k_pack(PackArgs A) { if (A.flags) for (int i = 0; i < A.nseq; ++i) A.flags[i] |= 0u; }
static void k_unpack_wire(WireArgs) {}
static void k_reverse_planes(RevArgs) {}

}  // namespace qe
