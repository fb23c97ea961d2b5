// skew_asm_bench.hip -- the hand-scheduled 4-slot skewed pass (tools/gen_skew_asm.py) against hipcc's run64_skew<4>:
// bit-exactness on random states and cycles per block-column per SIMD at 1 .. 4 waves per SIMD, registers only (the harness
// of tools/valu_rate.hip Part B).
//   hipcc --offload-arch=gfx950 -O3 -I<dir with the generated .inc files> tools/skew_asm_bench.hip -o tools/bin/skew_asm_bench
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../quicked_amd/csrc/qe_kernels.hip"
#include "skew_d8.inc"
#include "skew_prog.inc"
#include "skew_d32.inc"
#include "skew_vop3.inc"
#include "skew_low.inc"
#include "skew_c8.inc"
#include "skew2.inc"
#include "skew2_prog.inc"
#include "skew2_vop3.inc"

using qe::u32;
using qe::u64;

#define SKEW_ASM_BODY(TEXT)                                                                                                        \
    asm volatile(TEXT                                                                                                              \
                 : "+" QE_SKEW_ASM_K4_C32_P0(P[0]), "+" QE_SKEW_ASM_K4_C32_P1(P[1]), "+" QE_SKEW_ASM_K4_C32_P2(P[2]),              \
                   "+" QE_SKEW_ASM_K4_C32_P3(P[3]), "+" QE_SKEW_ASM_K4_C32_M0(M[0]), "+" QE_SKEW_ASM_K4_C32_M1(M[1]),              \
                   "+" QE_SKEW_ASM_K4_C32_M2(M[2]), "+" QE_SKEW_ASM_K4_C32_M3(M[3]), "=" QE_SKEW_ASM_K4_C32_GP(gP),                \
                   "=" QE_SKEW_ASM_K4_C32_GM(gM)                                                                                   \
                 : QE_SKEW_ASM_K4_C32_A0(a[0]), QE_SKEW_ASM_K4_C32_A1(a[1]), QE_SKEW_ASM_K4_C32_A2(a[2]), QE_SKEW_ASM_K4_C32_A3(a[3]), \
                   QE_SKEW_ASM_K4_C32_B0(b[0]), QE_SKEW_ASM_K4_C32_B1(b[1]), QE_SKEW_ASM_K4_C32_B2(b[2]), QE_SKEW_ASM_K4_C32_B3(b[3]), \
                   QE_SKEW_ASM_K4_C32_T0(t0), QE_SKEW_ASM_K4_C32_T1(t1), QE_SKEW_ASM_K4_C32_HP(hp), QE_SKEW_ASM_K4_C32_HM(hm)      \
                 : QE_SKEW_ASM_K4_C32_CLOBBERS)

#define SKEW_ASM_BODY_LOW(TEXT)                                                                                                    \
    asm volatile(TEXT                                                                                                              \
                 : "+" QE_SKEW_ASM_K4_C32_LOW_P0(P[0]), "+" QE_SKEW_ASM_K4_C32_LOW_P1(P[1]), "+" QE_SKEW_ASM_K4_C32_LOW_P2(P[2]),  \
                   "+" QE_SKEW_ASM_K4_C32_LOW_P3(P[3]), "+" QE_SKEW_ASM_K4_C32_LOW_M0(M[0]), "+" QE_SKEW_ASM_K4_C32_LOW_M1(M[1]),  \
                   "+" QE_SKEW_ASM_K4_C32_LOW_M2(M[2]), "+" QE_SKEW_ASM_K4_C32_LOW_M3(M[3]), "=" QE_SKEW_ASM_K4_C32_LOW_GP(gP),    \
                   "=" QE_SKEW_ASM_K4_C32_LOW_GM(gM)                                                                               \
                 : QE_SKEW_ASM_K4_C32_LOW_A0(a[0]), QE_SKEW_ASM_K4_C32_LOW_A1(a[1]), QE_SKEW_ASM_K4_C32_LOW_A2(a[2]), QE_SKEW_ASM_K4_C32_LOW_A3(a[3]), \
                   QE_SKEW_ASM_K4_C32_LOW_B0(b[0]), QE_SKEW_ASM_K4_C32_LOW_B1(b[1]), QE_SKEW_ASM_K4_C32_LOW_B2(b[2]), QE_SKEW_ASM_K4_C32_LOW_B3(b[3]), \
                   QE_SKEW_ASM_K4_C32_LOW_T0(t0), QE_SKEW_ASM_K4_C32_LOW_T1(t1), QE_SKEW_ASM_K4_C32_LOW_HP(hp), QE_SKEW_ASM_K4_C32_LOW_HM(hm) \
                 : QE_SKEW_ASM_K4_C32_LOW_CLOBBERS)
// the 8-column pass, GP / GM in-out: four calls per 32-column half (a rolled loop: ~5 KB of code instead of ~19 KB)
#define SKEW_ASM_BODY_ACC(TEXT)                                                                                                    \
    asm volatile(TEXT                                                                                                              \
                 : "+" QE_SKEW_ASM_K4_C32_P0(P[0]), "+" QE_SKEW_ASM_K4_C32_P1(P[1]), "+" QE_SKEW_ASM_K4_C32_P2(P[2]),              \
                   "+" QE_SKEW_ASM_K4_C32_P3(P[3]), "+" QE_SKEW_ASM_K4_C32_M0(M[0]), "+" QE_SKEW_ASM_K4_C32_M1(M[1]),              \
                   "+" QE_SKEW_ASM_K4_C32_M2(M[2]), "+" QE_SKEW_ASM_K4_C32_M3(M[3]), "+" QE_SKEW_ASM_K4_C32_GP(gP),                \
                   "+" QE_SKEW_ASM_K4_C32_GM(gM)                                                                                   \
                 : QE_SKEW_ASM_K4_C32_A0(a[0]), QE_SKEW_ASM_K4_C32_A1(a[1]), QE_SKEW_ASM_K4_C32_A2(a[2]), QE_SKEW_ASM_K4_C32_A3(a[3]), \
                   QE_SKEW_ASM_K4_C32_B0(b[0]), QE_SKEW_ASM_K4_C32_B1(b[1]), QE_SKEW_ASM_K4_C32_B2(b[2]), QE_SKEW_ASM_K4_C32_B3(b[3]), \
                   QE_SKEW_ASM_K4_C32_T0(t0), QE_SKEW_ASM_K4_C32_T1(t1), QE_SKEW_ASM_K4_C32_HP(hp), QE_SKEW_ASM_K4_C32_HM(hm)      \
                 : QE_SKEW_ASM_K4_C32_CLOBBERS)

#define SKEW2_BODY(TEXT)                                                                                                            \
    asm volatile(TEXT                                                                                                              \
                 : "+" QE_SKEW2_K4_P0(P[0]), "+" QE_SKEW2_K4_P1(P[1]), "+" QE_SKEW2_K4_P2(P[2]), "+" QE_SKEW2_K4_P3(P[3]),         \
                   "+" QE_SKEW2_K4_M0(M[0]), "+" QE_SKEW2_K4_M1(M[1]), "+" QE_SKEW2_K4_M2(M[2]), "+" QE_SKEW2_K4_M3(M[3]),         \
                   "=" QE_SKEW2_K4_GP(gP), "=" QE_SKEW2_K4_GM(gM), "+" QE_SKEW2_K4_T0(t0w), "+" QE_SKEW2_K4_T1(t1w),               \
                   "+" QE_SKEW2_K4_HP(hpw), "+" QE_SKEW2_K4_HM(hmw)                                                                \
                 : QE_SKEW2_K4_A0(a[0]), QE_SKEW2_K4_A1(a[1]), QE_SKEW2_K4_A2(a[2]), QE_SKEW2_K4_A3(a[3]),                         \
                   QE_SKEW2_K4_B0(b[0]), QE_SKEW2_K4_B1(b[1]), QE_SKEW2_K4_B2(b[2]), QE_SKEW2_K4_B3(b[3])                          \
                 : QE_SKEW2_K4_CLOBBERS)

template <int V>
__device__ __forceinline__ void skew4_asm(u64 (&P)[4], u64 (&M)[4], const u64 (&a)[4], const u64 (&b)[4], u64 T0, u64 T1, u64 hinP, u64 hinM,
                                          u64& houtP, u64& houtM) {
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? qe::hi32(T0) : qe::lo32(T0), t1 = half ? qe::hi32(T1) : qe::lo32(T1);
        const u32 hp = half ? qe::hi32(hinP) : qe::lo32(hinP), hm = half ? qe::hi32(hinM) : qe::lo32(hinM);
        u32 gP = 0, gM = 0;
        if (V == 0) SKEW_ASM_BODY(QE_SKEW_ASM_K4_C32_TEXT);
        else if (V == 1) SKEW_ASM_BODY(QE_SKEW_ASM_K4_C32_PROG_TEXT);
        else if (V == 2) SKEW_ASM_BODY(QE_SKEW_ASM_K4_C32_D32_TEXT);
        else if (V == 3) SKEW_ASM_BODY(QE_SKEW_ASM_K4_C32_VOP3_TEXT);
        else if (V == 4) SKEW_ASM_BODY_LOW(QE_SKEW_ASM_K4_C32_LOW_TEXT);
        else if (V >= 6) {
            u32 t0w = t0, t1w = t1, hpw = hp, hmw = hm;       // the pass consumes its text / carry words (running bit-reversed copies)
            if (V == 6) SKEW2_BODY(QE_SKEW2_K4_TEXT);
            else if (V == 7) SKEW2_BODY(QE_SKEW2_K4_PROG_TEXT);
            else SKEW2_BODY(QE_SKEW2_K4_VOP3_TEXT);
        } else {
            u32 t0 = half ? qe::hi32(T0) : qe::lo32(T0), t1 = half ? qe::hi32(T1) : qe::lo32(T1);
            u32 hp = half ? qe::hi32(hinP) : qe::lo32(hinP), hm = half ? qe::hi32(hinM) : qe::lo32(hinM);
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                SKEW_ASM_BODY_ACC(QE_SKEW_ASM_K4_C8_ACC_TEXT);
                t0 >>= 8; t1 >>= 8; hp >>= 8; hm >>= 8;
            }
        }
        const u32 rP = __builtin_bitreverse32(gP), rM = __builtin_bitreverse32(gM);
        if (half) { o[1] = rP; o[3] = rM; } else { o[0] = rP; o[2] = rM; }
    }
    houtP = qe::mk64(o[0], o[1]);
    houtM = qe::mk64(o[2], o[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// The block step on 32-ROW blocks, full-rate instructions only (profiles/r06_b_issue_classes.md: a quarter-rate instruction
// costs ~6.8 cycles in a mixed stream at two waves per SIMD, a full-rate one 2.1-2.6).  The Myers step is exact for any
// partition of a column into blocks (k_windowed_quad relies on it): a 64-row slot is two 32-row steps, the carry between
// them the step's own PHout / MHout.  sum = v_add_u32, x << 1 = x + x (an asm v_add_u32: hipcc would select the quarter-rate
// v_lshlrev_b32), the carry-in ORed into the consumers' truth tables.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 dbl(u32 x) { u32 r; asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ void step32(u32 e, u32& P, u32& M, u32 cP, u32 cM, u32& oP, u32& oM) {
    const u32 xv = e | M;
    const u32 ecl = e | cM;
    const u32 t = ecl & P, q = ecl | P;
    const u32 s = t + P;
    const u32 ph = qe::bitop3<0xF1>(M, s, q);                 // M | ~(s | q)
    const u32 mh = qe::bitop3<0xB0>(P, s, ecl);               // P & ((s ^ P) | Eqc)
    oP = ph >> 31; oM = mh >> 31;
    const u32 ph2 = dbl(ph), mh2 = dbl(mh);
    const u32 xvc = xv | cP;
    const u32 w = qe::bitop3<0xF1>(cM, xvc, ph2);             // MHin | ~(Xv | Phs)
    P = mh2 | w;
    M = qe::bitop3<0xA8>(ph2, cP, xv);                        // (ph2 | PHin) & Xv
}
template <int K>
__device__ __forceinline__ void run64_skew32(u64 (&P)[K], u64 (&M)[K], const u64 (&a)[K], const u64 (&b)[K],
                                             u64 T0, u64 T1, u64 hinP, u64 hinM, u64& houtP, u64& houtM) {
    using namespace qe;
    u32 alo[K], ahi[K], blo[K], bhi[K], Plo[K], Phi[K], Mlo[K], Mhi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        alo[k] = lo32(a[k]); ahi[k] = hi32(a[k]); blo[k] = lo32(b[k]); bhi[k] = hi32(b[k]);
        Plo[k] = lo32(P[k]); Phi[k] = hi32(P[k]); Mlo[k] = lo32(M[k]); Mhi[k] = hi32(M[k]);
    }
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // running bit-reversed words: bit 31 is the current column's
        u32 t0 = __builtin_bitreverse32(half ? hi32(T0) : lo32(T0)), t1 = __builtin_bitreverse32(half ? hi32(T1) : lo32(T1));
        u32 hp = __builtin_bitreverse32(half ? hi32(hinP) : lo32(hinP)), hm = __builtin_bitreverse32(half ? hi32(hinM) : lo32(hinM));
        u32 gP = 0, gM = 0;
        u32 m0[32], m1[32], cP[K], cM[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { cP[k] = 0; cM[k] = 0; }
#pragma unroll
        for (int s = 0; s < 32 + K - 1; ++s) {
            if (s < 32) {
                m0[s] = (u32)((int)t0 >> 31); m1[s] = (u32)((int)t1 >> 31);
                t0 = dbl(t0); t1 = dbl(t1);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const int c = s - k;
                if (c < 0 || c >= 32) continue;
                u32 inP, inM;
                if (k == 0) { inP = hp >> 31; inM = hm >> 31; hp = dbl(hp); hm = dbl(hm); }
                else { inP = cP[k]; inM = cM[k]; }
                const u32 elo = bitop3<0x90>(~(alo[k] ^ m0[c]), blo[k], m1[c]), ehi = bitop3<0x90>(~(ahi[k] ^ m0[c]), bhi[k], m1[c]);
                u32 midP, midM, outP, outM;
                step32(elo, Plo[k], Mlo[k], inP, inM, midP, midM);
                step32(ehi, Phi[k], Mhi[k], midP, midM, outP, outM);
                if (k + 1 < K) { cP[k + 1] = outP; cM[k + 1] = outM; }
                else { gP = dbl(gP) | outP; gM = dbl(gM) | outM; }
            }
        }
        const u32 rP = __builtin_bitreverse32(gP), rM = __builtin_bitreverse32(gM);
        if (half) { o[1] = rP; o[3] = rM; } else { o[0] = rP; o[2] = rM; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { P[k] = mk64(Plo[k], Phi[k]); M[k] = mk64(Mlo[k], Mhi[k]); }
    houtP = qe::mk64(o[0], o[1]);
    houtM = qe::mk64(o[2], o[3]);
}

struct Stamp { uint64_t cyc, real; };

// VAR 0: run64_skew<4> (hipcc's schedule), 1: asm scheduled dmin 8, 2: asm in program order, 3: asm scheduled dmin 32
template <int VAR>
__global__ __launch_bounds__(256) void k_step(u32* out, Stamp* stamps, int iters) {
    using namespace qe;
    extern __shared__ uint4 pin[];
    u64 P[4], M[4], a[4], b[4];
    const u64 seed = (u64)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        P[k] = ~(u64)0; M[k] = 0;
        a[k] = seed * (2 * k + 3) ^ (seed >> 17); b[k] = seed * (2 * k + 5) ^ (seed >> 13);
    }
    u64 T0 = seed ^ 0x0123456789abcdefull, T1 = seed * 7 + 1, hinP = ~(u64)0, hinM = 0, houtP = 0, houtM = 0;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (VAR == 0) run64_skew<4>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
        else if (VAR == 10) run64_skew32<4>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
        else skew4_asm<VAR - 1>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
        T0 = T0 * 6364136223846793005ull + 1442695040888963407ull;
        T1 ^= T0 >> 7;
        hinP = houtM | T1; hinM = houtP & ~hinP;
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    u64 acc = houtP ^ houtM;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= P[k] ^ M[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};
}

template <int V>
__global__ void k_verify(u32* bad) {
    using namespace qe;
    u64 s = (u64)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int it = 0; it < 32; ++it) {
        u64 P1[4], M1[4], P2[4], M2[4], a[4], b[4];
        for (int k = 0; k < 4; ++k) { const u64 x = rnd(), y = rnd(); P1[k] = P2[k] = x & ~y; M1[k] = M2[k] = y & ~x; a[k] = rnd(); b[k] = rnd(); }
        const u64 T0 = rnd(), T1 = rnd(), h1 = rnd(), h2 = rnd(), hinP = h1 & ~h2, hinM = h2 & ~h1;
        u64 o1P, o1M, o2P, o2M;
        run64_multi<4>(P1, M1, a, b, T0, T1, hinP, hinM, o1P, o1M);
        if (V == 9) run64_skew32<4>(P2, M2, a, b, T0, T1, hinP, hinM, o2P, o2M);
        else skew4_asm<(V == 9 ? 0 : V)>(P2, M2, a, b, T0, T1, hinP, hinM, o2P, o2M);
        bool ok = o1P == o2P && o1M == o2M;
        for (int k = 0; k < 4; ++k) ok = ok && P1[k] == P2[k] && M1[k] == M2[k];
        if (!ok) atomicAdd(bad, 1u);
    }
}

static u32* g_out; static Stamp* g_stamps;
static const int CUS = 256;
struct Result { double cyc_wave, clock_ghz, wall_ms; };
template <typename F> static Result measure(F launch, int wps) {
    const int blocks = CUS * wps;
    const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)255;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, lds, true);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, lds, false);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st((size_t)blocks * 4);
    hipMemcpy(st.data(), g_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto& s : st) { cyc.push_back((double)s.cyc); clk.push_back(s.real ? (double)s.cyc / (double)s.real * 0.1 : 0.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return Result{cyc[cyc.size() / 2], clk[clk.size() / 2], ms};
}
template <int VAR> static void step_row(const char* name, int wps, int iters) {
    auto launch = [&](int blocks, size_t lds, bool warm) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_step<VAR>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, warm ? 4 : iters);
    };
    const Result r = measure(launch, wps);
    const double bc = (double)iters * 64 * 4;
    const double per_simd = r.wall_ms * 1e-3 * r.clock_ghz * 1e9 / (bc * wps);
    const double rate = (double)CUS * 4 * wps * 64 * bc / (r.wall_ms * 1e-3);
    printf("%-34s w=%d  %6.1f cyc / block-column / SIMD (one wave: %6.1f)   chip %.3e block-columns/s   clock %.2f GHz  (%.2f ms)\n", name, wps,
           per_simd, r.cyc_wave / bc, rate, r.clock_ghz, r.wall_ms);
}

int main() {
    hipMalloc(&g_out, (size_t)CUS * 8 * 256 * 4);
    hipMalloc(&g_stamps, (size_t)CUS * 8 * 4 * sizeof(Stamp));
    hipMemset(g_out, 0, 64);
    hipLaunchKernelGGL((k_verify<0>), dim3(256), dim3(256), 0, 0, g_out);
    hipLaunchKernelGGL((k_verify<1>), dim3(256), dim3(256), 0, 0, g_out + 1);
    hipLaunchKernelGGL((k_verify<2>), dim3(256), dim3(256), 0, 0, g_out + 2);
    hipLaunchKernelGGL((k_verify<3>), dim3(256), dim3(256), 0, 0, g_out + 3);
    hipLaunchKernelGGL((k_verify<4>), dim3(256), dim3(256), 0, 0, g_out + 4);
    hipLaunchKernelGGL((k_verify<5>), dim3(256), dim3(256), 0, 0, g_out + 5);
    hipLaunchKernelGGL((k_verify<6>), dim3(256), dim3(256), 0, 0, g_out + 6);
    hipLaunchKernelGGL((k_verify<7>), dim3(256), dim3(256), 0, 0, g_out + 7);
    hipLaunchKernelGGL((k_verify<8>), dim3(256), dim3(256), 0, 0, g_out + 8);
    hipLaunchKernelGGL((k_verify<9>), dim3(256), dim3(256), 0, 0, g_out + 9);
    u32 bad[10] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    hipMemcpy(bad, g_out, 40, hipMemcpyDeviceToHost);
    printf("32-row blocks, full-rate instructions only (run64_skew32<4>, hipcc-scheduled) against run64_multi<4>: %u mismatches\n", bad[9]);
    printf("no-quarter-rate passes (scheduled / program order / all VOP3) against run64_multi<4>: %u / %u / %u mismatches\n", bad[6], bad[7], bad[8]);
    printf("asm passes (dmin 8 / program order / dmin 32 / all VOP3 / low registers / 8 columns rolled) against run64_multi<4> on %d random passes each: %u / %u / %u / %u / %u / %u mismatches\n",
           256 * 256 * 32, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    const int it = 400;
    for (int w : {1, 2}) {
        step_row<0>("run64_skew<4> (hipcc schedule)", w, it);
        step_row<2>("asm, program order", w, it);
        step_row<1>("asm, list-scheduled dmin 8", w, it);
        step_row<3>("asm, list-scheduled dmin 32", w, it);
        step_row<4>("asm, dmin 8, all VOP3 encodings", w, it);
        step_row<5>("asm, dmin 8, registers v16..v139", w, it);
        step_row<6>("asm, dmin 8, 8 columns x 4 rolled", w, it);
        step_row<7>("asm2 no quarter-rate, dmin 4", w, it);
        step_row<8>("asm2 no quarter-rate, program order", w, it);
        step_row<9>("asm2 no quarter-rate, all VOP3", w, it);
        step_row<10>("32-row blocks, full-rate only (C++)", w, it);
    }
    return 0;
}
