// valu_rate.hip -- what the VALU of one gfx950 chip really issues, measured in SHADER CYCLES (s_memtime), so that the
// figures do not depend on the clock the chip happens to hold under load; the clock itself is reported next to them
// (delta s_memtime / delta s_memrealtime x 100 MHz).
//
// Part A  per-instruction issue cost: every instruction is asm volatile (nothing can be folded), the stream consists of
//         D independent dependency chains (D = 1, 2, 4, 8: instruction i depends on instruction i - D), and exactly
//         w waves sit on every SIMD (w = 1, 2, 4, 8; 4-wave workgroups, LDS-pinned so that w of them share a CU).
//         Reported: cycles per instruction as one wave sees it, and per SIMD (= the former / w).
// Part B  the block-step loops of qe_kernels.hip (the production code, included) on register-resident state, no memory
//         in the loop: cycles per block-column per SIMD for the 1-, 2- and 4-slot forms and for experimental variants.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate [A|B|AB]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../quicked_amd/csrc/qe_kernels.hip"

using qe::u32;
using qe::u64;

enum Op { XOR, OR, AND, ADD, SUB, LSHR, LSHL, ASHR, MOV, BITOP3, ALIGNBIT, BFE_U, BFE_I, LSHL_OR, OR3, AND_OR, ADD3, BCNT, BFREV,
          PERM, CNDMASK, LSHL_ADD_U64, ADD_CO_PAIR, LSHL_B64, NOPS };
static const char* const OP_NAME[] = {"v_xor_b32", "v_or_b32", "v_and_b32", "v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_lshlrev_b32",
                                      "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32", "v_alignbit_b32", "v_bfe_u32", "v_bfe_i32",
                                      "v_lshl_or_b32", "v_or3_b32", "v_and_or_b32", "v_add3_u32", "v_bcnt_u32_b32", "v_bfrev_b32",
                                      "v_perm_b32", "v_cndmask_b32", "v_lshl_add_u64", "v_add_co+v_addc_co", "v_lshlrev_b64"};

// one instruction of a chain: r = f(r, x, y); x and y are never written (no dependency through them)
template <int OP> __device__ __forceinline__ void op32(u32& r, u32 x, u32 y) {
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
    if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
    if (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %0" : "+v"(r));
    if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(x), "v"(y));
    if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(r) : "v"(x));
    if (OP == BFE_U) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(r));
    if (OP == BFE_I) asm volatile("v_bfe_i32 %0, %0, 3, 29" : "+v"(r));
    if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(x));
    if (OP == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
    if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
    if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
    if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r) : "v"(x));
    if (OP == BFREV) asm volatile("v_bfrev_b32 %0, %0" : "+v"(r));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x));
}
template <int OP> __device__ __forceinline__ void op64(u64& r, u64 x) {
    if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(r) : "v"(x));
    if (OP == ADD_CO_PAIR) {
        u32 lo = (u32)r, hi = (u32)(r >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"((u32)x), "v"((u32)(x >> 32)) : "vcc");
        r = ((u64)hi << 32) | lo;
    }
    if (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r));
}

struct Stamp { uint64_t cyc, real; };

template <int OP, int D>
__global__ __launch_bounds__(256) void k_rate(u32* out, Stamp* stamps, int iters) {
    extern __shared__ uint4 pin[];
    u32 r[8]; u64 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { r[j] = threadIdx.x * 2654435761u + j; q[j] = ((u64)r[j] << 32) | (r[j] ^ 0x5bd1e995u); }
    const u32 x = threadIdx.x | 1u, y = ~threadIdx.x;
    const u64 x64 = ((u64)y << 32) | x;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (OP < LSHL_ADD_U64) op32<OP>(r[u % D], x, y);
            else op64<OP>(q[u % D], x64);
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    u32 acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= r[j] ^ (u32)q[j] ^ (u32)(q[j] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};
}

static u32* g_out; static Stamp* g_stamps;
static const int CUS = 256;

struct Result { double cyc_wave, clock_ghz, wall_ms; };
template <typename F> static Result measure(F launch, int wps) {
    const int blocks = CUS * wps;
    const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)255;      // wps of these workgroups fill a CU's LDS: no CU takes more
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

template <int OP, int D> static void rate_row(int wps, int iters) {
    auto launch = [&](int blocks, size_t lds, bool warm) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<OP, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_rate<OP, D>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, warm ? 8 : iters);
    };
    const Result r = measure(launch, wps);
    const double instr = (double)iters * 64 * (OP == ADD_CO_PAIR ? 2 : 1);
    const double per_wave = r.cyc_wave / instr;
    // per SIMD: from the wave's own cycles (exact while all w waves are resident together) and from the wall clock of the
    // whole launch at the measured clock (includes launch ramp and tail; the honest figure when w is large)
    printf("%-20s D=%d w=%d  wave %6.2f  SIMD %6.2f (wall %6.2f) cyc/instr   clock %.2f GHz  (%.2f ms)\n", OP_NAME[OP], D, wps, per_wave,
           per_wave / wps, r.wall_ms * 1e-3 * r.clock_ghz * 1e9 / (instr * wps), r.clock_ghz, r.wall_ms);
}
template <int OP> static void rate_op(bool full) {
    const int iters = 3000;
    if (full) {
        rate_row<OP, 1>(1, iters); rate_row<OP, 2>(1, iters); rate_row<OP, 4>(1, iters); rate_row<OP, 8>(1, iters);
        rate_row<OP, 1>(2, iters); rate_row<OP, 2>(2, iters); rate_row<OP, 4>(2, iters); rate_row<OP, 8>(2, iters);
        rate_row<OP, 1>(4, iters); rate_row<OP, 4>(4, iters); rate_row<OP, 8>(8, iters);
    } else {
        rate_row<OP, 1>(2, iters); rate_row<OP, 4>(2, iters); rate_row<OP, 8>(8, iters);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Part B: block-step loops on register state.  VAR 0: run64_multi<4> (production 4-slot pass), 1: run64_multi<2>,
// 2: run64_fast<0, true> (one slot), 3: run64_skew<4>, the 4-slot pass software-skewed (slot k runs k columns behind
// slot 0, so the four block steps of a pass step are independent: what slots_pass uses), 4: two slots skewed
// ---------------------------------------------------------------------------------------------------------------
using qe::run64_skew;

template <int VAR>
__global__ __launch_bounds__(256) void k_step(u32* out, Stamp* stamps, int iters) {
    using namespace qe;
    extern __shared__ uint4 pin[];
    constexpr int K = (VAR == 0 || VAR == 3) ? 4 : (VAR == 1 || VAR == 4) ? 2 : 1;
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
        if (K == 4) {
            if (VAR == 0) run64_multi<4>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
            else run64_skew<4>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
        } else if (K == 2) {
            u64 P2[2] = {P[0], P[1]}, M2[2] = {M[0], M[1]};
            const u64 a2[2] = {a[0], a[1]}, b2[2] = {b[0], b[1]};
            if (VAR == 1) run64_multi<2>(P2, M2, a2, b2, T0, T1, hinP, hinM, houtP, houtM);
            else run64_skew<2>(P2, M2, a2, b2, T0, T1, hinP, hinM, houtP, houtM);
            P[0] = P2[0]; P[1] = P2[1]; M[0] = M2[0]; M[1] = M2[1];
        } else {
            run64_fast<0, true>(P[0], M[0], a[0], b[0], T0, T1, hinP, hinM, houtP, houtM, true, nullptr, 0, nullptr);
        }
        T0 = T0 * 6364136223846793005ull + 1442695040888963407ull;      // the next chunk's text
        T1 ^= T0 >> 7;
        hinP = houtM | T1; hinM = houtP & ~hinP;                            // any carry words (never both bits set)
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    u64 acc = houtP ^ houtM;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= P[k] ^ M[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};
}

// run64_skew against run64_multi on random states, text and carry words: every output word must be identical
template <int K>
__global__ void k_verify(u32* bad) {
    using namespace qe;
    u64 s = (u64)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int it = 0; it < 64; ++it) {
        u64 P1[K], M1[K], P2[K], M2[K], a[K], b[K];
        for (int k = 0; k < K; ++k) { const u64 x = rnd(), y = rnd(); P1[k] = P2[k] = x & ~y; M1[k] = M2[k] = y & ~x; a[k] = rnd(); b[k] = rnd(); }
        const u64 T0 = rnd(), T1 = rnd(), h1 = rnd(), h2 = rnd(), hinP = h1 & ~h2, hinM = h2 & ~h1;
        u64 o1P, o1M, o2P, o2M;
        run64_multi<K>(P1, M1, a, b, T0, T1, hinP, hinM, o1P, o1M);
        run64_skew<K>(P2, M2, a, b, T0, T1, hinP, hinM, o2P, o2M);
        bool ok = o1P == o2P && o1M == o2M;
        for (int k = 0; k < K; ++k) ok = ok && P1[k] == P2[k] && M1[k] == M2[k];
        if (!ok) atomicAdd(bad, 1u);
    }
}

template <int VAR> static void step_row(const char* name, int K, int wps, int iters) {
    auto launch = [&](int blocks, size_t lds, bool warm) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_step<VAR>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, warm ? 4 : iters);
    };
    const Result r = measure(launch, wps);
    const double bc = (double)iters * 64 * K;                          // block-columns per wave
    const double per_simd = r.wall_ms * 1e-3 * r.clock_ghz * 1e9 / (bc * wps);   // from the launch's wall clock at the measured clock
    const double rate = (double)CUS * 4 * wps * 64 * bc / (r.wall_ms * 1e-3);   // lane block-columns per second, whole chip
    printf("%-26s w=%d  %6.1f cyc / block-column / SIMD (one wave: %6.1f)   chip %.3e block-columns/s   clock %.2f GHz  (%.2f ms)\n", name, wps,
           per_simd, r.cyc_wave / bc, rate, r.clock_ghz, r.wall_ms);
}

// ---------------------------------------------------------------------------------------------------------------
// Part C: does the VGPR bank of the three sources of a v_bitop3 matter?  Eight independent chains on physical
// registers; SAME: the three sources of every instruction are congruent mod 4 (v40 v44 v48 ...), DIFF: they differ.
// ---------------------------------------------------------------------------------------------------------------
#define BK_SAME(d, a, b) "v_bitop3_b32 v" #d ", v" #d ", v" #a ", v" #b " bitop3:0x96\n\t"
template <int MODE>
__global__ __launch_bounds__(256) void k_bank(u32* out, Stamp* stamps, int iters) {
    extern __shared__ uint4 pin[];
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // dst/src0 v40..v47, src1 / src2 in the SAME bank as src0
            asm volatile(BK_SAME(40, 48, 56) BK_SAME(41, 49, 57) BK_SAME(42, 50, 58) BK_SAME(43, 51, 59)
                         BK_SAME(44, 52, 60) BK_SAME(45, 53, 61) BK_SAME(46, 54, 62) BK_SAME(47, 55, 63)
                         BK_SAME(40, 52, 60) BK_SAME(41, 53, 61) BK_SAME(42, 54, 62) BK_SAME(43, 55, 63)
                         BK_SAME(44, 48, 56) BK_SAME(45, 49, 57) BK_SAME(46, 50, 58) BK_SAME(47, 51, 59)
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                             "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else if (MODE == 1) {   // src1 / src2 in two OTHER banks
            asm volatile(BK_SAME(40, 49, 58) BK_SAME(41, 50, 59) BK_SAME(42, 51, 56) BK_SAME(43, 48, 57)
                         BK_SAME(44, 53, 62) BK_SAME(45, 54, 63) BK_SAME(46, 55, 60) BK_SAME(47, 52, 61)
                         BK_SAME(40, 53, 62) BK_SAME(41, 54, 63) BK_SAME(42, 55, 60) BK_SAME(43, 52, 61)
                         BK_SAME(44, 49, 58) BK_SAME(45, 50, 59) BK_SAME(46, 51, 56) BK_SAME(47, 48, 57)
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                             "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
        } else {                  // two sources only (v_xor_b32 e32 encoding), same bank / -- reference
            asm volatile("v_xor_b32 v40, v40, v48\n\tv_xor_b32 v41, v41, v49\n\tv_xor_b32 v42, v42, v50\n\tv_xor_b32 v43, v43, v51\n\t"
                         "v_xor_b32 v44, v44, v52\n\tv_xor_b32 v45, v45, v53\n\tv_xor_b32 v46, v46, v54\n\tv_xor_b32 v47, v47, v55\n\t"
                         "v_xor_b32 v40, v40, v52\n\tv_xor_b32 v41, v41, v53\n\tv_xor_b32 v42, v42, v54\n\tv_xor_b32 v43, v43, v55\n\t"
                         "v_xor_b32 v44, v44, v48\n\tv_xor_b32 v45, v45, v49\n\tv_xor_b32 v46, v46, v50\n\tv_xor_b32 v47, v47, v51\n\t"
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)c1;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, t1 - t0};
}
template <int MODE> static void bank_row(const char* name, int wps) {
    const int iters = 12000;
    auto launch = [&](int blocks, size_t lds, bool warm) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bank<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_bank<MODE>), dim3(blocks), dim3(256), lds, 0, g_out, g_stamps, warm ? 8 : iters);
    };
    const Result r = measure(launch, wps);
    const double instr = (double)iters * 16;
    printf("%-44s w=%d  wave %6.2f cyc/instr   chip-wall %6.2f cyc/instr/SIMD   clock %.2f GHz\n", name, wps, r.cyc_wave / instr,
           r.wall_ms * 1e-3 * r.clock_ghz * 1e9 / (instr * wps), r.clock_ghz);
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "ABC";
    hipMalloc(&g_out, (size_t)CUS * 8 * 256 * 4);
    hipMalloc(&g_stamps, (size_t)CUS * 8 * 4 * sizeof(Stamp));
    if (strchr(what, 'A')) {
        printf("# Part A: cycles per wave64 instruction (shader cycles, s_memtime); D = independent chains, w = waves per SIMD\n");
        rate_op<XOR>(true); rate_op<BITOP3>(true); rate_op<LSHL_ADD_U64>(true); rate_op<ADD_CO_PAIR>(true);
        rate_op<OR>(false); rate_op<AND>(false); rate_op<ADD>(false); rate_op<SUB>(false); rate_op<LSHR>(false); rate_op<LSHL>(false);
        rate_op<ASHR>(false); rate_op<MOV>(false); rate_op<ALIGNBIT>(false); rate_op<BFE_U>(false); rate_op<BFE_I>(false);
        rate_op<LSHL_OR>(false); rate_op<OR3>(false); rate_op<AND_OR>(false); rate_op<ADD3>(false); rate_op<BCNT>(false);
        rate_op<BFREV>(false); rate_op<PERM>(false); rate_op<CNDMASK>(false); rate_op<LSHL_B64>(false);
    }
    if (strchr(what, 'C')) {
        printf("# Part C: VGPR banks of the sources (8 independent chains, physical registers)\n");
        for (int w : {1, 2, 4}) {
            bank_row<0>("v_bitop3_b32, 3 sources in ONE bank", w);
            bank_row<1>("v_bitop3_b32, 3 sources in THREE banks", w);
            bank_row<2>("v_xor_b32 (2 sources, 32-bit encoding)", w);
        }
    }
    if (strchr(what, 'B')) {
        printf("# Part B: block-step loops of qe_kernels.hip on registers (no memory in the loop)\n");
        hipMemset(g_out, 0, 8);
        hipLaunchKernelGGL((k_verify<4>), dim3(256), dim3(256), 0, 0, g_out);
        hipLaunchKernelGGL((k_verify<2>), dim3(256), dim3(256), 0, 0, g_out + 1);
        u32 bad[2] = {1, 1};
        hipMemcpy(bad, g_out, 8, hipMemcpyDeviceToHost);
        printf("run64_skew<4> / <2> against run64_multi on %d random passes each: %u / %u mismatches\n", 256 * 256 * 64, bad[0], bad[1]);
        const int it = 400;
        for (int w : {1, 2, 3, 4}) {
            step_row<0>("run64_multi<4>", 4, w, it);
            step_row<3>("run64_skew<4> (production)", 4, w, it);
            step_row<1>("run64_multi<2>", 2, w, it);
            step_row<4>("run64_skew<2> (production)", 2, w, it);
            step_row<2>("run64_fast<WIDE> (1 slot)", 1, w, it);
        }
    }
    return 0;
}
