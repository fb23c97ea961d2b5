// qe_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the QuickEd hot path.
//
// Execution model: ONE LANE PER ALIGNMENT, 64 alignments per
// wavefront, no cross-lane traffic.  The reference walks a text column at a
// time over all band blocks and carries one PHout/MHout bit from block to block
// (bpm_banded.c:232-262).  Here a lane instead walks ONE 64-row block over a
// whole 64-column chunk with the block's Pv/Mv in VGPRs, reading the 64 carry-in
// bits of the block above from a 64-bit "carry word" and producing the 64
// carry-out bits for the block below as another one.  The dependency graph is
// the same (block i, column c needs block i-1, column c and block i, column
// c-1), only the traversal order differs, so every value is bit-identical to
// the reference's; band bookkeeping then runs per lane exactly as the
// reference's does every 64 columns (bpm_banded.c:889-922 / 264-301).
//
// The 64-column inner loop is pure 32-bit integer VALU on registers (26-29
// instructions per 64 DP cells, built from gfx950's fast-issuing ops: see
// block_step_core and DESIGN.md 4.1); the score-only kernel walks four (else two)
// band slots per pass, sharing the text masks and passing carries in registers;
// global memory is touched once per (block, chunk): 16 B of state + 24 B of
// pattern planes in, 20 B out, all as [row][lane] rows.
//
// Kernels: k_pack (ASCII -> bit planes, 16-byte loads + SWAR), k_unpack_wire /
// k_reverse_planes (packed input), k_banded<false/true> (BandEd score / fill with
// checkpoints), k_banded_coop (G lanes per alignment), k_traceback (tile recompute
// in registers + straight-line path walk), k_windowed (WindowEd chain, full
// windows on chip), k_join (Hirschberg midpoint), k_format_segs / k_scan_offsets
// (CIGAR strings, three styles), k_check_segs / k_check_strings (CIGAR validator).
// No MFMA (bit manipulation, not a contraction), no CUDA-compat paths.
#include <type_traits>
#include <hip/hip_runtime.h>
#include "qe_types.h"

namespace qe {

#define QE_ONES (~(u64)0)
// One wave per group of 64 tasks.  A workgroup is 4 waves -- one per SIMD of the CU it lands on -- and
// claims enough LDS that at most two of them share a CU (launch_groups, qe_driver.hip): the number of
// waves that share a SIMD sets a lane-per-alignment kernel's duration, so it is not left to the dispatcher.
#define QE_WAVE_IN_BLOCK() ((int)(threadIdx.x >> 6))
#define QE_GROUP_INDEX() ((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)))
extern __shared__ uint4 qe_dyn_lds[];

__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }
__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// Stored-column layout of the WindowEd history (every column of the reachable blocks): 8
// columns of one (block, lane) are contiguous (one 128-byte line), lanes next to each other:
//   element(col, slot, lane) = ((col >> 3) * nslots + slot) * 512 + lane * 8 + (col & 7)     [16-byte units]
// A lane's in-window traceback walks columns one at a time, so 7 of 8 steps stay inside a
// line it already pulled; 8 consecutive column stores of a block complete every line they touch.
__device__ __forceinline__ int64_t tile_elem(int col, int slot, int nslots) {
    return ((int64_t)(col >> 3) * nslots + slot) * 512 + (col & 7);
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------
// Myers/Hyyro block step (bpm_commons.h:49-68) without the carry extraction:
// returns the pre-shift horizontal deltas so the caller can pick any row's bit.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void block_step(u64 Eq, u64& P, u64& M, u32 PHin, u32 MHin, u64& Ph_raw, u64& Mh_raw) {
    const u64 Xv = Eq | M;
    const u64 Eqc = Eq | (u64)MHin;
    const u64 Xh = (((Eqc & P) + P) ^ P) | Eqc;
    u64 Ph = M | ~(Xh | P);
    u64 Mh = P & Xh;
    Ph_raw = Ph;
    Mh_raw = Mh;
    Ph = (Ph << 1) | (u64)PHin;
    Mh = (Mh << 1) | (u64)MHin;
    P = Mh | ~(Xv | Ph);
    M = Ph & Xv;
}

// The same step on explicit 32-bit halves with gfx950's 3-input bit op (v_bitop3_b32; truth
// table = f(0xF0, 0xCC, 0xAA)) -- ~31 VALU ops per column instead of ~43:
//   Xh | P = sum | P | Eqc            (since (sum ^ P) | P = sum | P)
//   Ph = M | ~(sum | P | Eqc)         Mh = P & ((sum ^ P) | Eqc)
//   Pv' = Mhs | ~(Xv | Phs)           Mv' = Phs & Xv
// Carry-out bits (bit 63 of Ph / Mh) are shifted into accP / accM MSB-first by one
// v_alignbit each; the caller bit-reverses a group of 8.
template <int TT>
__device__ __forceinline__ u32 bitop3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }

// Issue cost on gfx950 (tools/valu_rate.hip, 8 waves per SIMD): v_and/or/xor/add_u32/ashr and v_bitop3
// take ~2.5 cycles of a SIMD, v_alignbit/bfe/or3/lshl_or/lshlrev and the 64-bit v_lshl_add_u64 ~4.2,
// a v_add_co + v_addc pair 6.4.  Hence: every boolean goes through v_bitop3 (also plain 3-input ORs),
// the 64-bit sum is one v_lshl_add_u64, and the two "<< 1 | carry-in" shifts are one v_lshl_add_u64
// each ((x << 1) + carry) instead of a v_lshl_or + v_alignbit pair.
__device__ __forceinline__ u64 lshl_add_u64(u64 x, u64 y) {          // x + y
    u64 r; asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(y)); return r;
}
__device__ __forceinline__ u64 shl1_add_u64(u64 x, u64 y) {          // (x << 1) + y
    u64 r; asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(y)); return r;
}

// the step proper; phhi / mhhi are the high halves of the pre-shift horizontal deltas (bit 31 = the row-63 carry out)
__device__ __forceinline__ void block_step_core(u32 elo, u32 ehi, u32& Plo, u32& Phi, u32& Mlo, u32& Mhi,
                                                u32 PHin, u32 MHin, u32& phhi, u32& mhhi) {
    const u32 xvlo = elo | Mlo, xvhi = ehi | Mhi;
    const u32 eclo = elo | MHin;
    const u64 sum = lshl_add_u64(mk64(eclo & Plo, ehi & Phi), mk64(Plo, Phi));
    const u32 slo = lo32(sum), shi = hi32(sum);
    const u32 phlo = bitop3<0xF3>(Mlo, bitop3<0xFE>(slo, Plo, eclo), 0u);   // M | ~(sum | P | Eqc)
    phhi = bitop3<0xF3>(Mhi, bitop3<0xFE>(shi, Phi, ehi), 0u);
    const u32 mhlo = bitop3<0xB0>(Plo, slo, eclo);                     // P & ((sum ^ P) | Eqc)
    mhhi = bitop3<0xB0>(Phi, shi, ehi);
    const u64 phs = shl1_add_u64(mk64(phlo, phhi), (u64)PHin);         // (Ph << 1) | PHin
    const u64 mhs = shl1_add_u64(mk64(mhlo, mhhi), (u64)MHin);
    const u32 pslo = lo32(phs), pshi = hi32(phs), mslo = lo32(mhs), mshi = hi32(mhs);
    Plo = bitop3<0xF1>(mslo, xvlo, pslo);                              // Mhs | ~(Xv | Phs)
    Phi = bitop3<0xF1>(mshi, xvhi, pshi);
    Mlo = pslo & xvlo;
    Mhi = pshi & xvhi;
}

__device__ __forceinline__ void block_step_fused(u32 elo, u32 ehi, u32& Plo, u32& Phi, u32& Mlo, u32& Mhi,
                                                 u32 PHin, u32 MHin, u32& accP, u32& accM) {
    u32 phhi, mhhi;
    block_step_core(elo, ehi, Plo, Phi, Mlo, Mhi, PHin, MHin, phhi, mhhi);
    accP = __builtin_amdgcn_alignbit(accP, phhi, 31);                  // (accP << 1) | (Ph >> 63)
    accM = __builtin_amdgcn_alignbit(accM, mhhi, 31);
}

// block_step_fused for a pass that keeps the collected carries: the collection is tied to the block's own next step (an
// empty asm that "needs" accP / accM to hand Plo on).  Left to itself the scheduler parks the 2 x 32 deltas of an unrolled
// pass in registers and runs the v_alignbit chain at the end of it (k_windowed_cp's K = 3 pass: 256 VGPRs + 772 B of
// scratch without the tie, 211 VGPRs with it).
__device__ __forceinline__ void block_step_collect(u32 elo, u32 ehi, u32& Plo, u32& Phi, u32& Mlo, u32& Mhi,
                                                   u32 PHin, u32 MHin, u32& accP, u32& accM) {
    block_step_fused(elo, ehi, Plo, Phi, Mlo, Mhi, PHin, MHin, accP, accM);
    asm("" : "+v"(Plo) : "v"(accP), "v"(accM));
}

// ---------------------------------------------------------------------------
// K vertically adjacent blocks (band slots i .. i+K-1) over 64 columns in one pass: the text masks are extracted
// once per column, every block below the first takes its carry-in straight from the horizontal deltas of the block
// above (one v_lshrrev each, no carry word), and only the lowest block's 64 carry-outs are collected -- the scores
// of the rows above follow from the cell identity (slots_pass).  Same arithmetic per cell as run64_fast, hence the
// same bits.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_planes(const u64* __restrict__ base, int bit, u64& a, u64& b, u64& nn);

template <int K>
__device__ __forceinline__ void run64_multi(u64 (&P)[K], u64 (&M)[K], const u64 (&a)[K], const u64 (&b)[K],
                                            u64 T0, u64 T1, u64 hinP, u64 hinM, u64& houtP, u64& houtM) {
    u32 alo[K], ahi[K], blo[K], bhi[K], Plo[K], Phi[K], Mlo[K], Mhi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        alo[k] = lo32(a[k]); ahi[k] = hi32(a[k]); blo[k] = lo32(b[k]); bhi[k] = hi32(b[k]);
        Plo[k] = lo32(P[k]); Phi[k] = hi32(P[k]); Mlo[k] = lo32(M[k]); Mhi[k] = hi32(M[k]);
    }
    u32 oPlo = 0, oPhi = 0, oMlo = 0, oMhi = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
        const u32 hp = half ? hi32(hinP) : lo32(hinP), hm = half ? hi32(hinM) : lo32(hinM);
        u32 gP = 0, gM = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0, c, 1);
            const u32 m1 = (u32)__builtin_amdgcn_sbfe((int)t1, c, 1);
            u32 cP = __builtin_amdgcn_ubfe(hp, c, 1), cM = __builtin_amdgcn_ubfe(hm, c, 1);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const u32 elo = bitop3<0x90>(~(alo[k] ^ m0), blo[k], m1), ehi = bitop3<0x90>(~(ahi[k] ^ m0), bhi[k], m1);
                if (k + 1 < K) {
                    u32 phhi, mhhi;
                    block_step_core(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], cP, cM, phhi, mhhi);
                    cP = phhi >> 31; cM = mhhi >> 31;
                } else {
                    block_step_collect(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], cP, cM, gP, gM);
                }
            }
        }
        const u32 rP = __builtin_bitreverse32(gP), rM = __builtin_bitreverse32(gM);
        if (half) { oPhi = rP; oMhi = rM; } else { oPlo = rP; oMlo = rM; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { P[k] = mk64(Plo[k], Phi[k]); M[k] = mk64(Mlo[k], Mhi[k]); }
    houtP = mk64(oPlo, oPhi);
    houtM = mk64(oMlo, oMhi);
}

// The same pass with the K slots SKEWED by one column each: at pass step s slot k works on column s - k, so the K block
// steps of a pass step do not depend on one another (the carry slot k needs was produced one pass step earlier) and the
// in-order wave finds independent work between the dependent instructions of one block step (an issue slot of a
// lone dependent chain costs 8 cycles instead of 4: tools/valu_rate.hip, Part A).  Bit-identical to run64_multi
// (4 M random passes, valu_rate Part B); 2-3 % faster at every occupancy.
template <int K>
__device__ __forceinline__ void run64_skew(u64 (&P)[K], u64 (&M)[K], const u64 (&a)[K], const u64 (&b)[K],
                                           u64 T0, u64 T1, u64 hinP, u64 hinM, u64& houtP, u64& houtM) {
    u32 alo[K], ahi[K], blo[K], bhi[K], Plo[K], Phi[K], Mlo[K], Mhi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        alo[k] = lo32(a[k]); ahi[k] = hi32(a[k]); blo[k] = lo32(b[k]); bhi[k] = hi32(b[k]);
        Plo[k] = lo32(P[k]); Phi[k] = hi32(P[k]); Mlo[k] = lo32(M[k]); Mhi[k] = hi32(M[k]);
    }
    u32 oPlo = 0, oPhi = 0, oMlo = 0, oMhi = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
        const u32 hp = half ? hi32(hinP) : lo32(hinP), hm = half ? hi32(hinM) : lo32(hinM);
        u32 gP = 0, gM = 0;
        u32 m0[32], m1[32], cP[K], cM[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { cP[k] = 0; cM[k] = 0; }
#pragma unroll
        for (int s = 0; s < 32 + K - 1; ++s) {
            if (s < 32) {
                m0[s] = (u32)__builtin_amdgcn_sbfe((int)t0, s, 1);
                m1[s] = (u32)__builtin_amdgcn_sbfe((int)t1, s, 1);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {           // lower slots first: they consume the carries of the previous step
                const int c = s - k;
                if (c < 0 || c >= 32) continue;
                const u32 elo = bitop3<0x90>(~(alo[k] ^ m0[c]), blo[k], m1[c]), ehi = bitop3<0x90>(~(ahi[k] ^ m0[c]), bhi[k], m1[c]);
                u32 inP, inM;
                if (k == 0) { inP = __builtin_amdgcn_ubfe(hp, c, 1); inM = __builtin_amdgcn_ubfe(hm, c, 1); }
                else { inP = cP[k]; inM = cM[k]; }
                if (k + 1 < K) {
                    u32 phhi, mhhi;
                    block_step_core(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], inP, inM, phhi, mhhi);
                    cP[k + 1] = phhi >> 31; cM[k + 1] = mhhi >> 31;
                } else {
                    block_step_collect(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], inP, inM, gP, gM);
                }
            }
        }
        const u32 rP = __builtin_bitreverse32(gP), rM = __builtin_bitreverse32(gM);
        if (half) { oPhi = rP; oMhi = rM; } else { oPlo = rP; oMlo = rM; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { P[k] = mk64(Plo[k], Phi[k]); M[k] = mk64(Mlo[k], Mhi[k]); }
    houtP = mk64(oPlo, oPhi);
    houtM = mk64(oMlo, oMhi);
}

// run64_skew for a WindowEd window (general W): every slot also leaves what the in-window traceback needs to recompute
// any 8-column tile of it -- {Pv, Mv} BEFORE every 8th column (st[group * gstride + k * 64], where keep[k]) and the 64
// carry-outs of EVERY slot (oP / oM[k]: slot k + 1's carry-in word).  Same arithmetic per cell as run64_fast.
template <int K>
__device__ __forceinline__ void run64_skew_cp(u64 (&P)[K], u64 (&M)[K], const u64 (&a)[K], const u64 (&b)[K],
                                              u64 T0, u64 T1, u64 hinP, u64 hinM, u64 (&oP)[K], u64 (&oM)[K],
                                              const bool (&keep)[K], uint4* st, int64_t gstride) {
    u32 alo[K], ahi[K], blo[K], bhi[K], Plo[K], Phi[K], Mlo[K], Mhi[K];
    u32 oPlo[K], oPhi[K], oMlo[K], oMhi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        alo[k] = lo32(a[k]); ahi[k] = hi32(a[k]); blo[k] = lo32(b[k]); bhi[k] = hi32(b[k]);
        Plo[k] = lo32(P[k]); Phi[k] = hi32(P[k]); Mlo[k] = lo32(M[k]); Mhi[k] = hi32(M[k]);
        oPlo[k] = 0; oPhi[k] = 0; oMlo[k] = 0; oMhi[k] = 0;
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
        const u32 hp = half ? hi32(hinP) : lo32(hinP), hm = half ? hi32(hinM) : lo32(hinM);
        u32 gP[K], gM[K], cP[K], cM[K];
        u32 m0[32], m1[32];
#pragma unroll
        for (int k = 0; k < K; ++k) { gP[k] = 0; gM[k] = 0; cP[k] = 0; cM[k] = 0; }
#pragma unroll
        for (int s = 0; s < 32 + K - 1; ++s) {
            if (s < 32) {
                m0[s] = (u32)__builtin_amdgcn_sbfe((int)t0, s, 1);
                m1[s] = (u32)__builtin_amdgcn_sbfe((int)t1, s, 1);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {           // lower slots first: they consume the carries of the previous step
                const int c = s - k;
                if (c < 0 || c >= 32) continue;
                if ((c & 7) == 0) {
                    if (keep[k]) st[(int64_t)(4 * half + (c >> 3)) * gstride + k * 64] = make_uint4(Plo[k], Phi[k], Mlo[k], Mhi[k]);
                }
                const u32 elo = bitop3<0x90>(~(alo[k] ^ m0[c]), blo[k], m1[c]), ehi = bitop3<0x90>(~(ahi[k] ^ m0[c]), bhi[k], m1[c]);
                u32 inP, inM;
                if (k == 0) { inP = __builtin_amdgcn_ubfe(hp, c, 1); inM = __builtin_amdgcn_ubfe(hm, c, 1); }
                else { inP = cP[k]; inM = cM[k]; }
                u32 phhi, mhhi;
                block_step_core(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], inP, inM, phhi, mhhi);
                gP[k] = __builtin_amdgcn_alignbit(gP[k], phhi, 31);
                gM[k] = __builtin_amdgcn_alignbit(gM[k], mhhi, 31);
                // the slot below takes its carry-in from the collected word: that keeps the collection on the critical path
                // (left to itself the scheduler parks all 32 x 2 deltas of every slot in registers until the end of the pass)
                if (k + 1 < K) { cP[k + 1] = gP[k] & 1u; cM[k + 1] = gM[k] & 1u; }
                else asm("" : "+v"(Plo[k]) : "v"(gP[k]), "v"(gM[k]));      // the lowest slot's collection: pinned to its own next step
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const u32 rP = __builtin_bitreverse32(gP[k]), rM = __builtin_bitreverse32(gM[k]);
            if (half) { oPhi[k] = rP; oMhi[k] = rM; } else { oPlo[k] = rP; oMlo[k] = rM; }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        P[k] = mk64(Plo[k], Phi[k]); M[k] = mk64(Mlo[k], Mhi[k]);
        oP[k] = mk64(oPlo[k], oPhi[k]); oM[k] = mk64(oMlo[k], oMhi[k]);
    }
}

// run64_skew for the BandEd fill: K adjacent band slots of one chunk, each lane taking part with the slots inside ITS band
// (tight QuickEd bands start and end at a different slot in every lane).  top[k] = 1 where slot k is the lane's topmost
// band slot: its carry-in is (1, 0) whatever the slot above computed (bpm_banded.c:236-238).  Every slot leaves its
// checkpoints {Pv, Mv} AFTER columns 15, 31, 47, 63 (run64_fast's STORE == 2 layout: st + k * 64 is slot k's checkpoint 0
// of this chunk, column 63 goes to the next chunk's numbering at stl) and collects its 64 carry-outs (oP / oM[k]).
template <int K>
__device__ __forceinline__ void run64_skew_fill(u64 (&P)[K], u64 (&M)[K], const u64 (&a)[K], const u64 (&b)[K],
                                                u64 T0, u64 T1, u64 hinP, u64 hinM, u64 (&oP)[K], u64 (&oM)[K],
                                                const bool (&act)[K], const u32 (&top)[K],
                                                uint4* st, int64_t st_stride, uint4* stl_first, uint4* stl) {
    static_assert(QE_CP_COLS == 16, "checkpoint columns are literal here");
    u32 alo[K], ahi[K], blo[K], bhi[K], Plo[K], Phi[K], Mlo[K], Mhi[K];
    u32 oPlo[K], oPhi[K], oMlo[K], oMhi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        alo[k] = lo32(a[k]); ahi[k] = hi32(a[k]); blo[k] = lo32(b[k]); bhi[k] = hi32(b[k]);
        Plo[k] = lo32(P[k]); Phi[k] = hi32(P[k]); Mlo[k] = lo32(M[k]); Mhi[k] = hi32(M[k]);
        oPlo[k] = 0; oPhi[k] = 0; oMlo[k] = 0; oMhi[k] = 0;
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
        const u32 hp = half ? hi32(hinP) : lo32(hinP), hm = half ? hi32(hinM) : lo32(hinM);
        u32 gP[K], gM[K];
        u32 m0[32], m1[32];
#pragma unroll
        for (int k = 0; k < K; ++k) { gP[k] = 0; gM[k] = 0; }
#pragma unroll
        for (int s = 0; s < 32 + K - 1; ++s) {
            if (s < 32) {
                m0[s] = (u32)__builtin_amdgcn_sbfe((int)t0, s, 1);
                m1[s] = (u32)__builtin_amdgcn_sbfe((int)t1, s, 1);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {           // lower slots first: they consume the carries of the previous step
                const int c = s - k;
                if (c < 0 || c >= 32) continue;
                const u32 elo = bitop3<0x90>(~(alo[k] ^ m0[c]), blo[k], m1[c]), ehi = bitop3<0x90>(~(ahi[k] ^ m0[c]), bhi[k], m1[c]);
                u32 inP, inM;
                if (k == 0) { inP = __builtin_amdgcn_ubfe(hp, c, 1); inM = __builtin_amdgcn_ubfe(hm, c, 1); }
                else {
                    // the carry the slot above collected one step ago (taking it from the collected word keeps the
                    // collection on the critical path, see block_step_collect) -- or (1, 0) at the top of the lane's band
                    inP = (gP[k - 1] & 1u) | top[k];
                    inM = (gM[k - 1] & 1u) & ~top[k];
                }
                u32 phhi, mhhi;
                block_step_core(elo, ehi, Plo[k], Phi[k], Mlo[k], Mhi[k], inP, inM, phhi, mhhi);
                gP[k] = __builtin_amdgcn_alignbit(gP[k], phhi, 31);
                gM[k] = __builtin_amdgcn_alignbit(gM[k], mhhi, 31);
                if (k + 1 == K) asm("" : "+v"(Plo[k]) : "v"(gP[k]), "v"(gM[k]));
                if ((c & 15) == 15) {
                    if (act[k]) {
                        uint4* q = st + (int64_t)(2 * half + (c >> 4) + 1) * st_stride + k * 64;
                        if (c == 31 && half == 1) q = (k == 0) ? stl_first : stl + k * 64;
                        *q = make_uint4(Plo[k], Phi[k], Mlo[k], Mhi[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const u32 rP = __builtin_bitreverse32(gP[k]), rM = __builtin_bitreverse32(gM[k]);
            if (half) { oPhi[k] = rP; oMhi[k] = rM; } else { oPlo[k] = rP; oMlo[k] = rM; }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        P[k] = mk64(Plo[k], Phi[k]); M[k] = mk64(Mlo[k], Mhi[k]);
        oP[k] = mk64(oPlo[k], oPhi[k]); oM[k] = mk64(oMlo[k], oMhi[k]);
    }
}

// K adjacent band slots i .. i+K-1 of one chunk in one pass, with the loads, score bookkeeping and the in-place band
// shift around it.  scores[] of the lowest row from its own bottom-row deltas, as always; of every row above from
//   sum_c hout_k(c) = sum_c hin_(k+1)(c) = sum_c hout_(k+1)(c) - (v_(k+1) after - v_(k+1) before),
// v = sum of a block's vertical deltas -- exact for any block state, because every cell of the step satisfies
// v' - v = h - h_above (the step evaluates the min-recurrence cell by cell).
template <int K>
// `stride` = elements between consecutive slots / rows of one lane's state (64 in k_banded's layout, the tasks per wave in
// k_banded_coop's); scores[] is read from Srd and written to Swr (the cooperative kernel double-buffers it by chunk parity)
__device__ __forceinline__ void slots_pass(bool act, int i, int r, u64* Pv, u64* Mv, const int32_t* Srd, int32_t* Swr, int64_t stride,
                                           const u64* pp, int p0, u64 T0, u64 T1, u64& hinP, u64& hinM, u32& adv) {
    u64 P[K], M[K], a[K], b[K];
    int sc[K], v0[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        P[k] = 0; M[k] = 0; a[k] = 0; b[k] = 0; sc[k] = 0;
        if (act) {
            u64 nn;
            P[k] = Pv[(int64_t)(i + k) * stride]; M[k] = Mv[(int64_t)(i + k) * stride]; sc[k] = Srd[(int64_t)(r + k) * stride];
            load_planes(pp, p0 + 64 * (r + k), a[k], b[k], nn);
        }
        v0[k] = __popcll(P[k]) - __popcll(M[k]);
    }
    u64 houtP, houtM;
    run64_skew<K>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
    if (act) {
        int d = __popcll(houtP) - __popcll(houtM);          // sum of the bottom-row deltas of slot k, from the lowest up
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            Swr[(int64_t)(r + k) * stride] = sc[k] + d;
            d -= (__popcll(P[k]) - __popcll(M[k])) - v0[k];
            Pv[(int64_t)(i + k - 1) * stride] = P[k]; Mv[(int64_t)(i + k - 1) * stride] = M[k];   // band shift (bpm_banded.c:903-909)
        }
        adv += 64u * K;
    }
    hinP = houtP; hinM = houtM;
}

// Second 64-column chunk of an on-chip WindowEd(2,1) window: its two blocks in one skewed pass (slot 1 one column behind
// slot 0, cf. run64_skew).  Slot 0 collects its 64 carry-outs (gP / gM: the traceback recomputes block 1's tiles from
// them), slot 1 leaves {Pv, Mv} BEFORE every 8th column in LDS (st[grp * st_stride], run64_fast's STORE == 3).
// patch: the x86 SSE kernel feeds block 1's LAST column the carries of block 0's column one past the window
// (bpm_windowed.c:428-444); Eq_x is that column's Eq word for block 0.
__device__ __forceinline__ void run64_win2(u64& P0, u64& M0, u64& P1, u64& M1, u64 a0, u64 b0, u64 a1, u64 b1, u64 T0, u64 T1,
                                           u64 hinP, bool patch, u64 Eq_x, u64& gP, u64& gM, uint4* st, int st_stride) {
    const u32 a0lo = lo32(a0), a0hi = hi32(a0), b0lo = lo32(b0), b0hi = hi32(b0);
    const u32 a1lo = lo32(a1), a1hi = hi32(a1), b1lo = lo32(b1), b1hi = hi32(b1);
    u32 P0lo = lo32(P0), P0hi = hi32(P0), M0lo = lo32(M0), M0hi = hi32(M0);
    u32 P1lo = lo32(P1), P1hi = hi32(P1), M1lo = lo32(M1), M1hi = hi32(M1);
    u32 oPlo = 0, oPhi = 0, oMlo = 0, oMhi = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
        const u32 hp = half ? hi32(hinP) : lo32(hinP);
        u32 g0P = 0, g0M = 0;
        u32 m0[32], m1[32];
#pragma unroll
        for (int s = 0; s < 33; ++s) {
            if (s < 32) {
                m0[s] = (u32)__builtin_amdgcn_sbfe((int)t0, s, 1);
                m1[s] = (u32)__builtin_amdgcn_sbfe((int)t1, s, 1);
            }
            if (s >= 1) {                                     // slot 1 first: it consumes the carries of the previous step
                const int c = s - 1;
                if ((c & 7) == 0) st[(4 * half + (c >> 3)) * st_stride] = make_uint4(P1lo, P1hi, M1lo, M1hi);
                u32 inP = g0P & 1u, inM = g0M & 1u;
                if (c == 31 && half == 1 && patch) {
                    u64 Ph, Mh, Pe = mk64(P0lo, P0hi), Me = mk64(M0lo, M0hi);
                    block_step(Eq_x, Pe, Me, 1u, 0u, Ph, Mh);
                    inP = (u32)(Ph >> 63); inM = (u32)(Mh >> 63);
                    g0P = (g0P & ~1u) | inP; g0M = (g0M & ~1u) | inM;      // the carry words the traceback reads carry the patch too
                }
                const u32 elo = bitop3<0x90>(~(a1lo ^ m0[c]), b1lo, m1[c]), ehi = bitop3<0x90>(~(a1hi ^ m0[c]), b1hi, m1[c]);
                u32 ph_, mh_;
                block_step_core(elo, ehi, P1lo, P1hi, M1lo, M1hi, inP, inM, ph_, mh_);      // the window's bottom carries go nowhere
            }
            if (s < 32) {
                const u32 elo = bitop3<0x90>(~(a0lo ^ m0[s]), b0lo, m1[s]), ehi = bitop3<0x90>(~(a0hi ^ m0[s]), b0hi, m1[s]);
                block_step_fused(elo, ehi, P0lo, P0hi, M0lo, M0hi, __builtin_amdgcn_ubfe(hp, s, 1), 0u, g0P, g0M);
            }
        }
        const u32 rP = __builtin_bitreverse32(g0P), rM = __builtin_bitreverse32(g0M);
        if (half) { oPhi = rP; oMhi = rM; } else { oPlo = rP; oMlo = rM; }
    }
    P0 = mk64(P0lo, P0hi); M0 = mk64(M0lo, M0hi); P1 = mk64(P1lo, P1hi); M1 = mk64(M1lo, M1hi);
    gP = mk64(oPlo, oPhi); gM = mk64(oMlo, oMhi);
}

// ---------------------------------------------------------------------------
// 64 columns of one block, fast form: every lane runs all 64 columns, bases are
// pure ACGT, exported row is bit 63.  Fully unrolled; c is a literal.
//   a,b     pattern code planes of this block      T0,T1  text code planes of this chunk
//   hinP/M  carry-in words (bit c = column c)      houtP/M carry-out words
//   st      element of stored column 64k (a multiple of 8) at this slot.  A stored element is
//           {Pv AFTER the column, Mv BEFORE it}: exactly the two words one traceback step at
//           that column reads (Pv[h+1], Mv[h]; bpm_banded.c:994-1003).  Chunk column c is stored
//           column 64k + c + 1 = st + ((c+1) >> 3) * tstride + ((c+1) & 7); c == 63 goes to st_last
// ---------------------------------------------------------------------------
// STORE: 0 nothing | 1 every column's {Pv after, Mv before} (tiled; WindowEd history) |
//        2 a checkpoint {Pv, Mv after} every QE_CP_COLS-th column (BandEd fill: the traceback recomputes the columns between)
//        3 the same checkpoints, taken before each group of 8 columns, to st[grp * st_stride] (on-chip WindowEd windows)
// WIDE: two rolled passes over 32 literal columns instead of eight over 8.  The text / carry words are then
// addressed as 32-bit halves with literal bit positions: the per-group 64-bit shifts (6 SIMD cycles each, seven per
// group) and bit reversals disappear; costs 4x the code, so only the BandEd kernels use it.
template <int STORE, bool WIDE = false>
__device__ __forceinline__ void run64_fast(u64& P, u64& M, u64 a, u64 b, u64 T0, u64 T1,
                                           u64 hinP, u64 hinM, u64& houtP, u64& houtM,
                                           bool act, uint4* st, int64_t st_stride, uint4* st_last) {
    const u32 alo = lo32(a), ahi = hi32(a), blo = lo32(b), bhi = hi32(b);
    u32 Plo = lo32(P), Phi = hi32(P), Mlo = lo32(M), Mhi = hi32(M);
    u64 oP = 0, oM = 0;
    if (WIDE) {
        u32 oPlo = 0, oPhi = 0, oMlo = 0, oMhi = 0;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const u32 t0 = half ? hi32(T0) : lo32(T0), t1 = half ? hi32(T1) : lo32(T1);
            const u32 hp = half ? hi32(hinP) : lo32(hinP), hm = half ? hi32(hinM) : lo32(hinM);
            u32 gP = 0, gM = 0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0, c, 1);
                const u32 m1 = (u32)__builtin_amdgcn_sbfe((int)t1, c, 1);
                const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1);
                const u32 ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                const u32 PHin = __builtin_amdgcn_ubfe(hp, c, 1);
                const u32 MHin = __builtin_amdgcn_ubfe(hm, c, 1);
                const u32 Mblo = Mlo, Mbhi = Mhi;
                if (STORE == 3 && (c & 7) == 0) st[(4 * half + (c >> 3)) * st_stride] = make_uint4(Plo, Phi, Mlo, Mhi);
                block_step_collect(elo, ehi, Plo, Phi, Mlo, Mhi, PHin, MHin, gP, gM);
                if (STORE == 1) {
                    if (act) {
                        // chunk column 32 half + c is stored column (64k) + 32 half + c + 1
                        uint4* q = st + (int64_t)(4 * half + ((c + 1) >> 3)) * st_stride + ((c + 1) & 7);
                        if (c == 31 && half == 1) q = st_last;
                        *q = make_uint4(Plo, Phi, Mblo, Mbhi);
                    }
                }
                if (STORE == 2 && (c & (QE_CP_COLS - 1)) == QE_CP_COLS - 1) {
                    if (act) {
                        uint4* q = st + (int64_t)((QE_CPC / 2) * half + (c / QE_CP_COLS) + 1) * st_stride;
                        if (c == 31 && half == 1) q = st_last;
                        *q = make_uint4(Plo, Phi, Mlo, Mhi);
                    }
                }
            }
            // gP / gM hold the 32 carries MSB-first
            const u32 rP = __builtin_bitreverse32(gP), rM = __builtin_bitreverse32(gM);
            if (half) { oPhi = rP; oMhi = rM; } else { oPlo = rP; oMlo = rM; }
        }
        P = mk64(Plo, Phi);
        M = mk64(Mlo, Mhi);
        houtP = mk64(oPlo, oPhi);
        houtM = mk64(oMlo, oMhi);
        return;
    }
    // 8 groups of 8 columns: the group loop stays rolled (register pressure, I-cache), the
    // 8 columns inside are literal so every bit extract is a single v_bfe.
#pragma unroll 1
    for (int grp = 0; grp < 8; ++grp) {
        const u32 t0 = (u32)(T0 >> (8 * grp)), t1 = (u32)(T1 >> (8 * grp));
        const u32 hp = (u32)(hinP >> (8 * grp)), hm = (u32)(hinM >> (8 * grp));
        u32 gP = 0, gM = 0;
        if (STORE == 3) st[grp * st_stride] = make_uint4(Plo, Phi, Mlo, Mhi);   // {Pv, Mv} BEFORE column 8 grp
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0, j, 1);    // 0 / ~0: text code bit 0
            const u32 m1 = (u32)__builtin_amdgcn_sbfe((int)t1, j, 1);
            const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1);          // Eq: both code bits equal
            const u32 ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
            const u32 PHin = __builtin_amdgcn_ubfe(hp, j, 1);
            const u32 MHin = __builtin_amdgcn_ubfe(hm, j, 1);
            const u32 Mblo = Mlo, Mbhi = Mhi;
            block_step_collect(elo, ehi, Plo, Phi, Mlo, Mhi, PHin, MHin, gP, gM);
            if (STORE == 1) {
                if (act) {
                    // chunk column c = 8 grp + j is stored column (64k) + c + 1
                    uint4* q = st + ((j + 1) >> 3) * st_stride + ((j + 1) & 7);
                    if (j == 7 && grp == 7) q = st_last;
                    *q = make_uint4(Plo, Phi, Mblo, Mbhi);   // {Pv after, Mv before} this column
                }
            }
            if (STORE == 2 && j == 7) {
                if (act && ((8 * grp + 8) % QE_CP_COLS) == 0) {
                    // checkpoint after chunk column 8 (grp + 1) - 1: st addresses the chunk's checkpoint 0 of this slot
                    uint4* q = (grp == 7) ? st_last : st + ((8 * grp + 8) / QE_CP_COLS) * st_stride;
                    *q = make_uint4(Plo, Phi, Mlo, Mhi);     // {Pv, Mv} after this column
                }
            }
        }
        // gP / gM hold the group's carries MSB-first in their low byte
        oP |= (u64)(__builtin_bitreverse32(gP) >> 24) << (8 * grp);
        oM |= (u64)(__builtin_bitreverse32(gM) >> 24) << (8 * grp);
        if (STORE == 1) st += st_stride;
    }
    P = mk64(Plo, Phi);
    M = mk64(Mlo, Mhi);
    houtP = oP;
    houtM = oM;
}

// ---------------------------------------------------------------------------
// General form: per-lane column count, non-ACGT symbols (N matches N,
// dna_text.c:41-46 + bpm_banded.c:69-75), and a separate "score row" lvl for
// the last pattern block (level_mask, bpm_banded.c:88-102).  sP/sM collect bit
// lvl of every column's horizontal delta, houtP/M bit 63.
// ---------------------------------------------------------------------------
template <int STORE>
__device__ __forceinline__ void run64_general(u64& P, u64& M, u64 a, u64 b, u64 nn, u64 T0, u64 T1, u64 TN,
                                           u64 hinP, u64 hinM, u64& houtP, u64& houtM, u64& sP, u64& sM,
                                           int lvl, int ncols, bool st_on, uint4* st, int64_t st_stride, uint4* st_last) {
    u64 oP = 0, oM = 0, qP = 0, qM = 0;
    for (int c = 0; c < 64; ++c) {
        if (c < ncols) {
            const u64 m0 = (u64)0 - ((T0 >> c) & 1);
            const u64 m1 = (u64)0 - ((T1 >> c) & 1);
            const u64 acgt = ~(a ^ m0) & ~(b ^ m1) & ~nn;
            const u64 Eq = ((TN >> c) & 1) ? nn : acgt;
            u64 Ph, Mh;
            const u64 Min = M;
            block_step(Eq, P, M, (u32)((hinP >> c) & 1), (u32)((hinM >> c) & 1), Ph, Mh);
            oP |= (Ph >> 63) << c;
            oM |= (Mh >> 63) << c;
            qP |= ((Ph >> lvl) & 1) << c;
            qM |= ((Mh >> lvl) & 1) << c;
            if (STORE == 1 && st_on) {
                uint4* q = (c == 63) ? st_last : st + (int64_t)((c + 1) >> 3) * st_stride + ((c + 1) & 7);
                *q = make_uint4(lo32(P), hi32(P), lo32(Min), hi32(Min));
            }
            if (STORE == 2 && st_on && (c & (QE_CP_COLS - 1)) == QE_CP_COLS - 1) {
                uint4* q = (c == 63) ? st_last : st + (int64_t)((c / QE_CP_COLS) + 1) * st_stride;
                *q = make_uint4(lo32(P), hi32(P), lo32(M), hi32(M));
            }
        }
    }
    houtP = oP; houtM = oM; sP = qP; sM = qM;
}

// planes of 64 bases starting at bit offset `bit` of a sequence (funnel shift
// over two rows; sequences are padded with two zero rows)
__device__ __forceinline__ void load_planes(const u64* __restrict__ base, int bit, u64& a, u64& b, u64& nn) {
    const int w = bit >> 6, sh = bit & 63;
    const u64* q = base + 3 * (int64_t)w;
    u64 a0 = q[0], b0 = q[1], n0 = q[2];
    if (sh) {
        const u64 a1 = q[3], b1 = q[4], n1 = q[5];
        a0 = (a0 >> sh) | (a1 << (64 - sh));
        b0 = (b0 >> sh) | (b1 << (64 - sh));
        n0 = (n0 >> sh) | (n1 << (64 - sh));
    }
    a = a0; b = b0; nn = n0;
}

// planes of 128 bases starting at bit offset `bit`: [bit, bit+64) -> x, [bit+64, bit+128) -> y
__device__ __forceinline__ void load_planes2(const u64* __restrict__ base, int bit, u64 (&x)[3], u64 (&y)[3]) {
    const int w = bit >> 6, sh = bit & 63;
    const u64* q = base + 3 * (int64_t)w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const u64 r0 = q[k], r1 = q[3 + k];
        if (sh) {
            const u64 r2 = q[6 + k];
            x[k] = (r0 >> sh) | (r1 << (64 - sh));
            y[k] = (r1 >> sh) | (r2 << (64 - sh));
        } else { x[k] = r0; y[k] = r1; }
    }
}

// code planes 0 / 1 of 64 bases at bit offset `bit`
__device__ __forceinline__ void load_planes_ab(const u64* __restrict__ base, int bit, u64& a, u64& b) {
    // no branch: both word pairs are loaded whatever the offset is (the first pair twice when it is 0, so nothing past the
    // words the branchy form read), and a wait for them comes where the planes are first used, not inside an `if`
    const int w = bit >> 6, sh = bit & 63, o = sh ? 3 : 0;
    const u64* q = base + 3 * (int64_t)w;
    const u64 a0 = q[0], b0 = q[1], a1 = q[o], b1 = q[o + 1];
    a = (a0 >> sh) | ((a1 << 1) << (63 - sh));
    b = (b0 >> sh) | ((b1 << 1) << (63 - sh));
}

// load_planes_ab in two halves: the loads (raw words), and the shifts where the planes are needed
__device__ __forceinline__ void planes_ab_raw(const u64* __restrict__ base, int bit, u64 (&w)[4]) {
    const int o = (bit & 63) ? 3 : 0;
    const u64* q = base + 3 * (int64_t)(bit >> 6);
    w[0] = q[0]; w[1] = q[1]; w[2] = q[o]; w[3] = q[o + 1];
}
__device__ __forceinline__ void planes_ab_finish(const u64 (&w)[4], int bit, u64& a, u64& b) {
    const int sh = bit & 63;
    a = (w[0] >> sh) | ((w[2] << 1) << (63 - sh));
    b = (w[1] >> sh) | ((w[3] << 1) << (63 - sh));
}
// what a tile of k_traceback_sys needs from memory (band-edge records, checkpoint, carry words, planes as raw words)
struct TbsFetch { int q, Rb, cf_a, cf_b, cl_b; uint4 c0, w0; u64 t[4], p[4]; };

// base code (0..3, 4 = not ACGT) at position `pos` of a packed sequence
__device__ __forceinline__ int plane_code(const u64* __restrict__ base, int pos) {
    const u64* q = base + 3 * (int64_t)(pos >> 6);
    const int s = pos & 63;
    if ((q[2] >> s) & 1) return 4;
    return (int)(((q[0] >> s) & 1) | (((q[1] >> s) & 1) << 1));
}

// ===========================================================================
// pack: ASCII -> planes.  One wave per sequence; a lane takes 16 consecutive
// bases with one 16-byte load (a wave reads 1 KB = 16 rows per load instruction)
// and classifies them four at a time in 32-bit SWAR:
//   (c >> 1) & 3 is an injective 2-bit code of A, C, G, T in either case
//   (A 0, C 1, T 2, G 3; any injective code works) and indexes a 4-byte table of
//   the upper-case letters (v_perm_b32); the base is ACGT iff the table byte
//   equals its upper-cased self, anything else is the reference's code 4
//   (dna_text.c:41-46).
// Per-byte flags live in bit 7 of their byte; three shift-ors gather the four
// flags of a word into a nibble.  Four neighbouring lanes make one 64-base row.
// ===========================================================================
__device__ __forceinline__ u32 gather_msb4(u32 x) {        // flags at bits 7, 15, 23, 31 -> bits 28..31
    u32 t = (x << 7) | x;
    t = (x << 14) | t;
    return (x << 21) | t;
}

__global__ __launch_bounds__(256) void k_pack(PackArgs A) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= A.nseq) return;
    const int len = A.len[seq];
    const uint8_t* __restrict__ src = A.asc + A.asc_off[seq];
    u64* __restrict__ dst = A.planes + A.pl_off[seq];
    const int nrows = ((len + 63) >> 6) + 2;                 // two zero rows of padding (funnel-shift over-read)
    const int nspan = (nrows + 15) >> 4;
    u32 fl = 0;
    constexpr int UNR = 4;                                    // 16-byte loads in flight per lane
    constexpr u32 ALL_A = 0x41414141u;                        // filler past the end: 'A' packs to all-zero planes
    for (int s0 = 0; s0 < nspan; s0 += UNR) {
        uint4 raw[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int pos = (s0 + u) * 1024 + 16 * lane;
            uint4 x = make_uint4(ALL_A, ALL_A, ALL_A, ALL_A);
            if (pos + 16 <= len) {
                if (A.reverse) {
                    uint4 y; __builtin_memcpy(&y, src + (len - 16 - pos), 16);
                    x = make_uint4(__builtin_bswap32(y.w), __builtin_bswap32(y.z), __builtin_bswap32(y.y), __builtin_bswap32(y.x));
                } else {
                    __builtin_memcpy(&x, src + pos, 16);
                }
            } else if (pos < len) {                           // the one ragged lane of a sequence
                u32 wv[4] = {ALL_A, ALL_A, ALL_A, ALL_A};
                for (int i = 0; i < len - pos; ++i) {
                    const u32 c = A.reverse ? src[len - 1 - pos - i] : src[pos + i];
                    wv[i >> 2] = (wv[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | (c << (8 * (i & 3)));
                }
                x = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            raw[u] = x;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int span = s0 + u;
            if (span >= nspan) break;
            const u32 wv[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            u32 accA = 0, accB = 0, accN = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 w = wv[k];
                const u32 want = __builtin_amdgcn_perm(0u, 0x47544341u, (w >> 1) & 0x03030303u);   // 'A','C','T','G'
                const u32 y = want ^ (w & 0xDFDFDFDFu);                                              // case-insensitive (dna_text.c:44-45)
                const u32 z = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;                                 // bit 7 of a byte set <=> not ACGT
                const u32 ok = ~z & 0x80808080u, bad = z & 0x80808080u;
                accA = (accA >> 4) | (gather_msb4((w << 6) & ok) & 0xF0000000u);                     // code bit 0 = ASCII bit 1
                accB = (accB >> 4) | (gather_msb4((w << 5) & ok) & 0xF0000000u);                     // code bit 1 = ASCII bit 2
                accN = (accN >> 4) | (gather_msb4(bad) & 0xF0000000u);
                if ((w << 2) & ok) fl |= FLAG_NONCANON;                                              // lower-case base: raw != encoded compare
                if (bad) {
                    fl |= FLAG_HAS_N;
                    for (int i = 0; i < 4; ++i)
                        if (((bad >> (8 * i + 7)) & 1) && ((w >> (8 * i)) & 0xFF) != 'N') fl |= FLAG_NONCANON;   // IUPAC and the rest
                }
            }
            // 16-bit masks of this lane's 16 bases -> 64-bit row words on every fourth lane
            const u32 ab = (accA >> 16) | (accB & 0xFFFF0000u), nn = accN >> 16;
            const u32 ab1 = __shfl_down(ab, 1), nn1 = __shfl_down(nn, 1);
            const u32 a32 = (ab & 0xFFFFu) | (ab1 << 16), b32 = (ab >> 16) | (ab1 & 0xFFFF0000u), n32 = nn | (nn1 << 16);
            const u32 a32h = __shfl_down(a32, 2), b32h = __shfl_down(b32, 2), n32h = __shfl_down(n32, 2);
            const int row = span * 16 + (lane >> 2);
            if ((lane & 3) == 0 && row < nrows) {
                u64* q = dst + 3 * (int64_t)row;
                q[0] = mk64(a32, a32h); q[1] = mk64(b32, b32h); q[2] = mk64(n32, n32h);
            }
        }
    }
    if (A.flags && __any(fl != 0)) {
        u32 f = fl;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o);
        if (lane == 0) atomicOr(&A.flags[seq], f);
    }
}

// ===========================================================================
// Packed input: the wire words become planes once, at batch creation; runs skip k_pack.
// ===========================================================================
__device__ __forceinline__ u32 even_bits(u64 x) {          // bits 0, 2, 4, ... of x, compacted
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (u32)x;
}

__global__ __launch_bounds__(256) void k_unpack_wire(WireArgs A) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= A.nseq) return;
    const int len = A.len[seq];
    const int nw = (len + 63) >> 6;
    const u64* __restrict__ src = A.words + A.w_off[seq];
    u64* __restrict__ dst = A.planes + A.pl_off[seq];
    bool any_n = false;
    for (int r = lane; r < nw + 2; r += 64) {
        u64 a = 0, b = 0, nn = 0;
        if (r < nw) {
            if (A.wire == 3) { a = src[3 * (int64_t)r]; b = src[3 * (int64_t)r + 1]; nn = src[3 * (int64_t)r + 2]; }
            else {
                const int nwords = (len + 31) >> 5;               // 32 bases per wire word
                const u64 w0 = src[2 * (int64_t)r], w1 = (2 * r + 1 < nwords) ? src[2 * (int64_t)r + 1] : 0;
                a = (u64)even_bits(w0) | ((u64)even_bits(w1) << 32);
                b = (u64)even_bits(w0 >> 1) | ((u64)even_bits(w1 >> 1) << 32);
            }
            const int valid = min(len - 64 * r, 64);
            const u64 mask = (valid >= 64) ? QE_ONES : ((((u64)1) << valid) - 1);
            nn &= mask; a &= mask & ~nn; b &= mask & ~nn;         // a non-ACGT base has code bits 0
            any_n |= nn != 0;
        }
        u64* q = dst + 3 * (int64_t)r;
        q[0] = a; q[1] = b; q[2] = nn;
    }
    if (A.flags && __any(any_n)) { if (lane == 0) atomicOr(&A.flags[seq], (u32)FLAG_HAS_N); }
}

__global__ __launch_bounds__(256) void k_reverse_planes(RevArgs A) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= A.nseq) return;
    const int len = A.len[seq];
    const int nw = (len + 63) >> 6;
    const u64* __restrict__ src = A.fwd + A.pl_off[seq];
    u64* __restrict__ dst = A.rev + A.pl_off[seq];
    for (int r = lane; r < nw + 2; r += 64) {
        u64 o[3] = {0, 0, 0};
        if (r < nw) {
            // reversed positions [64 r, 64 r + 64) are forward positions e-1 down to e-64, e = len - 64 r
            const int s = len - 64 * r - 64;
            u64 x[3];
            if (s >= 0) load_planes(src, s, x[0], x[1], x[2]);
            else { x[0] = src[0] << (-s); x[1] = src[1] << (-s); x[2] = src[2] << (-s); }
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = __brevll(x[k]);
        }
        u64* q = dst + 3 * (int64_t)r;
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    }
}

// ---------------------------------------------------------------------------
// band geometry (bpm_banded.c:121-135)
// ---------------------------------------------------------------------------
struct Geom { int cutoff, diff, prolog, ebb, fin; };
__device__ __forceinline__ Geom band_geometry(int m, int n, int cutoff_in) {
    Geom g;
    const int kend = abs(n - m) + 1;
    g.cutoff = max(max(kend, cutoff_in), 65);
    g.diff = m - n;
    const int rel = (g.cutoff - abs(g.diff) + 1) / 2;
    if (g.diff >= 0) {
        g.prolog = (rel + 63) / 64;
        g.ebb = (rel + g.diff + 63) / 64 + 1 + g.prolog;
    } else {
        g.prolog = (rel - g.diff + 63) / 64;
        g.ebb = (rel + 63) / 64 + 1 + g.prolog;
    }
    g.fin = g.prolog * 64 + g.diff;
    return g;
}

struct GroupWs {
    u64* Pv; u64* Mv; int32_t* S; int16_t* cf; int16_t* cl;
};
__device__ __forceinline__ GroupWs group_ws(uint8_t* ws, int64_t off, int ns, int nrows, int nch) {
    GroupWs w;
    uint8_t* p = ws + off;
    w.Pv = (u64*)p;              p += (int64_t)(ns + 1) * 64 * 8;
    w.Mv = (u64*)p;              p += (int64_t)(ns + 1) * 64 * 8;
    w.S = (int32_t*)p;           p += (int64_t)nrows * 64 * 4;
    w.cf = (int16_t*)p;          p += (int64_t)nch * 64 * 2;
    w.cl = (int16_t*)p;
    return w;
}

// ===========================================================================
// BandEd: score-only (FILL = false, bpm_banded.c:791-964) or full-matrix fill
// (FILL = true, bpm_banded.c:199-316).  One lane per task.
// ===========================================================================
#ifndef QE_FILL_K
#define QE_FILL_K 3          // slots per skewed pass of the fill (1: single-slot passes only)
#endif
template <bool FILL>
__global__ __launch_bounds__(512) void k_banded(BandedArgs A) {
    const int g = QE_GROUP_INDEX(), lane = threadIdx.x & 63, t = g * 64 + lane;
    if (g * 64 >= A.T.ntasks) return;
    int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    if (A.only_if != nullptr && pair >= 0 && A.only_if[t] == 0) pair = -1;
    const bool valid = pair >= 0;
    if (!__any(valid)) return;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0, tfin = 0;
    const u64* pp = A.P.pl_p;
    const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        cut_in = A.T.cutoff[t];
        tfin = FILL ? n : A.T.tfin[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair];
        tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const bool hasN = (fl & FLAG_HAS_N) != 0;
    const Geom G = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    // the score-only kernels use their own narrower band (bpm_banded.c:801-803) -- unless the launch asks for the fill's
    const bool fgeom = FILL || A.fill_geom != 0;
    const int nsl = fgeom ? G.ebb : ((G.cutoff + 63) >> 6) + 1;
    const int stop_row = fgeom ? nw - 1 : nw;                      // bpm_banded.c:295 / 917
    const int lvl_last = (m - 1) & 63;                             // level_mask of the last block
    int first = G.prolog, last = nsl - 1, pos_v = -G.prolog, pos_h = 0;
    int max_row_init = nsl - 1;
    u32 adv = 0;

    const int gns = A.g_nslots[g], gnr = A.g_nrows[g], gnch = A.g_nch[g];
    const GroupWs W = group_ws(A.ws, A.g_ws_off[g], gns, gnr, gnch);
    u64* const Pv = W.Pv + 64 + lane;        // slot s lives at Pv[s * 64]; slot -1 is addressable
    u64* const Mv = W.Mv + 64 + lane;
    int32_t* const S = W.S + lane;           // scores[] indexed by absolute block row (bpm_banded.c:180-197)
    // fill: per group  cp[QE_CPC nch][ns][64] = {Pv, Mv} after every QE_CP_COLS-th stored column (in the slot numbering of the
    // chunk that starts at / contains it), then  hw[nch][ns][64] = the carry-in words of every (chunk, slot)
    uint4* const cp = FILL ? A.mat + A.g_mat_off[g] + lane : nullptr;
    const int64_t cps = (int64_t)gns * 64;                         // uint4 units between checkpoint columns
    uint4* const hw = FILL ? cp + (int64_t)QE_CPC * gnch * cps : nullptr;

    // bpm_reset_search (bpm_banded.c:180-197)
    for (int s = 0; s < gns; ++s) {
        if (valid && s < nsl) {
            Pv[(int64_t)s * 64] = QE_ONES;
            Mv[(int64_t)s * 64] = 0;
            S[(int64_t)s * 64] = 64 * (s + 1);
            if (FILL) cp[(int64_t)s * 64] = make_uint4(~0u, ~0u, 0u, 0u);
        }
    }
    if (FILL && valid) { W.cf[lane] = (int16_t)first; W.cl[lane] = (int16_t)last; }

    const int nfull = tfin >> 6, tail = tfin & 63;
    const int my_chunks = valid ? nfull + (tail ? 1 : 0) : 0;
    const int wave_chunks = wave_max(my_chunks);

    for (int k = 0; k < wave_chunks; ++k) {
        const int ncols = (k < nfull) ? 64 : ((k == nfull) ? tail : 0);
        const bool on = valid && ncols > 0;
        u64 T0 = 0, T1 = 0, TN = 0;
        if (on) load_planes(tp, t0 + 64 * k, T0, T1, TN);
        const int rhi = min(last, nw - 1 - pos_v);                 // rows >= nw are never computed (A.7(2))
        // The slots a wave walks in this chunk.  Uniform batches: every lane's band sits at the same slots, the wave walks
        // first .. last of all of them (i = x).  Pairs whose paths drift apart (large indels: the bands of a wave's 64
        // lanes, a few slots tall each, lie scattered over 40-60 slots) would make every lane wait through the UNION of the
        // bands; lane-relative, step x of a chunk is slot first + x of each lane's own band, and the walk is as long as the
        // tallest band.  The rows a lane touches are then its own (state, checkpoints and carry words are [slot][lane]
        // rows: a wave's access is one row only while its lanes agree on the slot), which costs less than the idle steps.
        const int fmin = wave_min(on ? first : 0x7fffffff), fmax = wave_max(on ? first : -0x7fffffff);
        const bool rel = A.lane_rel != 0 && fmin < fmax;
        const int ib = rel ? (on ? first : 0) : 0;
        const int i0 = rel ? 0 : fmin;
        const int i1 = rel ? wave_max(on ? rhi - first : -0x7fffffff) : wave_max(on ? rhi : -0x7fffffff);
        u64 hinP = QE_ONES, hinM = 0;
        // fill: the state and pattern planes of the NEXT single-slot pass are loaded before the current one computes
        // (unconditionally, indices clamped; issued ahead of the pass's checkpoint stores they return under its compute --
        // loaded at the point of use they queue behind those stores: 42 % of the kernel was s_waitcnt, profiles/README.md)
        u64 qP = 0, qM = 0, qa0 = 0, qb0 = 0, qa1 = 0, qb1 = 0;
        int qS = 0, qi = -0x7fffffff;
        const int psh = p0 & 63;
        const bool unaligned = __any(psh != 0);
        auto prefetch = [&](int j) {
            const int jc = min(max(j, 0), gns - 1), rc = min(max(j + pos_v, 0), gnr - 1), rp = min(max(j + pos_v, 0), nw);
            qP = Pv[(int64_t)jc * 64]; qM = Mv[(int64_t)jc * 64]; qS = S[(int64_t)rc * 64];
            const u64* w = pp + 3 * (int64_t)((p0 >> 6) + rp);
            qa0 = w[0]; qb0 = w[1];
            if (unaligned) { qa1 = w[3]; qb1 = w[4]; }            // Hirschberg children start anywhere in their pair's pattern
            qi = j;
        };
        if (FILL) prefetch(i0 + ib);
        for (int x = i0; x <= i1; ++x) {
            const int i = x + ib;
            const bool act = on && i >= first && i <= rhi;
            const int r = i + pos_v;
            if (!FILL) {
                // K slots in one pass when every lane has all of them or none, full ACGT chunks, not the last block row
                const int lo = on ? first : 0x7fffffff, hi_ = on ? rhi : -0x7fffffff;
                const bool plain = !(on && (ncols != 64 || hasN));
                auto uniform = [&](int K) {
                    const bool all = i >= lo && i + K - 1 <= hi_, none = i + K - 1 < lo || i > hi_;
                    const bool bad = !(all || none) || (all && (!plain || r + K - 1 >= nw - 1));
                    return !__any(bad);
                };
                if (i == first) { hinP = QE_ONES; hinM = 0; }
                if (x + 3 <= i1 && uniform(4)) { slots_pass<4>(act, i, r, Pv, Mv, S, S, 64, pp, p0, T0, T1, hinP, hinM, adv); x += 3; continue; }
                if (x + 1 <= i1 && uniform(2)) { slots_pass<2>(act, i, r, Pv, Mv, S, S, 64, pp, p0, T0, T1, hinP, hinM, adv); x += 1; continue; }
            }
            if (FILL && QE_FILL_K > 1 && A.fill_multi && x + QE_FILL_K - 1 <= i1) {
                // K slots in one skewed pass, every lane with the slots inside its own band, unless a lane needs the
                // general form for one of them (partial chunk, N, the pattern's last block row)
                constexpr int K = QE_FILL_K;
                bool actk[K], slowk = false;
                u32 topk[K];
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    actk[q] = on && (i + q) >= first && (i + q) <= rhi;
                    topk[q] = ((i + q) == first) ? 1u : 0u;
                    slowk |= actk[q] && (ncols != 64 || hasN || (i + q + pos_v) == nw - 1);
                }
                if (!__any(slowk)) {
                    u64 P[K], M[K], a[K], b[K], oP[K], oM[K];
                    int sc[K];
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        P[q] = 0; M[q] = 0; a[q] = 0; b[q] = 0; sc[q] = 0;
                        if (actk[q]) {
                            u64 nn;
                            P[q] = Pv[(int64_t)(i + q) * 64]; M[q] = Mv[(int64_t)(i + q) * 64]; sc[q] = S[(int64_t)(r + q) * 64];
                            load_planes(pp, p0 + 64 * (r + q), a[q], b[q], nn);
                        }
                    }
                    if (i == first) { hinP = QE_ONES; hinM = 0; }
                    uint4* const st0 = cp + (int64_t)(QE_CPC * k) * cps + (int64_t)i * 64;
                    uint4* const stl0 = cp + (int64_t)(QE_CPC * k + QE_CPC) * cps + (int64_t)(i - 1) * 64;
                    run64_skew_fill<K>(P, M, a, b, T0, T1, hinP, hinM, oP, oM, actk, topk, st0, cps,
                                       (i == 0) ? st0 + QE_CPC * cps : stl0, stl0);
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        if (actk[q]) {
                            const u64 iP = (q == 0) ? hinP : (topk[q] ? QE_ONES : oP[q > 0 ? q - 1 : 0]);
                            const u64 iM = (q == 0) ? hinM : (topk[q] ? (u64)0 : oM[q > 0 ? q - 1 : 0]);
                            hw[((int64_t)k * gns + (i + q)) * 64] = make_uint4(lo32(iP), hi32(iP), lo32(iM), hi32(iM));
                            S[(int64_t)(r + q) * 64] = sc[q] + __popcll(oP[q]) - __popcll(oM[q]);
                            Pv[(int64_t)(i + q - 1) * 64] = P[q]; Mv[(int64_t)(i + q - 1) * 64] = M[q];   // band shift (bpm_banded.c:903-909)
                            adv += 64u;
                        }
                    }
                    hinP = oP[K - 1]; hinM = oM[K - 1];
                    x += K - 1;
                    continue;
                }
            }
            u64 P = 0, M = 0, a = 0, b = 0, nn = 0;
            int sc = 0;
            if (FILL) {
                if (qi != i) prefetch(i);                          // after a multi-slot pass
                if (act) {
                    P = qP; M = qM; sc = qS;
                    a = (qa0 >> psh) | ((qa1 << 1) << (63 - psh));
                    b = (qb0 >> psh) | ((qb1 << 1) << (63 - psh));
                    if (hasN || r == nw - 1) load_planes(pp, p0 + 64 * r, a, b, nn);
                }
                prefetch(i + 1);
            } else if (act) {
                P = Pv[(int64_t)i * 64];
                M = Mv[(int64_t)i * 64];
                sc = S[(int64_t)r * 64];
                load_planes(pp, p0 + 64 * r, a, b, nn);
            }
            if (i == first) { hinP = QE_ONES; hinM = 0; }          // PHin = 1 into the band's top block
            const bool lastblk = (r == nw - 1);
            uint4* st = nullptr; uint4* st_last = nullptr;
            if (FILL) {
                // the chunk's last column belongs to the NEXT chunk's slot numbering (bpm_banded.c:279-287)
                st = cp + (int64_t)(QE_CPC * k) * cps + (int64_t)i * 64;
                st_last = cp + (int64_t)(QE_CPC * k + QE_CPC) * cps + (int64_t)(i - 1) * 64;
                if (i == 0) st_last = st + QE_CPC * cps;           // slot -1 does not exist; dropped row, never read
                if (act) hw[((int64_t)k * gns + i) * 64] = make_uint4(lo32(hinP), hi32(hinP), lo32(hinM), hi32(hinM));
            }
            u64 houtP, houtM, sP, sM;
            const bool slow = act && (ncols != 64 || hasN || lastblk);
            if (!__any(slow)) {
                run64_fast<FILL ? 2 : 0, true>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM, act, st, cps, st_last);
                sP = houtP; sM = houtM;
            } else {
                run64_general<FILL ? 2 : 0>(P, M, a, b, nn, T0, T1, TN, hinP, hinM, houtP, houtM, sP, sM,
                                            lastblk ? lvl_last : 63, act ? ncols : 0, true, st, cps, st_last);
            }
            if (act) {
                sc += __popcll(sP) - __popcll(sM);
                S[(int64_t)r * 64] = sc;
                // in-place band shift: slot i of this chunk is slot i-1 of the next (bpm_banded.c:903-909)
                const int64_t dst = (ncols == 64) ? (int64_t)(i - 1) * 64 : (int64_t)i * 64;
                Pv[dst] = P;
                Mv[dst] = M;
                adv += (u32)ncols;
            }
            hinP = houtP; hinM = houtM;
        }
        if (on && ncols == 64) {
            // every-64-columns bookkeeping (bpm_banded.c:889-922 / 264-301)
            const bool c1 = (first + 2 < last) && (G.fin > 64 * (first + 1));
            bool cut_lo = false;
            if (c1) cut_lo = S[(int64_t)(first + pos_v + 1) * 64] + (G.fin - 64 * (first + 1)) > G.cutoff;
            if (cut_lo && pos_h >= G.prolog) first++;
            else if (!cut_lo && pos_h < G.prolog) first--;
            Pv[(int64_t)last * 64] = QE_ONES;
            Mv[(int64_t)last * 64] = 0;
            if (FILL) cp[(int64_t)(QE_CPC * k + QE_CPC) * cps + (int64_t)last * 64] = make_uint4(~0u, ~0u, 0u, 0u);
            const int pos = last + pos_v;
            S[(int64_t)(pos + 1) * 64] = S[(int64_t)pos * 64] + 64;
            max_row_init = max(max_row_init, pos + 1);
            const bool c2 = (first + 2 < last) && (64 * (last - 1) > G.fin);
            bool cut_hi = false;
            if (c2) cut_hi = S[(int64_t)(last + pos_v - 1) * 64] + (64 * (last - 1) - G.fin) > G.cutoff;
            if (cut_hi || (pos_v + last >= stop_row)) last--;
            pos_v++;
            pos_h++;
            if (FILL) { W.cf[(int64_t)pos_h * 64 + lane] = (int16_t)first; W.cl[(int64_t)pos_h * 64 + lane] = (int16_t)last; }
        }
    }
    if (valid) {
        // final score read-out (bpm_banded.c:952-961); -1: band never reached the last block (A.7(3))
        const int row = nw - 1;
        int score = -1;
        if (row <= max_row_init) {
            score = S[(int64_t)row * 64];
            if (m & 63) score -= 64 - (m & 63);
        }
        A.o_score[t] = score;
        A.o_first[t] = first;
        A.o_last[t] = last;
        A.o_posv[t] = pos_v;
        A.o_maxrow[t] = max_row_init;
        A.o_adv[t] = adv;                                     // block-advances (one per block per column)
    }
}

template __global__ void k_banded<false>(BandedArgs);
template __global__ void k_banded<true>(BandedArgs);


// ===========================================================================
// BandEd score-only, cooperative form: G adjacent lanes share one alignment.
//
// Lane g of a group owns the fixed band-slot range [g c, (g+1) c) (c >= 2 slots)
// and runs ONE CHUNK BEHIND lane g-1: at step s it walks its slots for chunk
// s - g, so the group is a systolic pipeline over (slot range, chunk) with the
// same carry-word block walk as k_banded.  What moves between lanes:
//   * the 64 carry bits out of lane g-1's last slot: one __shfl_up per step;
//   * band state: slot i's result is stored to slot i-1 (the reference's 64-column
//     band shift, bpm_banded.c:903-909), so lane g+1's first slot becomes lane g's
//     last slot through the group's shared workspace -- lane g+1 writes it in
//     iteration 0 of a step, lane g reads it in a later iteration of that step;
//   * band-edge decisions (bpm_banded.c:889-922): first/last of every chunk live in
//     CF[] / CL[] with "known up to" counters; the lane owning slot first+1 decides
//     the top, the lane owning slot last the bottom.  A lane that starts a chunk
//     before its `first` is decided assumes "no cut" -- harmless after the prologue,
//     (the extra slot it computes is dropped), otherwise the task is flagged.
// Every value is bit-identical to k_banded<false>; a flagged task (o_abort) is
// simply recomputed by k_banded<false>.
// ===========================================================================
__global__ __launch_bounds__(512) void k_banded_coop(CoopArgs A) {
    const int lane = threadIdx.x & 63, w = QE_GROUP_INDEX();
    const int G = A.G, NA = 64 / G;
    if (w * NA >= A.T.ntasks) return;
    const int q = lane / G, g = lane - q * G;
    const int t = w * NA + q;
    const int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    const bool valid = pair >= 0;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0, tfin = 0;
    const u64* pp = A.P.pl_p;
    const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        cut_in = A.T.cutoff[t]; tfin = A.T.tfin[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair];
        tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const bool hasN = (fl & FLAG_HAS_N) != 0;
    const Geom GE = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    const int nsl = ((GE.cutoff + 63) >> 6) + 1;
    const int lvl_last = (m - 1) & 63;
    const int prolog = GE.prolog;
    // my slots
    const int cper = max((nsl + G - 1) / G, 2);
    const int slo = min(g * cper, nsl), shi = min(slo + cper, nsl) - 1;

    const int wns = A.w_nslots[w], wnr = A.w_nrows[w], wnch = A.w_nch[w];
    uint8_t* base = A.ws + A.w_ws_off[w];
    u64* const Pv = (u64*)base + NA + q;                                   base += (int64_t)(wns + 1) * NA * 8;
    u64* const Mv = (u64*)base + NA + q;                                   base += (int64_t)(wns + 1) * NA * 8;
    // scores[] is double-buffered by chunk parity: a row's value AFTER chunk k lives in S[k & 1].  A lane
    // that is one chunk ahead may already have advanced a row the deciding lane still needs at chunk k.
    int32_t* const S0 = (int32_t*)base + q;                                base += (int64_t)2 * wnr * NA * 4;
    const int64_t Spar = (int64_t)wnr * NA;
    volatile int16_t* const CF = (volatile int16_t*)base + q;              base += (int64_t)wnch * NA * 2;
    volatile int16_t* const CL = (volatile int16_t*)base + q;              base += (int64_t)wnch * NA * 2;
    volatile int32_t* const KF = (volatile int32_t*)base + q;              base += (int64_t)NA * 4;
    volatile int32_t* const KL = (volatile int32_t*)base + q;

    // bpm_reset_search, each lane its own slots
    if (valid) {
        for (int s = slo; s <= shi; ++s) {
            Pv[(int64_t)s * NA] = QE_ONES;
            Mv[(int64_t)s * NA] = 0;
            S0[(int64_t)s * NA] = 64 * (s + 1);
            S0[Spar + (int64_t)s * NA] = 64 * (s + 1);
        }
        if (g == 0) { CF[0] = (int16_t)prolog; CL[0] = (int16_t)(nsl - 1); *KF = 1; *KL = 1; }
    }
    // The lanes of a group hand band state, scores and band edges to each other through the group's workspace: a store by
    // one lane, a load by another in a LATER instruction of the same wave -- ordered by the hardware (one wave's memory
    // operations to an address reach the cache in program order), and kept in that order by the compiler through
    // wavefront-scope fences between the passes of a step and around the decision rounds
#define QE_COOP_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
    QE_COOP_FENCE();
    const int nfull = tfin >> 6, tail = tfin & 63;
    const int my_chunks = valid ? nfull + (tail ? 1 : 0) : 0;
    const int nsteps = wave_max(my_chunks > 0 ? my_chunks + G - 1 : 0);
    u64 lastP = 0, lastM = 0;          // carry words out of my last processed slot in the previous step
    u32 adv = 0;
    int maxrow = nsl - 1, aborted = 0;

    for (int s = 0; s < nsteps; ++s) {
        const u64 cinP = __shfl_up(lastP, 1), cinM = __shfl_up(lastM, 1);
        const int k = s - g;
        const bool on = valid && k >= 0 && k < my_chunks;
        const int ncols = (k < nfull) ? 64 : tail;
        const int pos_v = k - prolog;
        int32_t* const Srd = S0 + (((k - 1) & 1) ? Spar : 0);     // values after chunk k-1
        int32_t* const Swr = S0 + ((k & 1) ? Spar : 0);           // values after chunk k
        int fk = 0, lk = -1;
        if (on) {
            const int kf = *KF, kl = *KL;
            if (k < kf) fk = CF[(int64_t)k * NA];
            else {
                // `first` of my chunk is not decided yet (the deciding lane runs behind me).  It drops by
                // at most one slot per undecided chunk: if even that bound stays below my slots I have
                // nothing to do; one chunk of doubt is resolved by assuming "no cut" (bpm_banded.c:894-901)
                const int prev = CF[(int64_t)(kf - 1) * NA];
                const int d = k - (kf - 1);
                if (shi < prev - d) fk = shi + 1;
                else if (d == 1) fk = (k - 1 < prolog) ? prev - 1 : prev;
                else { fk = prev; aborted = 1; }
            }
            lk = (k < kl) ? CL[(int64_t)k * NA] : CL[(int64_t)(kl - 1) * NA];   // exact or an upper bound; my range end is the same
        }
        const int lo = max(slo, fk), hi = min(min(shi, lk), nw - 1 - pos_v);
        const int cnt = on ? max(hi - lo + 1, 0) : 0;
        // my only slot is my last slot: lane g+1 hands it over in iteration 0, take it in iteration 1
        const int off = (cnt == 1 && lo == shi) ? 1 : 0;
        const int maxit = wave_max(cnt + off);
        u64 T0 = 0, T1 = 0, TN = 0;
        if (cnt > 0) load_planes(tp, t0 + 64 * k, T0, T1, TN);
        u64 hinP = (lo == fk) ? QE_ONES : cinP, hinM = (lo == fk) ? 0 : cinM;   // PHin = 1 into the band's top block
        for (int it = 0; it < maxit; ++it) {
            const int i = lo + it - off;
            const bool act = cnt > 0 && it >= off && i <= hi;
            const int r = i + pos_v;
            {
                // K of my slots in one skewed pass (slots_pass, as in k_banded) when every lane of the wave has K more or
                // none: full ACGT chunks, not the last block row -- and never my range's last slot shi: lane g+1 hands
                // its state over during this step's first pass, so it is read in a later one (see `off`)
                const int hi_m = min(hi, shi - 1);
                const bool mine = cnt > 0 && it >= off;
                const bool plainrow = !(on && (ncols != 64 || hasN));
                auto uniform = [&](int K) {
                    const bool all = mine && i + K - 1 <= hi_m, none = cnt == 0 || (mine && i > hi);
                    const bool bad = !(all || none) || (all && (!plainrow || r + K - 1 >= nw - 1));
                    return !__any(bad);
                };
                const bool all4 = mine && i + 3 <= hi_m, all2 = mine && i + 1 <= hi_m;
                if (it + 3 < maxit && uniform(4)) {
                    slots_pass<4>(all4, i, r, Pv, Mv, Srd, Swr, NA, pp, p0, T0, T1, hinP, hinM, adv);
                    if (all4) { lastP = hinP; lastM = hinM; }
                    it += 3;
                    QE_COOP_FENCE();
                    continue;
                }
                if (it + 1 < maxit && uniform(2)) {
                    slots_pass<2>(all2, i, r, Pv, Mv, Srd, Swr, NA, pp, p0, T0, T1, hinP, hinM, adv);
                    if (all2) { lastP = hinP; lastM = hinM; }
                    it += 1;
                    QE_COOP_FENCE();
                    continue;
                }
            }
            u64 P = 0, M = 0, a = 0, b = 0, nn = 0;
            int sc = 0;
            if (act) {
                P = Pv[(int64_t)i * NA];
                M = Mv[(int64_t)i * NA];
                sc = Srd[(int64_t)r * NA];
                load_planes(pp, p0 + 64 * r, a, b, nn);
            }
            const bool lastblk = (r == nw - 1);
            u64 houtP = 0, houtM = 0, sP, sM;
            const bool slow = act && (ncols != 64 || hasN || lastblk);
            if (!__any(slow)) {
                run64_fast<0, true>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM, act, nullptr, 0, nullptr);
                sP = houtP; sM = houtM;
            } else {
                run64_general<0>(P, M, a, b, nn, T0, T1, TN, hinP, hinM, houtP, houtM, sP, sM,
                                     lastblk ? lvl_last : 63, act ? ncols : 0, false, nullptr, 0, nullptr);
            }
            if (act) {
                sc += __popcll(sP) - __popcll(sM);
                Swr[(int64_t)r * NA] = sc;
                const int64_t dst = (ncols == 64) ? (int64_t)(i - 1) * NA : (int64_t)i * NA;
                Pv[dst] = P;
                Mv[dst] = M;
                adv += (u32)ncols;
                hinP = houtP; hinM = houtM;
                lastP = houtP; lastM = houtM;
            }
            QE_COOP_FENCE();          // lane g+1 stored the slot lane g loads in a later iteration of this step
        }
        // ---- band-edge decisions of every chunk that completed its deciding slot in this step.
        // Two rounds: a decision made in round 1 can enable the lane above (one chunk ahead) in round 2.
        const bool full = on && ncols == 64;
        for (int round = 0; round < 6; ++round) {
            bool decided = false;
            if (full) {
                // top (bpm_banded.c:889-901): decided by the owner of slot first + 1
                int kf = *KF;
                if (kf == k + 1) {
                    const int f = CF[(int64_t)k * NA];
                    const int dslot = min(f + 1, nsl - 1);
                    if (dslot >= slo && dslot <= shi) {
                        const int kl = *KL;
                        const int lub = CL[(int64_t)(min(k, kl - 1)) * NA];
                        const int llb = (k < kl) ? lub : lub - (k - (kl - 1));     // last drops by at most one per chunk
                        bool tall;
                        if (f + 2 < llb) tall = true;
                        else if (f + 2 >= lub) tall = false;
                        else { tall = true; aborted = 1; }
                        bool cut_lo = false;
                        if (tall && GE.fin > 64 * (f + 1))
                            cut_lo = Swr[(int64_t)(f + pos_v + 1) * NA] + (GE.fin - 64 * (f + 1)) > GE.cutoff;
                        int fnew = f;
                        if (cut_lo && k >= prolog) fnew = f + 1;
                        else if (!cut_lo && k < prolog) fnew = f - 1;
                        CF[(int64_t)(k + 1) * NA] = (int16_t)fnew;
                        *KF = k + 2;
                        decided = true;
                    }
                }
                // bottom (bpm_banded.c:903-921): decided by the owner of slot last, once the new first is known
                kf = *KF;
                const int kl = *KL;
                if (kl == k + 1 && kf >= k + 2) {
                    const int l = CL[(int64_t)k * NA];
                    if (l >= slo && l <= shi) {
                        const int fnew = CF[(int64_t)(k + 1) * NA];
                        Pv[(int64_t)l * NA] = QE_ONES;
                        Mv[(int64_t)l * NA] = 0;
                        const int pos = l + pos_v;
                        Swr[(int64_t)(pos + 1) * NA] = Swr[(int64_t)pos * NA] + 64;
                        maxrow = max(maxrow, pos + 1);
                        bool cut_hi = false;
                        if ((fnew + 2 < l) && (64 * (l - 1) > GE.fin))
                            cut_hi = Swr[(int64_t)(l + pos_v - 1) * NA] + (64 * (l - 1) - GE.fin) > GE.cutoff;
                        const int lnew = (cut_hi || (pos_v + l >= nw)) ? l - 1 : l;
                        CL[(int64_t)(k + 1) * NA] = (int16_t)lnew;
                        *KL = k + 2;
                        decided = true;
                    }
                }
            }
            QE_COOP_FENCE();
            if (!__any(decided)) break;
        }
        // a wrong "no cut" guess before the end of the prologue put the carry chain on the wrong top slot
        if (on && k >= 1 && k <= prolog && fk <= shi) {
            const int kf = *KF;
            if (kf > k && CF[(int64_t)k * NA] != fk) aborted = 1;
        }
    }
    QE_COOP_FENCE();
    // group reductions: adv (sum), maxrow (max), aborted (or)
    for (int o = 1; o < G; o <<= 1) {
        adv += __shfl_xor(adv, o);
        maxrow = max(maxrow, __shfl_xor(maxrow, o));
        aborted |= __shfl_xor(aborted, o);
    }
    if (valid && g == 0) {
        if (*KF <= nfull || *KL <= nfull) aborted = 1;          // a decision never got made: recompute
        const int row = nw - 1;
        int score = -1;
        if (row <= maxrow) {
            // which parity holds the row's latest value: processed in the last chunk (in band), or the
            // never-processed row created by the last bookkeeping; anything else is recomputed
            const int flast = CF[(int64_t)nfull * NA], llast = CL[(int64_t)nfull * NA];
            const int slot = row - (nfull - prolog);
            int par = -1;
            if (my_chunks == 0) par = 1;
            else if (slot >= flast && slot <= llast) par = (my_chunks - 1) & 1;
            else if (slot == llast + 1 && nfull > 0) par = (nfull - 1) & 1;
            if (par < 0) aborted = 1;
            else {
                score = S0[(par ? Spar : 0) + (int64_t)row * NA];
                if (m & 63) score -= 64 - (m & 63);
            }
        }
        A.o_score[t] = score;
        A.o_first[t] = CF[(int64_t)nfull * NA];
        A.o_last[t] = CL[(int64_t)nfull * NA];
        A.o_posv[t] = nfull - prolog;
        A.o_maxrow[t] = maxrow;
        A.o_adv[t] = adv;
        A.o_abort[t] = aborted;
    }
}
#undef QE_COOP_FENCE

// ===========================================================================
// The cooperative form with the band state ON CHIP (k_banded_coop_lds).  Same systolic pipeline as k_banded_coop --
// lane g of a group owns band slots [g c, (g + 1) c) and runs one chunk behind lane g-1 -- same band-edge protocol,
// same bits; what changes:
//   * Pv / Mv of every slot, the scores[] window (a ring over block rows, two parities) and the band-edge records
//     (rings over chunks) of a wave's 64 / G tasks live in LDS: no global round trip per (slot, chunk), and the
//     hand-over between neighbouring lanes (lane g+1 stores the slot that becomes lane g's last slot; lane g reads it
//     in a LATER pass of the same step) goes through LDS in program order of one wave -- ordered by the hardware, and
//     kept in that order by wavefront-scope fences for the compiler;
//   * a lane's c slots are walked in the SAME passes by every lane of the wave (c split into >= 2 passes of <= 4 slots),
//     each a skewed multi-slot pass (run64_skew): a slot that is not in its task's band at this chunk is carried along
//     masked instead of breaking the wave into one-slot passes.  A slot ABOVE the band's top runs on the fixed point
//     Pv = 0, Mv = ~0 with carry-in (1, 0): whatever Eq is, its carry-out is (1, 0) again and its state does not
//     change -- exactly the boundary carry PHin = 1 the band's top block takes (bpm_banded.c:238), at no instruction in
//     the inner loop; a slot BELOW the bottom computes values nobody reads.  The scores[] chain of a pass (slots_pass)
//     holds for any block state, so live rows get their exact sums whatever the dead slots below them did.
// The last block row (level mask), N symbols and the partial last chunk take the one-slot general loop as before.
// At the end the stopped band is written to the launch's global workspace in k_banded_coop's layout: the Hirschberg
// join reads it there.
// ===========================================================================
struct ScoreRing {            // scores[] of block row (slot + chunk - prolog) lives at ring index (slot + chunk) mod rr
    int kb, rr;
    __device__ __forceinline__ int at(int slot) const {
        int x = slot + kb;
        x -= (x >= rr) ? rr : 0;
        x += (x < 0) ? rr : 0;
        return x;
    }
};

struct FillSink {             // where a FILL pass leaves what the traceback reads (k_banded<true>'s layout, this task's column)
    uint4* cp;                // checkpoints: cp[(8 chunk + j) * cps + slot * 64]
    uint4* hw;                // carry-in words: hw[(chunk * gns + slot) * 64]
    int64_t cps; int gns;
};

template <int K>
__device__ __forceinline__ void coop_pass(bool on, int i0, int fk, int hi, int pos_v, int NA, u64* Pv, u64* Mv, const int32_t* Srd, int32_t* Swr,
                                          const ScoreRing R, const u64* pp, int p0, u64 T0, u64 T1, u64 hinP, u64 hinM, u64& houtP, u64& houtM, u32& adv) {
    u64 P[K], M[K], a[K], b[K];
    int sc[K], v0[K];
    bool live[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k;
        live[k] = on && i >= fk && i <= hi;
        P[k] = 0; M[k] = (on && i < fk) ? QE_ONES : 0;            // above the band: the fixed point; below / idle lane: anything
        a[k] = 0; b[k] = 0; sc[k] = 0;
        if (live[k]) {
            u64 nn;
            P[k] = Pv[i * NA]; M[k] = Mv[i * NA]; sc[k] = Srd[R.at(i) * NA];
            load_planes(pp, p0 + 64 * (i + pos_v), a[k], b[k], nn);
        }
        v0[k] = __popcll(P[k]) - __popcll(M[k]);
    }
    run64_skew<K>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM);
    int d = __popcll(houtP) - __popcll(houtM);                      // sum of the bottom-row deltas of slot k, from the lowest up
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        if (live[k]) {
            Swr[R.at(i0 + k) * NA] = sc[k] + d;
            Pv[(i0 + k - 1) * NA] = P[k]; Mv[(i0 + k - 1) * NA] = M[k];      // band shift (bpm_banded.c:903-909)
            adv += 64u;
        }
        d -= (__popcll(P[k]) - __popcll(M[k])) - v0[k];
    }
}

template <bool FILL>
__global__ __launch_bounds__(512) void k_banded_coop_lds(CoopLdsArgs X) {
    const CoopArgs& A = X.A;
    const int lane = threadIdx.x & 63, w = QE_GROUP_INDEX();
    const int G = A.G, lgG = X.lgG, NA = 64 >> lgG;
    if (w * NA >= A.T.ntasks) return;
    const int q = lane >> lgG, g = lane & (G - 1);
    const int t = w * NA + q;
    int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    // FILL: a task that arrives flagged is not this kernel's (the host leaves the leaves whose bands are too short for G
    // lanes to the one-lane kernel, which runs over the flagged tasks afterwards)
    if (FILL && pair >= 0 && A.o_abort[t] != 0) pair = -1;
    const bool valid = pair >= 0;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0, tfin = 0;
    const u64* pp = A.P.pl_p;
    const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        cut_in = A.T.cutoff[t]; tfin = FILL ? n : A.T.tfin[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair];
        tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const bool hasN = (fl & FLAG_HAS_N) != 0;
    const Geom GE = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    const bool fgeom = FILL || A.fill_geom != 0;                        // score-only over the fill's cells (BandedArgs::fill_geom)
    const int nsl = fgeom ? GE.ebb : ((GE.cutoff + 63) >> 6) + 1;       // the score-only passes use their own narrower band (bpm_banded.c:801-803)
    const int stop_row = fgeom ? nw - 1 : nw;                           // bpm_banded.c:295 / 917
    const int lvl_last = (m - 1) & 63;
    const int prolog = GE.prolog;
    // my slots, and the passes every lane of the wave walks them in
    const int grp = (w * NA) >> 6, lane_t = (w * NA + q) & 63;           // FILL: my task's group and column in the traceback's layout
    const int wns = FILL ? X.g_nslots[grp] : A.w_nslots[w];
    const int cper = max((wns + G - 1) >> lgG, 2);
    const int slo = g * cper, shi = slo + cper - 1;
    const int npass = max(2, (cper + 3) >> 2), kbase = cper / npass, kextra = cper - kbase * npass;

    // LDS of this wave: Pv[(ns+1)][NA] | Mv[(ns+1)][NA] | S[2][rr][NA] | CF[cr][NA] | CL[cr][NA] | KF[NA] | KL[NA]
    uint8_t* lb = (uint8_t*)qe_dyn_lds + (size_t)QE_WAVE_IN_BLOCK() * (size_t)X.lds_per_wave;
    const int ns = X.ns, rr = X.rr, crm = X.cr - 1;
    u64* const Pv = (u64*)lb + NA + q;                       lb += (size_t)(ns + 1) * NA * 8;      // slot -1 is addressable
    u64* const Mv = (u64*)lb + NA + q;                       lb += (size_t)(ns + 1) * NA * 8;
    int32_t* const S0 = (int32_t*)lb + q;                    lb += (size_t)2 * rr * NA * 4;
    const int Spar = rr * NA;
    volatile int16_t* const CF = (volatile int16_t*)lb + q;  lb += (size_t)X.cr * NA * 2;      // shared by the G lanes of a task
    volatile int16_t* const CL = (volatile int16_t*)lb + q;  lb += (size_t)X.cr * NA * 2;
    volatile int32_t* const KF = (volatile int32_t*)lb + q;  lb += (size_t)NA * 4;
    volatile int32_t* const KL = (volatile int32_t*)lb + q;
#define QE_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
    // FILL: checkpoints, carry words and band-edge records go where k_traceback reads them (k_banded<true>'s layout)
    FillSink FS; FS.cp = nullptr; FS.hw = nullptr; FS.cps = 0; FS.gns = 0;
    int16_t* gcf = nullptr; int16_t* gcl = nullptr;
    if (FILL) {
        const int gns = X.g_nslots[grp], gnch = X.g_nch[grp];
        const GroupWs GW = group_ws(X.gws, X.g_ws_off[grp], gns, X.g_nrows[grp], gnch);
        gcf = GW.cf + lane_t; gcl = GW.cl + lane_t;
        FS.cp = X.mat + X.g_mat_off[grp] + lane_t;
        FS.cps = (int64_t)gns * 64; FS.gns = gns;
        FS.hw = FS.cp + (int64_t)QE_CPC * gnch * FS.cps;
    }

    // bpm_reset_search (bpm_banded.c:180-197): slots 0 .. nsl-1, block rows 0 .. nsl-1 (row r at ring index r + prolog)
    if (valid) {
        for (int x = g; x < nsl; x += G) {
            Pv[x * NA] = QE_ONES;
            Mv[x * NA] = 0;
            int ri = x + prolog; ri -= (ri >= rr) ? rr : 0;
            S0[ri * NA] = 64 * (x + 1);
            S0[Spar + ri * NA] = 64 * (x + 1);
            if (FILL) FS.cp[(int64_t)x * 64] = make_uint4(~0u, ~0u, 0u, 0u);
        }
        if (g == 0) {
            CF[0] = (int16_t)prolog; CL[0] = (int16_t)(nsl - 1); *KF = 1; *KL = 1;
            if (FILL) { gcf[0] = (int16_t)prolog; gcl[0] = (int16_t)(nsl - 1); }
        }
    }
    QE_WAVE_FENCE();
    const int nfull = tfin >> 6, tail = tfin & 63;
    const int my_chunks = valid ? nfull + (tail ? 1 : 0) : 0;
    const int nsteps = wave_max(my_chunks > 0 ? my_chunks + G - 1 : 0);
    u64 lastP = 0, lastM = 0;          // carry words out of my last slot in the previous step
    u32 adv = 0;
    int maxrow = nsl - 1, aborted = 0;
    ScoreRing R; R.rr = rr; R.kb = (g == 0) ? 0 : rr - g;           // (s - g) mod rr at s = 0

    for (int s = 0; s < nsteps; ++s) {
        const u64 cinP = __shfl_up(lastP, 1), cinM = __shfl_up(lastM, 1);
        const int k = s - g;
        const bool on = valid && k >= 0 && k < my_chunks;
        const int ncols = (k < nfull) ? 64 : tail;
        const int pos_v = k - prolog;
        const int32_t* const Srd = S0 + (((k - 1) & 1) ? Spar : 0);     // values after chunk k-1
        int32_t* const Swr = S0 + ((k & 1) ? Spar : 0);                 // values after chunk k
        int fk = 0, lk = -1;
        if (on) {
            const int kf = *KF, kl = *KL;
            if (k < kf) fk = CF[(k & crm) * NA];
            else {
                // `first` of my chunk is not decided yet (the deciding lane runs behind me): see k_banded_coop
                const int prev = CF[((kf - 1) & crm) * NA];
                const int d = k - (kf - 1);
                if (shi < prev - d) fk = shi + 1;
                else if (d == 1) fk = (k - 1 < prolog) ? prev - 1 : prev;
                else { fk = prev; aborted = 1; }
            }
            lk = (k < kl) ? CL[(k & crm) * NA] : CL[((kl - 1) & crm) * NA];   // exact or an upper bound; my range end is the same
        }
        const int hi = min(lk, nw - 1 - pos_v);                          // rows >= nw are never computed (A.7(2))
        const int lo_l = max(slo, fk), hi_l = min(shi, hi);
        const bool any_live = on && hi_l >= lo_l;
        u64 T0 = 0, T1 = 0, TN = 0;
        if (any_live) load_planes(tp, t0 + 64 * k, T0, T1, TN);
        const int lastrow_slot = nw - 1 - pos_v;                         // the slot of the pattern's last block row in this chunk
        u64 cP = cinP, cM = cinM;                                        // carry words into the next slot of my walk
        int i0 = slo;
        for (int p = 0; p < npass; ++p) {
            const int K = kbase + (p < kextra ? 1 : 0);                  // the same in every lane of the wave
            const bool plive = any_live && i0 + K - 1 >= lo_l && i0 <= hi_l;
            const bool slowl = plive && (ncols != 64 || hasN || (lastrow_slot >= max(i0, lo_l) && lastrow_slot <= min(i0 + K - 1, hi_l)));
            // FILL: every slot leaves its checkpoints and carry words for the traceback -- one slot at a time (run64_fast's
            // STORE == 2, as in k_banded<true>; a multi-slot pass that also stores spills hundreds of registers): what the
            // cooperative form buys a fill is lanes, for launches of few leaves
            if (!FILL && !__any(slowl)) {
                const u64 hP = (i0 <= fk) ? QE_ONES : cP, hM = (i0 <= fk) ? 0 : cM;      // PHin = 1 into the band's top block
                switch (K) {
                    case 4: coop_pass<4>(on, i0, fk, hi, pos_v, NA, Pv, Mv, Srd, Swr, R, pp, p0, T0, T1, hP, hM, cP, cM, adv); break;
                    case 3: coop_pass<3>(on, i0, fk, hi, pos_v, NA, Pv, Mv, Srd, Swr, R, pp, p0, T0, T1, hP, hM, cP, cM, adv); break;
                    case 2: coop_pass<2>(on, i0, fk, hi, pos_v, NA, Pv, Mv, Srd, Swr, R, pp, p0, T0, T1, hP, hM, cP, cM, adv); break;
                    default: coop_pass<1>(on, i0, fk, hi, pos_v, NA, Pv, Mv, Srd, Swr, R, pp, p0, T0, T1, hP, hM, cP, cM, adv); break;
                }
            } else {
                for (int kk = 0; kk < K; ++kk) {
                    const int i = i0 + kk, r = i + pos_v;
                    const bool act = on && i >= fk && i <= hi;
                    u64 P = 0, M = 0, a = 0, b = 0, nn = 0;
                    int sc = 0;
                    if (act) {
                        P = Pv[i * NA]; M = Mv[i * NA]; sc = Srd[R.at(i) * NA];
                        load_planes(pp, p0 + 64 * r, a, b, nn);
                    }
                    const bool lastblk = (r == nw - 1);
                    const u64 hP = (i <= fk) ? QE_ONES : cP, hM = (i <= fk) ? 0 : cM;
                    u64 houtP = 0, houtM = 0, sP, sM;
                    uint4* st = nullptr; uint4* st_last = nullptr;
                    if (FILL) {
                        // the chunk's last column belongs to the NEXT chunk's slot numbering (bpm_banded.c:279-287)
                        st = FS.cp + (int64_t)(QE_CPC * k) * FS.cps + (int64_t)i * 64;
                        st_last = FS.cp + (int64_t)(QE_CPC * k + QE_CPC) * FS.cps + (int64_t)(i - 1) * 64;
                        if (i == 0) st_last = st + QE_CPC * FS.cps;            // slot -1 does not exist; dropped row, never read
                        if (act) FS.hw[((int64_t)k * FS.gns + i) * 64] = make_uint4(lo32(hP), hi32(hP), lo32(hM), hi32(hM));
                    }
                    const bool slow = act && (ncols != 64 || hasN || lastblk);
                    if (!__any(slow)) {
                        run64_fast<FILL ? 2 : 0, true>(P, M, a, b, T0, T1, hP, hM, houtP, houtM, act, st, FS.cps, st_last);
                        sP = houtP; sM = houtM;
                    } else {
                        run64_general<FILL ? 2 : 0>(P, M, a, b, nn, T0, T1, TN, hP, hM, houtP, houtM, sP, sM,
                                                    lastblk ? lvl_last : 63, act ? ncols : 0, true, st, FS.cps, st_last);
                    }
                    if (act) {
                        sc += __popcll(sP) - __popcll(sM);
                        Swr[R.at(i) * NA] = sc;
                        const int dst = (ncols == 64) ? i - 1 : i;
                        Pv[dst * NA] = P;
                        Mv[dst * NA] = M;
                        adv += (u32)ncols;
                        cP = houtP; cM = houtM;
                    }
                    QE_WAVE_FENCE();
                }
            }
            QE_WAVE_FENCE();          // lane g+1's first pass stored the slot lane g reads in its last pass
            i0 += K;
        }
        if (on) { lastP = cP; lastM = cM; }
        // ---- band-edge decisions of every chunk that completed its deciding slot in this step (as in k_banded_coop)
        const bool full = on && ncols == 64;
        for (int round = 0; round < 6; ++round) {
            bool decided = false;
            if (full) {
                // top (bpm_banded.c:889-901): decided by the owner of slot first + 1
                int kf = *KF;
                if (kf == k + 1) {
                    const int f = CF[(k & crm) * NA];
                    const int dslot = min(f + 1, nsl - 1);
                    if (dslot >= slo && dslot <= shi) {
                        const int kl = *KL;
                        const int lub = CL[(min(k, kl - 1) & crm) * NA];
                        const int llb = (k < kl) ? lub : lub - (k - (kl - 1));     // last drops by at most one per chunk
                        bool tall;
                        if (f + 2 < llb) tall = true;
                        else if (f + 2 >= lub) tall = false;
                        else { tall = true; aborted = 1; }
                        bool cut_lo = false;
                        if (tall && GE.fin > 64 * (f + 1))
                            cut_lo = Swr[R.at(f + 1) * NA] + (GE.fin - 64 * (f + 1)) > GE.cutoff;
                        int fnew = f;
                        if (cut_lo && k >= prolog) fnew = f + 1;
                        else if (!cut_lo && k < prolog) fnew = f - 1;
                        CF[((k + 1) & crm) * NA] = (int16_t)fnew;
                        if (FILL) gcf[(int64_t)(k + 1) * 64] = (int16_t)fnew;
                        *KF = k + 2;
                        decided = true;
                    }
                }
                QE_WAVE_FENCE();
                // bottom (bpm_banded.c:903-921): decided by the owner of slot last, once the new first is known
                kf = *KF;
                const int kl = *KL;
                if (kl == k + 1 && kf >= k + 2) {
                    const int l = CL[(k & crm) * NA];
                    if (l >= slo && l <= shi) {
                        const int fnew = CF[((k + 1) & crm) * NA];
                        Pv[l * NA] = QE_ONES;
                        Mv[l * NA] = 0;
                        if (FILL) FS.cp[(int64_t)(QE_CPC * k + QE_CPC) * FS.cps + (int64_t)l * 64] = make_uint4(~0u, ~0u, 0u, 0u);
                        const int pos = l + pos_v;
                        Swr[R.at(l + 1) * NA] = Swr[R.at(l) * NA] + 64;
                        maxrow = max(maxrow, pos + 1);
                        bool cut_hi = false;
                        if ((fnew + 2 < l) && (64 * (l - 1) > GE.fin))
                            cut_hi = Swr[R.at(l - 1) * NA] + (64 * (l - 1) - GE.fin) > GE.cutoff;
                        const int lnew = (cut_hi || (pos_v + l >= stop_row)) ? l - 1 : l;
                        CL[((k + 1) & crm) * NA] = (int16_t)lnew;
                        if (FILL) gcl[(int64_t)(k + 1) * 64] = (int16_t)lnew;
                        *KL = k + 2;
                        decided = true;
                    }
                }
            }
            QE_WAVE_FENCE();
            if (!__any(decided)) break;
        }
        // a wrong "no cut" guess before the end of the prologue put the carry chain on the wrong top slot
        if (on && k >= 1 && k <= prolog && fk <= shi) {
            const int kf = *KF;
            if (kf > k && CF[(k & crm) * NA] != fk) aborted = 1;
        }
        R.kb = (R.kb + 1 == rr) ? 0 : R.kb + 1;
    }
    QE_WAVE_FENCE();
    // group reductions: adv (sum), maxrow (max), aborted (or)
    for (int o = 1; o < G; o <<= 1) {
        adv += __shfl_xor(adv, o);
        maxrow = max(maxrow, __shfl_xor(maxrow, o));
        aborted |= __shfl_xor(aborted, o);
    }
    // ---- the stopped band goes to the launch's global workspace, k_banded_coop's layout (ColDist reads it there)
    if (!FILL) {
        const int gns = wns, gnr = A.w_nrows[w];
        uint8_t* base = A.ws + A.w_ws_off[w];
        u64* const gP = (u64*)base + NA + q;                                   base += (int64_t)(gns + 1) * NA * 8;
        u64* const gM = (u64*)base + NA + q;                                   base += (int64_t)(gns + 1) * NA * 8;
        int32_t* const gS = (int32_t*)base + q;
        const int posv_end = nfull - prolog;
        int kbe = nfull % rr;                                                   // ring offset of chunk nfull's numbering
        if (valid) {
            for (int i = slo; i <= min(shi, min(nsl - 1, gns - 1)); ++i) {
                gP[(int64_t)i * NA] = Pv[i * NA];
                gM[(int64_t)i * NA] = Mv[i * NA];
                const int row = i + posv_end;
                if (row >= 0 && row < gnr) {
                    int ri = i + kbe; ri -= (ri >= rr) ? rr : 0;
                    gS[(int64_t)row * NA] = S0[ri * NA];
                    gS[(int64_t)gnr * NA + (int64_t)row * NA] = S0[Spar + ri * NA];
                }
            }
        }
    }
    if (valid && g == 0) {
        if (*KF <= nfull || *KL <= nfull) aborted = 1;          // a decision never got made: recompute
        const int row = nw - 1;
        int score = -1;
        const int flast = CF[(nfull & crm) * NA], llast = CL[(nfull & crm) * NA];
        if (row <= maxrow) {
            // which parity holds the row's latest value: processed in the last chunk (in band), or the
            // never-processed row created by the last bookkeeping; anything else is recomputed
            const int slot = row - (nfull - prolog);
            int par = -1;
            if (my_chunks == 0) par = 1;
            else if (slot >= flast && slot <= llast) par = (my_chunks - 1) & 1;
            else if (slot == llast + 1 && nfull > 0) par = (nfull - 1) & 1;
            if (par < 0) aborted = 1;
            else {
                const int ri = (row + prolog) % rr;
                score = S0[(par ? Spar : 0) + ri * NA];
                if (m & 63) score -= 64 - (m & 63);
            }
        }
        A.o_score[t] = score;
        A.o_first[t] = flast;
        A.o_last[t] = llast;
        A.o_posv[t] = nfull - prolog;
        A.o_maxrow[t] = maxrow;
        A.o_adv[t] = adv;
        A.o_abort[t] = aborted;
    }
#undef QE_WAVE_FENCE
}
template __global__ void k_banded_coop_lds<false>(CoopLdsArgs);
template __global__ void k_banded_coop_lds<true>(CoopLdsArgs);

// ===========================================================================
// BandEd score-only, ONE WAVEFRONT PER ALIGNMENT (BASELINE.json's form; for few alignments: a single pair, a
// handful of long reads -- where one lane per alignment leaves the chip empty and the latency of one lane is the run
// time).  Lane j owns the block rows r = j (mod 64); row r works on text column c = step - r, so the rows of the band
// are a systolic array skewed by one column per row: (r, c) needs (r - 1, c) -- the neighbour lane's carry bits of
// the previous step, one v_mov_dpp wave_ror:1 each -- and (r, c - 1), the lane's own registers.  State never leaves
// the registers: a row's Pv / Mv / scores[] entry / pattern planes live in its lane from the chunk it enters the band
// (the reference's "new bottom block", bpm_banded.c:903-912) until the band leaves it behind.
// The 64-column band bookkeeping (bpm_banded.c:889-922) is scalar code that runs when its inputs are complete: the top
// rule of chunk k once row first+1 has finished the chunk (one step later than the old top row would start chunk k+1: that
// row speculates "no cut" while the prologue lasts, which is harmless -- a cut drops it and the row below takes the
// boundary carry), the bottom rule once row `last` has.  scores[] entries the rules read come by v_readlane.
// Same cells, same values as k_banded<false>; needs the band to fit the wave (<= 62 slots).
// ===========================================================================
__device__ __forceinline__ int dpp_ror1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x13C, 0xf, 0xf, false); }   // lane i <- lane i-1, wraps
// a ring of four band-edge values (slot numbers, < 256) packed into one register: an indexed private array -- and a
// select chain, which the compiler turns into one -- would live in scratch memory
struct Ring4 {
    u32 v;
    __device__ __forceinline__ int get(int k) const { return (int)((v >> (8 * (k & 3))) & 0xffu); }
    __device__ __forceinline__ void set(int k, int x) { const int sh = 8 * (k & 3); v = (v & ~(0xffu << sh)) | ((u32)x << sh); }
};

__global__ __launch_bounds__(256) void k_banded_wave(BandedArgs A) {
    const int t = __builtin_amdgcn_readfirstlane(QE_GROUP_INDEX()), lane = threadIdx.x & 63;     // one task per wave: uniform
    if (t >= A.T.ntasks) return;
    const int pair = A.T.pair[t];
    if (pair < 0) return;
    const int m = A.T.m[t], n = A.T.n[t], p0 = A.T.p0[t], t0 = A.T.t0[t], cut_in = A.T.cutoff[t], tfin = A.T.tfin[t];
    const u64* __restrict__ pp = A.P.pl_p + A.P.pl_p_off[pair];
    const u64* __restrict__ tp = A.P.pl_t + A.P.pl_t_off[pair];
    (void)n;
    const Geom G0 = band_geometry(m, A.T.n[t], cut_in);
    const int g_cutoff = G0.cutoff, g_fin = G0.fin, prolog = G0.prolog;      // scalars, not a struct the lambda below would keep in memory
    const int nw = (m + 63) >> 6, nsl = ((g_cutoff + 63) >> 6) + 1, lvl_last = (m - 1) & 63;
    const int nfull = tfin >> 6;
    // band edges (slot numbers first / last, as in the reference) of chunk k at index k & 3: at most two chunks are alive
    // among the rows of the band and decisions run one chunk ahead
    Ring4 fr{(u32)prolog}, lr{(u32)(nsl - 1)};
    int dec_top = 1, dec_bot = 1;              // edges are decided for chunks < dec_top / < dec_bot
    int maxrow = nsl - 1;
    // lane state: the row I hold
    int my_row = lane;
    bool created = lane <= nsl - 1 - prolog;   // bpm_reset_search: slots prolog .. nsl-1 are rows 0 .. nsl-1-prolog
    u64 P = QE_ONES, M = 0;
    int sc = 64 * (lane + 1);                  // scores[r] = 64 (r + 1) (bpm_banded.c:180-197)
    int sc_end = sc;                           // ... as it stood at the end of my row's last completed chunk: the rows above the
                                               // deciding row are already a few columns into the next chunk when a rule reads them
    u64 a = 0, b = 0, nn = 0, an = 0, bn = 0, nnn = 0;     // pattern planes of my row / of the row I will hold next
    if (created && my_row < nw) load_planes(pp, p0 + 64 * my_row, a, b, nn);
    if (lane == ((nsl - prolog) & 63) && nsl - prolog < nw) load_planes(pp, p0 + 64 * (nsl - prolog), an, bn, nnn);
    int hP = 0, hM = 0;                        // my carry-out bits of the previous step
    int sym = 0;                               // my text symbol of the previous step: code bit 0, code bit 1, not-ACGT
    u32 adv = 0;
    bool in_band = false, is_top = false;
    // the text is read by ONE row, the top of the band (scalar loads of its chunk's plane words, the next chunk's
    // prefetched); every other row takes its symbol from the row above, one step later, like its carries
    int kq = -2;                               // chunk whose words Tc holds (-2: none, and not the predecessor of chunk 0)
    u64 Tc0 = 0, Tc1 = 0, TcN = 0, Tn0 = 0, Tn1 = 0, TnN = 0;
    auto text_words = [&](int k, u64& w0, u64& w1, u64& wn) {          // uniform k: the planes of text chunk k
        w0 = 0; w1 = 0; wn = 0;
        if (64 * k < tfin) {
            const int bitpos = t0 + 64 * k, w = bitpos >> 6, sh = bitpos & 63;
            const u64* q = tp + 3 * (int64_t)w;
            w0 = q[0]; w1 = q[1]; wn = q[2];
            if (sh) { w0 = (w0 >> sh) | (q[3] << (64 - sh)); w1 = (w1 >> sh) | (q[4] << (64 - sh)); wn = (wn >> sh) | (q[5] << (64 - sh)); }
        }
    };

    auto decide = [&](int step) {
        for (;;) {
            bool did = false;
            {   // top rule of chunk k -> first of chunk k + 1 (bpm_banded.c:889-901)
                const int k = dec_top - 1;
                if (k < nfull && dec_bot > k) {
                    const int f = fr.get(k), l = lr.get(k), rlo = f + k - prolog;
                    if (step >= 64 * k + 65 + rlo) {
                        bool cut_lo = false;
                        if ((f + 2 < l) && (g_fin > 64 * (f + 1)))
                            cut_lo = __builtin_amdgcn_readlane(sc_end, (rlo + 1) & 63) + (g_fin - 64 * (f + 1)) > g_cutoff;
                        int fnew = f;
                        if (cut_lo && k >= prolog) fnew = f + 1;
                        else if (!cut_lo && k < prolog) fnew = f - 1;
                        fr.set(k + 1, fnew);
                        ++dec_top;
                        did = true;
                    }
                }
            }
            {   // bottom rule of chunk k -> the new bottom row, last of chunk k + 1 (bpm_banded.c:903-921)
                const int k = dec_bot - 1;
                if (k < nfull && dec_top > k + 1) {
                    const int l = lr.get(k), pos_v = k - prolog, pos = l + pos_v;
                    if (step >= 64 * k + 64 + pos) {
                        const int spos = __builtin_amdgcn_readlane(sc_end, pos & 63);
                        const int rb = pos + 1;
                        if (lane == (rb & 63)) {
                            my_row = rb; created = true; P = QE_ONES; M = 0; sc = spos + 64; sc_end = sc; in_band = false;
                            a = an; b = bn; nn = nnn;
                        }
                        if (lane == ((rb + 1) & 63)) { an = 0; bn = 0; nnn = 0; if (rb + 1 < nw) load_planes(pp, p0 + 64 * (rb + 1), an, bn, nnn); }
                        maxrow = max(maxrow, rb);
                        const int fnew = fr.get(k + 1);
                        bool cut_hi = false;
                        if ((fnew + 2 < l) && (64 * (l - 1) > g_fin))
                            cut_hi = __builtin_amdgcn_readlane(sc_end, (pos - 1) & 63) + (64 * (l - 1) - g_fin) > g_cutoff;
                        lr.set(k + 1, (cut_hi || (pos_v + l >= nw)) ? l - 1 : l);
                        ++dec_bot;
                        did = true;
                    }
                }
            }
            if (!did) break;
        }
    };
    // the step at which the next rule falls due (one of the two is always eligible while any is pending)
    auto next_due = [&]() {
        int due = 0x7fffffff;
        const int k1 = dec_top - 1, k2 = dec_bot - 1;
        if (k1 < nfull && dec_bot > k1) due = min(due, 64 * k1 + 65 + fr.get(k1) + k1 - prolog);
        if (k2 < nfull && dec_top > k2 + 1) due = min(due, 64 * k2 + 64 + lr.get(k2) + k2 - prolog);
        return __builtin_amdgcn_readfirstlane(due);
    };

    const bool hasN = (A.P.flags[pair] & FLAG_HAS_N) != 0;               // uniform: N / IUPAC symbols take the general Eq
    int due = next_due();
    const int steps = tfin + nw;               // row r runs column tfin - 1 at step tfin - 1 + r, r <= nw - 1
    for (int step = 0; step < steps; ++step) {
        if (step >= due) { decide(step); due = next_due(); }
        const int inP = dpp_ror1(hP), inM = dpp_ror1(hM), relay = dpp_ror1(sym);
        const int c = step - my_row;
        const bool valid = created && my_row < nw && (u32)c < (u32)tfin;
        const int bit = c & 63;
        if (__any(valid && bit <= 1)) {
            // entering a chunk (and the step after, when the one decision a row can be ahead of has landed): am I still in
            // the band, am I its top row
            const int kk = max(c >> 6, 0);
            const int f = (kk < dec_top) ? fr.get(kk) : ((kk - 1 < prolog) ? fr.get(kk - 1) - 1 : fr.get(kk - 1));   // undecided: only the old top row
            const int rlo = f + kk - prolog;
            // bottom edge: only the row the bookkeeping has just created can lie below it (cut_hi / the stop rule keep
            // the band from growing, bpm_banded.c:913-919), and that row enters its first chunk after the decision
            const bool below = kk < dec_bot && my_row > lr.get(kk) + kk - prolog;
            const bool ent = valid && bit <= 1;
            in_band = ent ? (my_row >= rlo && !below) : in_band;
            is_top = ent ? (my_row == rlo) : is_top;
        }
        const bool work = valid && in_band;
        const u64 topmask = __ballot(work && is_top);
        int fresh = 0;
        if (topmask != 0) {                                            // uniform
            const int tl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)topmask) - 1);
            const int ctop = step - __builtin_amdgcn_readlane(my_row, tl);
            const int ka = __builtin_amdgcn_readfirstlane(ctop >> 6), tb = __builtin_amdgcn_readfirstlane(ctop & 63);
            if (ka != kq) {
                if (ka == kq + 1) { Tc0 = Tn0; Tc1 = Tn1; TcN = TnN; }
                else text_words(ka, Tc0, Tc1, TcN);
                kq = ka;
                text_words(ka + 1, Tn0, Tn1, TnN);                       // needed 64 steps from now
            }
            fresh = (int)((Tc0 >> tb) & 1) | ((int)((Tc1 >> tb) & 1) << 1) | ((int)((TcN >> tb) & 1) << 2);
        }
        sym = (work && is_top) ? fresh : relay;
        hP = 0; hM = 0;
        if (work) {
            const u32 PHin = is_top ? 1u : (u32)inP, MHin = is_top ? 0u : (u32)inM;
            u32 Plo = lo32(P), Phi = hi32(P), Mlo = lo32(M), Mhi = hi32(M), phhi, mhhi, phlo_bit = 0, mhlo_bit = 0;
            const bool lastrow = my_row == nw - 1;                       // level_mask of the last block (bpm_banded.c:88-102)
            if (!hasN) {
                const u32 m0 = (u32)__builtin_amdgcn_sbfe(sym, 0, 1), m1 = (u32)__builtin_amdgcn_sbfe(sym, 1, 1);
                const u32 elo = bitop3<0x90>(~(lo32(a) ^ m0), lo32(b), m1), ehi = bitop3<0x90>(~(hi32(a) ^ m0), hi32(b), m1);
                if (__any(lastrow && lvl_last != 63)) {
                    // the row whose exported bit is not bit 63 needs the whole horizontal delta words
                    u64 Ph, Mh;
                    block_step(mk64(elo, ehi), P, M, PHin, MHin, Ph, Mh);
                    Plo = lo32(P); Phi = hi32(P); Mlo = lo32(M); Mhi = hi32(M);
                    phhi = hi32(Ph); mhhi = hi32(Mh);
                    const int lv = lastrow ? lvl_last : 63;
                    phlo_bit = (u32)((Ph >> lv) & 1); mhlo_bit = (u32)((Mh >> lv) & 1);
                } else {
                    block_step_core(elo, ehi, Plo, Phi, Mlo, Mhi, PHin, MHin, phhi, mhhi);
                    phlo_bit = phhi >> 31; mhlo_bit = mhhi >> 31;
                }
            } else {
                const u64 m0 = (u64)0 - (u64)(sym & 1), m1 = (u64)0 - (u64)((sym >> 1) & 1);
                const u64 Eq = (sym & 4) ? nn : (~(a ^ m0) & ~(b ^ m1) & ~nn);
                u64 Ph, Mh;
                block_step(Eq, P, M, PHin, MHin, Ph, Mh);
                Plo = lo32(P); Phi = hi32(P); Mlo = lo32(M); Mhi = hi32(M);
                phhi = hi32(Ph); mhhi = hi32(Mh);
                const int lv = lastrow ? lvl_last : 63;
                phlo_bit = (u32)((Ph >> lv) & 1); mhlo_bit = (u32)((Mh >> lv) & 1);
            }
            P = mk64(Plo, Phi); M = mk64(Mlo, Mhi);
            sc += (int)phlo_bit - (int)mhlo_bit;
            hP = (int)(phhi >> 31); hM = (int)(mhhi >> 31);
            if (bit == 63 || c == tfin - 1) { adv += (u32)(bit + 1); sc_end = sc; }      // a completed chunk: its block-advances, its score
        }
    }
    decide(0x7fffffff);                        // the bookkeeping after the last full chunk (its new row counts for the read-out)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) adv += __shfl_xor(adv, o);
    const int row = nw - 1;
    int score = -1;
    if (row <= maxrow) {
        score = __builtin_amdgcn_readlane(sc, row & 63);
        if (m & 63) score -= 64 - (m & 63);
    }
    if (lane == 0) {
        A.o_score[t] = score;
        A.o_first[t] = fr.get(nfull);
        A.o_last[t] = lr.get(nfull);
        A.o_posv[t] = nfull - prolog;
        A.o_maxrow[t] = maxrow;
        A.o_adv[t] = adv;
    }
}

// ===========================================================================
// BandEd with SIXTEEN (or 64) LANES PER ALIGNMENT (k_banded_sys): the cooperative form of k_banded<true> / <false> for
// launches of few waves (one batch of a few thousand QuickEd pairs, a single pair), where one lane's serial chain over
// the 6-10 slots of a band pruned to QuickEd's bound is what the launch lasts (bpm_banded.c:199-316).
// Lane j of a 16-lane group owns the block row r with r mod 16 = j that lies in the band (a band of <= 15 slots has at
// most one such row): the row's Pv / Mv, its scores[] entry and its pattern planes stay in the lane's registers from the
// chunk the bookkeeping creates the row (bpm_banded.c:903-912) until the band has moved past it -- the reference's
// 64-column band shift moves nothing here.  Inside a chunk the rows form a systolic array skewed by one column per row
// (row first + i works on column s - i at step s, taking its two carry bits from the lane above by v_mov_dpp row_ror:1;
// the lane above the top row holds the constant boundary carry (1, 0), bpm_banded.c:238): 64 + H - 1 steps of ONE block
// step per chunk instead of H x 64, and all rows meet at the chunk's end, where the reference's own bookkeeping
// (bpm_banded.c:264-301) runs for the whole group exactly as written -- no band-edge protocol between lanes running
// chunks apart (k_banded_coop), which is what tight bands broke.  scores[]: a chunk's sum of a row's exported
// horizontal deltas is the sum of its carry-ins plus the change of its vertical-delta sum (cell identity, cf.
// slots_pass), masked to the pattern's last row in the last block (level_mask, bpm_banded.c:88-102) -- one prefix sum
// over the band's rows per chunk instead of two adds per step.
// What the traceback reads is written in k_banded<true>'s layout (checkpoints every QE_CP_COLS columns captured in
// registers when a row passes them, stored at four fixed steps of the chunk; carry-in words per (chunk, slot); band
// edges per chunk), so k_traceback runs unchanged.  Tasks with N or a taller band are flagged (o_abort) for k_banded<true>.
// ===========================================================================
template <int LG>
__device__ __forceinline__ u32 grp_ror1(u32 x) {            // lane G g + j <- lane G g + (j + G - 1) % G;  G = 16: row_ror:1, G = 64: wave_ror:1
    return (u32)__builtin_amdgcn_mov_dpp((int)x, LG == 4 ? 0x121 : 0x13C, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ int row_ror_n(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x120 + N, 0xf, 0xf, false); }
// inclusive prefix sum over the 64 lanes of a wave in lane order, DPP only: row_shr 1 / 2 / 4 / 8 inside the rows of 16, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (six DPP moves: a chain of six ds_bpermute round trips
// -- ~100 cycles each for a lone wave -- is what the per-chunk bookkeeping of a wave-per-task kernel used to wait for)
__device__ __forceinline__ int wave_scan_add(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// LG = 4: sixteen lanes per task, four tasks per wave (bands of <= 15 slots: QuickEd's tight bounds); LG = 6: one task per
// wave (bands of <= 63 slots: the bounds of pairs with large indels, a user bandwidth).  FILL = false is the score-only
// pass (bpm_banded.c:791-964: its own narrower band and stop rule) for whole texts -- QuickEd's stage-3 doubling rounds, a
// BandEd score-only call on few pairs; a pass that stops early (a Hirschberg half pass exports its band) is flagged.
template <int LG, bool FILL>
// (occupancy is not what bounds a launch of many waves here: 12.5 k leaves at 16 lanes each put three to four waves on a SIMD,
// every one adds ~0.6 ms to the 1.3 ms of a lone wave -- 3.1 ms; forcing <= 128 VGPRs changed nothing)
__global__ __launch_bounds__(256) void k_banded_sys(BandedArgs A) {
    if (A.prio) __builtin_amdgcn_s_setprio(3);          // few waves, each a serial chain: first in line at the SIMD's issue arbiter
    constexpr int GL = 1 << LG, NT = 64 >> LG, GM = GL - 1;
    const int wv = QE_GROUP_INDEX(), lane = threadIdx.x & 63, j = lane & GM, gl = lane & ~GM;
    const int t = wv * NT + (lane >> LG);
    if (wv * NT >= A.T.ntasks) return;
    int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    if (A.only_if != nullptr && pair >= 0 && A.only_if[t] == 0) pair = -1;
    const bool valid = pair >= 0;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0, tfin = 1;
    const u64* pp = A.P.pl_p;
    const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        cut_in = A.T.cutoff[t];
        tfin = FILL ? n : A.T.tfin[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair];
        tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const int nw = (m + 63) >> 6;
    const bool fgeom = FILL || A.fill_geom != 0;                   // score-only over the fill's cells (BandedArgs::fill_geom)
    const int stop_row = fgeom ? nw - 1 : nw;                      // bpm_banded.c:295 / 917
    const u64 lvl_mask = (m & 63) ? (((u64)1 << (m & 63)) - 1) : QE_ONES;     // rows of the last block up to the pattern's end
    u32 adv = 0;
    // A.doubling (score-only): QuickEd's stage-3 loop (quicked.c:248-278) on the device -- while the pass's score says the
    // cutoff was too small ((score > max_len / 4 && cutoff * 3 / 2 < score) || score < 0) the cutoff doubles and the pass runs
    // again, in this launch, every task at its own pace; a task whose doubled band no longer fits the group's lanes is
    // handed back with the cutoff it was about to run (o_cutoff) and the block-advances it has counted so far
    int cut_cur = cut_in;
    bool want = valid;
    for (;;) {
    const Geom G = band_geometry(m, n, cut_cur);
    const int nsl = fgeom ? G.ebb : ((G.cutoff + 63) >> 6) + 1;   // the score-only kernels' own band (bpm_banded.c:801-803)
    const bool elig = (fl & FLAG_HAS_N) == 0 && nsl <= GM && tfin == n;
    const bool ok = want && elig;
    if (want && !elig) {
        u32 asum = adv;
#pragma unroll
        for (int o = GL / 2; o > 0; o >>= 1) asum += __shfl_xor(asum, o);
        if (j == 0) {
            A.o_abort[t] = 1;
            if (A.o_cutoff) { A.o_cutoff[t] = cut_cur; A.o_adv[t] = asum; }
        }
        want = false;
    }
    if (!__any(ok)) break;
    int first = G.prolog, last = nsl - 1, pos_v = -G.prolog, pos_h = 0;
    int max_row_init = nsl - 1;

    const int g = ok ? (t >> 6) : 0, col = t & 63;
    int gns = 0, gnch = 0;
    int16_t* cf = nullptr; int16_t* cl = nullptr;
    uint4* cp = nullptr; uint4* hw = nullptr;
    int64_t cps = 0;
    if (FILL) {
        gns = A.g_nslots[g]; gnch = A.g_nch[g];
        const GroupWs W = group_ws(A.ws, A.g_ws_off[g], gns, A.g_nrows[g], gnch);
        cf = W.cf + col; cl = W.cl + col;
        cp = A.mat + A.g_mat_off[g] + col;
        cps = (int64_t)gns * 64;
        hw = cp + (int64_t)QE_CPC * gnch * cps;
    }

    // bpm_reset_search (bpm_banded.c:180-197): lane j holds row j
    u32 Plo = ~0u, Phi = ~0u, Mlo = 0, Mhi = 0;
    int sc = 64 * (j + 1);
    u64 pa = 0, pb = 0;                                            // pattern planes of this lane's row
    if (ok) {
        if (FILL && j < nsl) cp[(int64_t)j * 64] = make_uint4(~0u, ~0u, 0u, 0u);
        if (FILL && j == 0) { cf[0] = (int16_t)first; cl[0] = (int16_t)last; }
        if (j < nw) load_planes_ab(pp, p0 + 64 * j, pa, pb);
    }
    const int nfull = n >> 6, tail = n & 63;
    const int my_chunks = ok ? nfull + (tail ? 1 : 0) : 0;
    const int wave_chunks = __builtin_amdgcn_readfirstlane(wave_max(my_chunks));
    // text planes of the next chunk, loaded a chunk ahead
    u64 nT0 = 0, nT1 = 0;
    if (ok) load_planes_ab(tp, t0, nT0, nT1);

    for (int k = 0; k < wave_chunks; ++k) {
        const int ncols = (k < nfull) ? 64 : ((k == nfull) ? tail : 0);
        const bool on = ok && ncols > 0;
        const u64 T0 = nT0, T1 = nT1;
        if (ok && k + 1 < my_chunks) load_planes_ab(tp, t0 + 64 * (k + 1), nT0, nT1);
        const int r_first = first + pos_v;
        const int rhi = min(last, nw - 1 - pos_v);                 // rows >= nw are never computed (A.7(2))
        const int i = (j - r_first) & GM;                          // this lane's row of the band, counted from its top
        const int si = first + i;                                  // its slot
        const bool inband = on && si <= rhi;
        const int my_row = r_first + i;
        // the row the bookkeeping will create at this chunk's end: its lane loads its planes now
        u64 qa = 0, qb = 0;
        const int new_row = last + pos_v + 1;
        if (on && ncols == 64 && ((new_row & GM) == j) && new_row < nw) load_planes_ab(pp, p0 + 64 * new_row, qa, qb);
        // (a wave per task: every lane holds the same band; four tasks per wave: the tallest of the four)
        const int hm_own = on ? rhi - first + 1 : 0;
        const int Hm = (LG == 6) ? __builtin_amdgcn_readfirstlane(hm_own)
                                 : max(max(__builtin_amdgcn_readlane(hm_own, 0), __builtin_amdgcn_readlane(hm_own, 16)),
                                       max(__builtin_amdgcn_readlane(hm_own, 32), __builtin_amdgcn_readlane(hm_own, 48)));
        // bit s of these is the text's column s - i (mod 64)
        const int ri = i & 63;
        const u64 R0 = ri ? ((T0 << ri) | (T0 >> (64 - ri))) : T0, R1 = ri ? ((T1 << ri) | (T1 >> (64 - ri))) : T1;
        const u64 vm = (my_row == nw - 1) ? lvl_mask : QE_ONES;
        const int Vb = __popcll(mk64(Plo, Phi) & vm) - __popcll(mk64(Mlo, Mhi) & vm);
        const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
        u64 gP = 0, gM = 0;                                        // carry-ins of this row, shifted in from the right
        u32 oP = 0, oM = 0;
        if (i == GM) { oP = 1u; oM = 0u; }                         // the lane above the top row: PHin = 1 into the band (bpm_banded.c:238)
        u32 cPlo = Plo, cPhi = Phi, cMlo = Mlo, cMhi = Mhi;        // G = 16: the checkpoint this row passed last
        const u32 len = inband ? (u32)ncols : 0u;
        const int nsteps = 64 + Hm - 1;
        // (not in the fill of tall bands: its per-step checkpoint stores in one straight block cost 60 registers more;
        // lane_rel = 2: tests switch the steady blocks off)
        const bool steady_ok = !(FILL && LG == 6) && !__any(on && ncols != 64) && A.lane_rel != 2;
        // where this row's checkpoints go: q = 1 .. 3 in this chunk's slot numbering, q = 4 = the next chunk's checkpoint 0
        // one slot up (the band shift, bpm_banded.c:279-287; slot -1 does not exist and is never read)
        uint4* const cpk = FILL ? cp + (int64_t)(QE_CPC * k) * cps + (int64_t)si * 64 : nullptr;
#pragma unroll 1
        for (int blk = 0; blk < (LG == 4 ? 3 : 4); ++blk) {
            if (32 * blk >= nsteps) break;
            const u32 w0 = (blk & 1) ? hi32(R0) : lo32(R0), w1 = (blk & 1) ? hi32(R1) : lo32(R1);
            // 16 steps (half a block); `whole`: all of them are inside the chunk's 64 + H - 1, so no step asks -- three scalar
            // instructions per step of a lone wave's chain, 5 % of the kernel
            // `steady`: sixteen steps from H - 1 on and before 64, in a chunk whose columns are all there -- every row of every
            // band is at work in every one of them, so the block runs under ONE mask (the rows in their bands) and
            // no step asks whether its lane is inside its row's columns (two vector and four scalar instructions per step).
            // The carries come by DPP as ever; the top row's source -- the idle lane above it -- is masked out there, a DPP
            // move leaves such a lane's destination as it was, and that is made the boundary carry (1, 0)
            auto steps16 = [&](auto whole_tag, auto steady_tag, auto half_tag) {
            constexpr bool whole = decltype(whole_tag)::value, steady = decltype(steady_tag)::value;
            constexpr int half = decltype(half_tag)::value;
#pragma unroll
            for (int sb = 16 * half; sb < 16 * half + 16; ++sb) {
                const int s = 32 * blk + sb;
                if (!whole && s >= nsteps) continue;               // (uniform; a break would keep the loop rolled)
                const u32 inP = steady ? (u32)__builtin_amdgcn_update_dpp(1, (int)oP, LG == 4 ? 0x121 : 0x13C, 0xf, 0xf, false) : grp_ror1<LG>(oP);
                const u32 inM = steady ? (u32)__builtin_amdgcn_update_dpp(0, (int)oM, LG == 4 ? 0x121 : 0x13C, 0xf, 0xf, false) : grp_ror1<LG>(oM);
                const u32 c = (u32)(s - i);
                if (steady || c < len) {
                    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)w0, sb, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)w1, sb, 1);
                    const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                    u32 phhi, mhhi;
                    block_step_core(elo, ehi, Plo, Phi, Mlo, Mhi, inP, inM, phhi, mhhi);
                    oP = phhi >> 31; oM = mhhi >> 31;
                    if (FILL) {
                        gP = shl1_add_u64(gP, (u64)inP); gM = shl1_add_u64(gM, (u64)inM);
                        const bool cap = (c & (QE_CP_COLS - 1)) == QE_CP_COLS - 1;
                        if (LG == 4) {
                            cPlo = cap ? Plo : cPlo; cPhi = cap ? Phi : cPhi; cMlo = cap ? Mlo : cMlo; cMhi = cap ? Mhi : cMhi;
                        } else if (cap && c != 63) {
                            // a tall band's rows pass their checkpoints up to 62 steps apart: stored as they come
                            cpk[(int64_t)((c >> 4) + 1) * cps] = make_uint4(Plo, Phi, Mlo, Mhi);
                        }
                    }
                }
                // G = 16: every row of the band has passed column 16 q - 1 by step 16 q - 1 + 15 and none has passed 16 (q + 1) - 1
                if (FILL && LG == 4 && (s & 15) == 14 && s >= 30) {
                    const int q = (s - 14) >> 4;                   // 1 .. 3
                    if (inband && 16 * q - 1 < ncols) cpk[(int64_t)q * cps] = make_uint4(cPlo, cPhi, cMlo, cMhi);
                }
            }
            };
            {   // the block's two halves, each in the cheapest form it qualifies for
                const int s0 = 32 * blk;
                if (steady_ok && s0 >= Hm - 1 && s0 + 16 <= 64) { if (inband) steps16(std::true_type{}, std::true_type{}, std::integral_constant<int, 0>{}); }
                else if (s0 + 16 <= nsteps) steps16(std::true_type{}, std::false_type{}, std::integral_constant<int, 0>{});
                else steps16(std::false_type{}, std::false_type{}, std::integral_constant<int, 0>{});
                const int s1 = s0 + 16;
                if (steady_ok && s1 >= Hm - 1 && s1 + 16 <= 64) { if (inband) steps16(std::true_type{}, std::true_type{}, std::integral_constant<int, 1>{}); }
                else if (s1 + 16 <= nsteps) steps16(std::true_type{}, std::false_type{}, std::integral_constant<int, 1>{});
                else if (s1 < nsteps) steps16(std::false_type{}, std::false_type{}, std::integral_constant<int, 1>{});
            }
        }
        if (FILL && inband && ncols == 64 && si > 0) cpk[(int64_t)QE_CPC * cps - 64] = make_uint4(Plo, Phi, Mlo, Mhi);
        if (inband) {
            if (FILL) {
                // carry-in words of (chunk, slot), bit c = column c
                const int sh = 64 - ncols;
                const u64 xP = __builtin_bitreverse64(sh ? (gP << sh) : gP), xM = __builtin_bitreverse64(sh ? (gM << sh) : gM);
                hw[((int64_t)k * gns + si) * 64] = make_uint4(lo32(xP), hi32(xP), lo32(xM), hi32(xM));
            }
            adv += (u32)ncols;
        }
        // scores[] of the band's rows: carry-ins of the top row (ncols ones) plus the vertical-delta changes down to the row
        {
            const int Va = __popcll(mk64(Plo, Phi) & vm) - __popcll(mk64(Mlo, Mhi) & vm);
            int x = inband ? Va - Vb : 0;
            int y;
            if (LG == 4) {
                y = row_ror_n<1>(x); if (i >= 1) x += y;
                y = row_ror_n<2>(x); if (i >= 2) x += y;
                y = row_ror_n<4>(x); if (i >= 4) x += y;
                y = row_ror_n<8>(x); if (i >= 8) x += y;
            } else {
                // band order is lane order rotated by the top row's lane: the lane-order scan, less what precedes the top row,
                // plus the wave's total for the lanes the band wraps around to
                const int rf = __builtin_amdgcn_readfirstlane(r_first & GM);
                const int S = wave_scan_add(x);
                const int tot = __builtin_amdgcn_readlane(S, 63), before = __builtin_amdgcn_readlane(S, (rf + GM) & GM);
                x = S - (rf ? before : 0) + ((j < rf) ? tot : 0);
                (void)y;
            }
            if (inband) sc += ncols + x;
        }
        {
            // every-64-columns bookkeeping (bpm_banded.c:264-301 / 889-922), the same for all lanes of a group
            int s_top1, s_bot, s_bot1;
            if (LG == 6) {                                          // one task per wave: the rows' lanes are the same in all lanes
                s_top1 = __builtin_amdgcn_readlane(sc, __builtin_amdgcn_readfirstlane((r_first + 1) & GM));
                s_bot = __builtin_amdgcn_readlane(sc, __builtin_amdgcn_readfirstlane((last + pos_v) & GM));
                s_bot1 = __builtin_amdgcn_readlane(sc, __builtin_amdgcn_readfirstlane((last + pos_v - 1) & GM));
            } else {
                s_top1 = __shfl(sc, gl | ((r_first + 1) & GM));
                s_bot = __shfl(sc, gl | ((last + pos_v) & GM));
                s_bot1 = __shfl(sc, gl | ((last + pos_v - 1) & GM));
            }
            if (on && ncols == 64) {
                const bool c1 = (first + 2 < last) && (G.fin > 64 * (first + 1));
                const bool cut_lo = c1 && (s_top1 + (G.fin - 64 * (first + 1)) > G.cutoff);
                if (cut_lo && pos_h >= G.prolog) first++;
                else if (!cut_lo && pos_h < G.prolog) first--;
                const int pos = last + pos_v;
                if (((pos + 1) & GM) == j) {                       // the new bottom row is this lane's
                    Plo = ~0u; Phi = ~0u; Mlo = 0; Mhi = 0;
                    sc = s_bot + 64;
                    pa = qa; pb = qb;
                    if (FILL) cp[(int64_t)(QE_CPC * k + QE_CPC) * cps + (int64_t)last * 64] = make_uint4(~0u, ~0u, 0u, 0u);
                }
                max_row_init = max(max_row_init, pos + 1);
                const bool c2 = (first + 2 < last) && (64 * (last - 1) > G.fin);
                const bool cut_hi = c2 && (s_bot1 + (64 * (last - 1) - G.fin) > G.cutoff);
                if (cut_hi || (pos_v + last >= stop_row)) last--;
                pos_v++;
                pos_h++;
                if (FILL && j == 0) { cf[(int64_t)pos_h * 64] = (int16_t)first; cl[(int64_t)pos_h * 64] = (int16_t)last; }
            }
        }
    }
    // final score read-out (bpm_banded.c:952-961)
    const int s_last = __shfl(sc, gl | ((nw - 1) & GM));
    u32 asum = adv;
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) asum += __shfl_xor(asum, o);
    if (ok) {
        int score = -1;
        if (nw - 1 <= max_row_init) {
            score = s_last;
            if (m & 63) score -= 64 - (m & 63);
        }
        bool again = false;
        if (!FILL && A.doubling) {
            const int mx = max(m, n);
            again = (score > mx / 4 && (int64_t)cut_cur * 3 / 2 < (int64_t)score) || score < 0;
        }
        if (again) cut_cur = (int)min((int64_t)max(cut_cur, 0) * 2 + (cut_cur <= 0 ? 1 : 0), (int64_t)1 << 30);      // a cutoff of 0 doubles to 1 (oracle header)
        else {
            want = false;
            if (j == 0) {
                A.o_abort[t] = 0;
                A.o_score[t] = score;
                A.o_first[t] = first;
                A.o_last[t] = last;
                A.o_posv[t] = pos_v;
                A.o_maxrow[t] = max_row_init;
                A.o_adv[t] = asum;
                if (A.o_cutoff) A.o_cutoff[t] = cut_cur;
            }
        }
    }
    if (FILL || !A.doubling) break;
    }
}
template __global__ void k_banded_sys<4, true>(BandedArgs);
template __global__ void k_banded_sys<6, true>(BandedArgs);
template __global__ void k_banded_sys<4, false>(BandedArgs);
template __global__ void k_banded_sys<6, false>(BandedArgs);

// ===========================================================================
// k_banded_sys<6, ..> for bands of 64 .. 127 slots (k_banded_sys2): ONE WAVE PER TASK, TWO ROWS PER LANE.  Lane j holds the
// band's rows r = j (mod 64) in two layers (layer = bit 6 of r: two rows of one lane are 64 apart, and a band of < 128
// slots holds at most one of each); a chunk is walked in two SWEEPS of up to 64 rows -- band rows 0 .. 63, then 64 .. H - 1 --
// each the systolic array of k_banded_sys.  Between the sweeps travels what travels between passes of k_banded: the 64
// carry-out bits of row 63 as a carry word (collected by its lane, handed to the lane of row 64) and the sum of its deltas
// for the scores[] chain.  The top row of a sweep takes its carries from a per-step select (the boundary (1, 0), or the
// carry word's bit) instead of an idle lane.  Same cells, same bookkeeping (bpm_banded.c:264-301 / 889-922), same outputs
// and layout as k_banded_sys; the bounds of pairs with several large indels (QuickEd's leaves and stage-3 passes there)
// and wide user bandwidths.
// ===========================================================================
template <bool FILL>
__global__ __launch_bounds__(256) void k_banded_sys2(BandedArgs A) {
    if (A.prio) __builtin_amdgcn_s_setprio(3);          // few waves, each a serial chain: first in line at the SIMD's issue arbiter
    constexpr int GL = 64, GM = 63;
    const int t = QE_GROUP_INDEX(), j = threadIdx.x & 63;
    if (t >= A.T.ntasks) return;
    int pair = A.T.pair[t];
    if (A.only_if != nullptr && pair >= 0 && A.only_if[t] == 0) pair = -1;
    if (pair < 0) return;                                          // (wave-uniform: one task per wave)
    const int m = A.T.m[t], n = A.T.n[t], p0 = A.T.p0[t], t0 = A.T.t0[t], cut_in = A.T.cutoff[t];
    const int tfin = FILL ? n : A.T.tfin[t];
    const u64* pp = A.P.pl_p + A.P.pl_p_off[pair];
    const u64* tp = A.P.pl_t + A.P.pl_t_off[pair];
    const u32 fl = A.P.flags[pair];
    const Geom G = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    const bool fgeom = FILL || A.fill_geom != 0;
    const int nsl = fgeom ? G.ebb : ((G.cutoff + 63) >> 6) + 1;
    const bool ok = (fl & FLAG_HAS_N) == 0 && nsl <= 2 * GL - 1 && tfin == n;
    if (j == 0) A.o_abort[t] = ok ? 0 : 1;
    if (!ok) return;
    const int stop_row = fgeom ? nw - 1 : nw;
    const u64 lvl_mask = (m & 63) ? (((u64)1 << (m & 63)) - 1) : QE_ONES;
    int first = G.prolog, last = nsl - 1, pos_v = -G.prolog, pos_h = 0;
    int max_row_init = nsl - 1;
    u32 adv = 0;
    const int g = t >> 6, col = t & 63;
    int gns = 0, gnch = 0;
    int16_t* cf = nullptr; int16_t* cl = nullptr;
    uint4* cp = nullptr; uint4* hw = nullptr;
    int64_t cps = 0;
    if (FILL) {
        gns = A.g_nslots[g]; gnch = A.g_nch[g];
        const GroupWs W = group_ws(A.ws, A.g_ws_off[g], gns, A.g_nrows[g], gnch);
        cf = W.cf + col; cl = W.cl + col;
        cp = A.mat + A.g_mat_off[g] + col;
        cps = (int64_t)gns * 64;
        hw = cp + (int64_t)QE_CPC * gnch * cps;
    }
    // bpm_reset_search: lane j holds row j (layer 0) and row j + 64 (layer 1)
    u32 L0Plo = ~0u, L0Phi = ~0u, L0Mlo = 0, L0Mhi = 0, L1Plo = ~0u, L1Phi = ~0u, L1Mlo = 0, L1Mhi = 0;
    int L0sc = 64 * (j + 1), L1sc = 64 * (j + 65);
    u64 L0pa = 0, L0pb = 0, L1pa = 0, L1pb = 0;
    if (FILL) {
        if (j < nsl) cp[(int64_t)j * 64] = make_uint4(~0u, ~0u, 0u, 0u);
        if (j + 64 < nsl) cp[(int64_t)(j + 64) * 64] = make_uint4(~0u, ~0u, 0u, 0u);
        if (j == 0) { cf[0] = (int16_t)first; cl[0] = (int16_t)last; }
    }
    if (j < nw) load_planes_ab(pp, p0 + 64 * j, L0pa, L0pb);
    if (j + 64 < nw) load_planes_ab(pp, p0 + 64 * (j + 64), L1pa, L1pb);
    const int nfull = n >> 6, tail = n & 63;
    const int nchunks = nfull + (tail ? 1 : 0);
    u64 nT0 = 0, nT1 = 0;
    load_planes_ab(tp, t0, nT0, nT1);

    for (int k = 0; k < nchunks; ++k) {
        const int ncols = (k < nfull) ? 64 : tail;
        const u64 T0 = nT0, T1 = nT1;
        if (k + 1 < nchunks) load_planes_ab(tp, t0 + 64 * (k + 1), nT0, nT1);
        const int r_first = first + pos_v;
        const int rhi = min(last, nw - 1 - pos_v);
        const int H = rhi - first + 1;                             // rows computed in this chunk (wave-uniform)
        u64 qa = 0, qb = 0;
        const int new_row = last + pos_v + 1;
        if (ncols == 64 && ((new_row & GM) == j) && new_row < nw) load_planes_ab(pp, p0 + 64 * new_row, qa, qb);
        const int nsweeps = H > GL ? 2 : 1;
        u64 cwP = 0, cwM = 0;                                      // carry words into sweep 1's top row, bit c = column c
        int base_delta = ncols;                                    // sum of the carry-ins of the sweep's top row
#pragma unroll 1
        for (int q = 0; q < nsweeps; ++q) {
            const int i_in = (j - (r_first + q * GL)) & GM;        // this lane's row of the sweep, counted from its top
            const int i = q * GL + i_in, si = first + i, my_row = r_first + i;
            const bool inband = si <= rhi;
            const bool lay = ((my_row >> 6) & 1) != 0;
            u32 Plo = lay ? L1Plo : L0Plo, Phi = lay ? L1Phi : L0Phi, Mlo = lay ? L1Mlo : L0Mlo, Mhi = lay ? L1Mhi : L0Mhi;
            int sc = lay ? L1sc : L0sc;
            const u64 pa = lay ? L1pa : L0pa, pb = lay ? L1pb : L0pb;
            const u64 R0 = i_in ? ((T0 << i_in) | (T0 >> (64 - i_in))) : T0, R1 = i_in ? ((T1 << i_in) | (T1 >> (64 - i_in))) : T1;
            const u64 vm = (my_row == nw - 1) ? lvl_mask : QE_ONES;
            const int Vb = __popcll(mk64(Plo, Phi) & vm) - __popcll(mk64(Mlo, Mhi) & vm);
            const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
            u64 gP = 0, gM = 0, gOP = 0, gOM = 0;                  // this row's carry-ins / carry-outs, shifted in from the right
            u32 oP = 0, oM = 0;
            const bool top = i_in == 0;
            const u32 len = inband ? (u32)ncols : 0u;
            const int Hq = min(H - q * GL, GL);
            const int nsteps = 64 + Hq - 1;
            const bool more = q + 1 < nsweeps;
            uint4* const cpk = FILL ? cp + (int64_t)(QE_CPC * k) * cps + (int64_t)si * 64 : nullptr;
#pragma unroll 1
            for (int blk = 0; blk < 4; ++blk) {
                if (32 * blk >= nsteps) break;
                const u32 w0 = (blk & 1) ? hi32(R0) : lo32(R0), w1 = (blk & 1) ? hi32(R1) : lo32(R1);
                // the top row's carries: the band's boundary (1, 0) in sweep 0, row 63's carry words in sweep 1 (column = step)
                const u32 twP = q ? ((blk == 0) ? lo32(cwP) : ((blk == 1) ? hi32(cwP) : 0u)) : ~0u;
                const u32 twM = q ? ((blk == 0) ? lo32(cwM) : ((blk == 1) ? hi32(cwM) : 0u)) : 0u;
#pragma unroll
                for (int sb = 0; sb < 32; ++sb) {
                    const int s = 32 * blk + sb;
                    if (s >= nsteps) continue;
                    u32 inP = grp_ror1<6>(oP), inM = grp_ror1<6>(oM);
                    inP = top ? __builtin_amdgcn_ubfe(twP, sb, 1) : inP;
                    inM = top ? __builtin_amdgcn_ubfe(twM, sb, 1) : inM;
                    const u32 c = (u32)(s - i_in);
                    if (c < len) {
                        const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)w0, sb, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)w1, sb, 1);
                        const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                        u32 phhi, mhhi;
                        block_step_core(elo, ehi, Plo, Phi, Mlo, Mhi, inP, inM, phhi, mhhi);
                        oP = phhi >> 31; oM = mhhi >> 31;
                        if (more) { gOP = shl1_add_u64(gOP, (u64)oP); gOM = shl1_add_u64(gOM, (u64)oM); }
                        if (FILL) {
                            gP = shl1_add_u64(gP, (u64)inP); gM = shl1_add_u64(gM, (u64)inM);
                            if ((c & (QE_CP_COLS - 1)) == QE_CP_COLS - 1 && c != 63)
                                cpk[(int64_t)((c >> 4) + 1) * cps] = make_uint4(Plo, Phi, Mlo, Mhi);
                        }
                    }
                }
            }
            if (FILL && inband && ncols == 64 && si > 0) cpk[(int64_t)QE_CPC * cps - 64] = make_uint4(Plo, Phi, Mlo, Mhi);
            const int sh = 64 - ncols;
            if (inband) {
                if (FILL) {
                    const u64 xP = __builtin_bitreverse64(sh ? (gP << sh) : gP), xM = __builtin_bitreverse64(sh ? (gM << sh) : gM);
                    hw[((int64_t)k * gns + si) * 64] = make_uint4(lo32(xP), hi32(xP), lo32(xM), hi32(xM));
                }
                adv += (u32)ncols;
            }
            // scores[]: the sweep's rows from the sum of its top row's carry-ins down
            const int Va = __popcll(mk64(Plo, Phi) & vm) - __popcll(mk64(Mlo, Mhi) & vm);
            int x = inband ? Va - Vb : 0;
#pragma unroll
            for (int d = 1; d < GL; d <<= 1) { const int y = __shfl(x, (j - d) & GM); if (i_in >= d) x += y; }
            const int dl = base_delta + x;                         // this row's sum of exported deltas
            if (inband) sc += dl;
            if (more) {
                // to sweep 1: row 63's carry-outs and the sum of its deltas
                const int src = (r_first + GL - 1) & GM;
                const u64 oPn = __builtin_bitreverse64(sh ? (gOP << sh) : gOP), oMn = __builtin_bitreverse64(sh ? (gOM << sh) : gOM);
                cwP = mk64((u32)__shfl((int)lo32(oPn), src), (u32)__shfl((int)hi32(oPn), src));
                cwM = mk64((u32)__shfl((int)lo32(oMn), src), (u32)__shfl((int)hi32(oMn), src));
                base_delta = __shfl(dl, src);
            }
            if (lay) { L1Plo = Plo; L1Phi = Phi; L1Mlo = Mlo; L1Mhi = Mhi; L1sc = sc; }
            else { L0Plo = Plo; L0Phi = Phi; L0Mlo = Mlo; L0Mhi = Mhi; L0sc = sc; }
        }
        {
            // every-64-columns bookkeeping (bpm_banded.c:264-301 / 889-922)
            auto row_sc = [&](int row) {
                const int a0 = __shfl(L0sc, row & GM), a1 = __shfl(L1sc, row & GM);
                return ((row >> 6) & 1) ? a1 : a0;
            };
            const int s_top1 = row_sc(r_first + 1), s_bot = row_sc(last + pos_v), s_bot1 = row_sc(last + pos_v - 1);
            if (ncols == 64) {
                const bool c1 = (first + 2 < last) && (G.fin > 64 * (first + 1));
                const bool cut_lo = c1 && (s_top1 + (G.fin - 64 * (first + 1)) > G.cutoff);
                if (cut_lo && pos_h >= G.prolog) first++;
                else if (!cut_lo && pos_h < G.prolog) first--;
                const int pos = last + pos_v;
                if (((pos + 1) & GM) == j) {                       // the new bottom row is this lane's, in the layer of its bit 6
                    if (((pos + 1) >> 6) & 1) { L1Plo = ~0u; L1Phi = ~0u; L1Mlo = 0; L1Mhi = 0; L1sc = s_bot + 64; L1pa = qa; L1pb = qb; }
                    else { L0Plo = ~0u; L0Phi = ~0u; L0Mlo = 0; L0Mhi = 0; L0sc = s_bot + 64; L0pa = qa; L0pb = qb; }
                    if (FILL) cp[(int64_t)(QE_CPC * k + QE_CPC) * cps + (int64_t)last * 64] = make_uint4(~0u, ~0u, 0u, 0u);
                }
                max_row_init = max(max_row_init, pos + 1);
                const bool c2 = (first + 2 < last) && (64 * (last - 1) > G.fin);
                const bool cut_hi = c2 && (s_bot1 + (64 * (last - 1) - G.fin) > G.cutoff);
                if (cut_hi || (pos_v + last >= stop_row)) last--;
                pos_v++;
                pos_h++;
                if (FILL && j == 0) { cf[(int64_t)pos_h * 64] = (int16_t)first; cl[(int64_t)pos_h * 64] = (int16_t)last; }
            }
        }
    }
    const int row = nw - 1;
    const int a0 = __shfl(L0sc, row & GM), a1 = __shfl(L1sc, row & GM);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) adv += __shfl_xor(adv, o);
    if (j == 0) {
        int score = -1;
        if (row <= max_row_init) {
            score = ((row >> 6) & 1) ? a1 : a0;
            if (m & 63) score -= 64 - (m & 63);
        }
        A.o_score[t] = score;
        A.o_first[t] = first;
        A.o_last[t] = last;
        A.o_posv[t] = pos_v;
        A.o_maxrow[t] = max_row_init;
        A.o_adv[t] = adv;
    }
}
template __global__ void k_banded_sys2<true>(BandedArgs);
template __global__ void k_banded_sys2<false>(BandedArgs);

// ---------------------------------------------------------------------------
// RLE emitter shared by the tracebacks: ops arrive back to front
// ---------------------------------------------------------------------------
#ifndef QE_TB_ELIDE
#define QE_TB_ELIDE 0      // tools/pmc_tb_elide.sh: k_traceback with one kind of memory access left out (results garbage, time is the datum)
#endif
struct RunSink {
    u32* runs; int cap; int nruns; int cur_op; int cur_len; int nops; int edits; int stride;
    // stride: elements between consecutive runs of this lane's task: 64 in the [idx][lane] layout (one coalesced row per store
    // of a wave, what the one-lane-per-alignment formatter reads), 1 when every task has its runs to itself (run_base)
    __device__ __forceinline__ void init(u32* r, int c, int stride_ = 64) { runs = r; cap = c; nruns = 0; cur_op = -1; cur_len = 0; nops = 0; edits = 0; stride = stride_; }
    __device__ __forceinline__ void push(int op) {
        if (op == cur_op) { ++cur_len; }
        else {
            if (cur_len > 0 && nruns < cap) runs[(int64_t)nruns * stride] = ((u32)cur_len << 2) | (u32)cur_op;
            if (cur_len > 0) ++nruns;
            cur_op = op; cur_len = 1;
        }
        ++nops;
        edits += (op != (int)OP_M);
    }
    __device__ __forceinline__ void push_n(int op, int count) {
        if (count <= 0) return;
        if (op == cur_op) { cur_len += count; }
        else {
            if (cur_len > 0 && nruns < cap) runs[(int64_t)nruns * stride] = ((u32)cur_len << 2) | (u32)cur_op;
            if (cur_len > 0) ++nruns;
            cur_op = op; cur_len = count;
        }
        nops += count;
        edits += (op != (int)OP_M) ? count : 0;
    }
    // predicated form for straight-line walks: nothing happens when !pred; one guarded store, no other branch
    __device__ __forceinline__ void emit(int op, int count, bool pred) {
        const bool brk = pred && op != cur_op;
        const bool st = brk && cur_len > 0;
        if (!(QE_TB_ELIDE & 1)) if (st && nruns < cap) runs[(int64_t)nruns * stride] = ((u32)cur_len << 2) | (u32)cur_op;
        nruns += st ? 1 : 0;
        cur_len = brk ? count : cur_len + (pred ? count : 0);
        cur_op = brk ? op : cur_op;
        nops += pred ? count : 0;
        edits += (pred && op != (int)OP_M) ? count : 0;
    }
    // emit without the op / edit counters (walk_tile_lean derives them from its step and match counts)
    __device__ __forceinline__ void emit_run(int op, int count, bool pred) {
        const bool brk = pred && op != cur_op;
        const bool st = brk && cur_len > 0;
        if (st && nruns < cap) runs[(int64_t)nruns * stride] = ((u32)cur_len << 2) | (u32)cur_op;
        nruns += st ? 1 : 0;
        cur_len = brk ? count : cur_len + (pred ? count : 0);
        cur_op = brk ? op : cur_op;
    }
    __device__ __forceinline__ void flush() {
        if (cur_len > 0) {
            if (nruns < cap) runs[(int64_t)nruns * stride] = ((u32)cur_len << 2) | (u32)cur_op;
            ++nruns;
            cur_len = 0;
        }
    }
};

// raw-byte equality of text[h] and pattern[v] (bpm_banded.c:1012): encoded equality is the
// same thing for upper-case ACGTN input; other input compares bytes.  A path moves mostly along
// a diagonal, so the test is done 64 cells at a time: the text block's planes are rotated onto
// the pattern block's bit positions for the current diagonal (v - h) and compared once; a step
// then reads one bit.  The word is rebuilt when the path changes block or diagonal.
struct EqTest {
    const u64* pp; const u64* tp; const uint8_t* ap; const uint8_t* at; bool raw;
    int rp = -1, rt = -1;     // >= 0: the planes hold the reversed strings, ASCII index = r - i
    int kp = -1, kt = -1, kd = 0;
    u64 eqw = 0;
    __device__ __forceinline__ bool eq(int v, int h) {
        if (raw) return ap[rp >= 0 ? rp - v : v] == at[rt >= 0 ? rt - h : h];
        const int d = (v - h) & 63;
        if (((v >> 6) != kp) | ((h >> 6) != kt) | (d != kd)) {
            kp = v >> 6; kt = h >> 6; kd = d;
            const u64* q = pp + 3 * (int64_t)kp;
            const u64* r = tp + 3 * (int64_t)kt;
            const u64 pa = q[0], pb = q[1], pn = q[2];
            u64 ta = r[0], tb = r[1], tn = r[2];
            if (d) {   // text bit (h & 63) must land on pattern bit (v & 63): rotate left by d
                ta = (ta << d) | (ta >> (64 - d)); tb = (tb << d) | (tb >> (64 - d)); tn = (tn << d) | (tn >> (64 - d));
            }
            eqw = ~((pa ^ ta) | (pb ^ tb) | (pn ^ tn));       // equal codes <=> all three plane bits equal
        }
        return (eqw >> (v & 63)) & 1;
    }
};

// One traceback round over a TW-column tile held in registers: tP/tM/tE[j] = {Pv after, Mv before, Eq}
// of tile column j for the lane's block row Rb.  Priority D -> I -> M/X (bpm_banded.c:994-1020).  A lane
// enters at column h & 7, and leaves to the left (h < 8 q), upwards (v leaves block Rb) or at an edge.
template <bool RAW, int TW>
__device__ __forceinline__ void walk_tile(const u64 (&tP)[TW], const u64 (&tM)[TW], const u64 (&tE)[TW], bool in_tile,
                                          u32 inb_same, u32 inb_7, int Rb, int& v, int& h, u32& steps,
                                          RunSink& R, EqTest& E, int p0, int t0) {
#pragma unroll
    for (int j = TW - 1; j >= 0; --j) {
        const bool mine = in_tile && (h & (TW - 1)) == j;
        const u32 inb = (j == TW - 1) ? inb_7 : inb_same;
        const int bit = v & 63;
        // deletions: Pv bits bit, bit-1, ... while set (v moves up, h stays)
        const u64 x = tP[j] << (63 - bit);
        int r = inb ? min(__clzll((long long)~x), bit + 1) : 0;
        if (!mine) r = 0;
        R.emit((int)OP_D, r, r > 0);
        const bool up = r == bit + 1;                 // the run reached the top of the block: next round, same column
        const bool go = mine && !up;
        const int b1 = (bit - r) & 63;
        const u32 isI = inb & (u32)((tM[j] >> b1) & 1);
        u32 eq = (u32)((tE[j] >> b1) & 1);
        if (RAW) { if (go && E.raw) eq = E.eq(p0 + v - r, t0 + h) ? 1u : 0u; }
        R.emit(isI ? (int)OP_I : (eq ? (int)OP_M : (int)OP_X), 1, go);
        v -= r + ((go && !isI) ? 1 : 0);
        h -= go ? 1 : 0;
        steps += (u32)r + (go ? 1u : 0u);
        in_tile = in_tile && !(mine && up) && v >= 0 && (v >> 6) == Rb;
    }
}

// walk_tile for canonical input with the rare things behind branches the wave skips when no lane takes them
// (s_cbranch_execz): a column costs ~30 instructions instead of ~85 while no lane of the wave starts a deletion run or
// ends a run of equal operations at it (a path is mostly a diagonal of matches).  The same steps and the same runs; the operation / edit counters are not kept per
// column: nops = steps (+ the leftovers), edits = steps - nmatch.
template <int TW>
__device__ __forceinline__ void walk_tile_lean(const u64 (&tP)[TW], const u64 (&tM)[TW], const u64 (&tE)[TW], bool in_tile,
                                               u32 inb_same, u32 inb_7, int Rb, int& v, int& h, u32& steps, int& nmatch, RunSink& R) {
#pragma unroll
    for (int j = TW - 1; j >= 0; --j) {
        const bool mine = in_tile && (h & (TW - 1)) == j;
        const u32 inb = (j == TW - 1) ? inb_7 : inb_same;
        const int bit = v & 63;
        const u32 pb = inb & (u32)(tP[j] >> bit) & 1u;
        int r = 0, b1 = bit;
        if (mine && pb != 0) {                        // deletions: Pv bits bit, bit-1, ... while set (v moves up, h stays)
            const u64 x = tP[j] << (63 - bit);
            r = min(__clzll((long long)~x), bit + 1);
            R.emit_run((int)OP_D, r, true);
            b1 = (bit - r) & 63;
        }
        const bool up = r == bit + 1;                 // the run reached the top of the block: next round, same column
        const bool go = mine && !up;
        const u32 isI = inb & (u32)(tM[j] >> b1) & 1u;
        const u32 eq = (u32)(tE[j] >> b1) & 1u;
        const int op = isI ? (int)OP_I : (eq ? (int)OP_M : (int)OP_X);
        const bool brk = go && op != R.cur_op;
        if (brk) {
            if (R.cur_len > 0) {
                if (R.nruns < R.cap) R.runs[(int64_t)R.nruns * R.stride] = ((u32)R.cur_len << 2) | (u32)R.cur_op;
                ++R.nruns;
            }
            R.cur_len = 0; R.cur_op = op;
        }
        R.cur_len += go ? 1 : 0;
        nmatch += (go && !isI && eq) ? 1 : 0;
        v -= r + ((go && !isI) ? 1 : 0);
        h -= go ? 1 : 0;
        steps += (u32)r + (go ? 1u : 0u);
        in_tile = in_tile && !(mine && up) && v >= 0 && (v >> 6) == Rb;
        // opaque to the optimizer: InstCombine's known-bits walk over the v / h phi chains of TW conditional columns inside
        // a loop is exponential in TW (28 s of a 29 s compile at TW = 16 in isolation, no end in the kernels)
        asm("" : "+v"(v), "+v"(h));
    }
}


// ===========================================================================
// BandEd traceback (bpm_banded.c:967-1036): priority D -> I -> M/X.  One lane per task walks its own
// path.  The fill left a checkpoint {Pv, Mv} every QE_CP_COLS (16) columns and the carry-in words of every
// (chunk, slot); a round of this kernel recomputes, for every lane at once, the 16 columns of the
// (column tile, block row) its path is in -- 16 block steps from the checkpoint, the same arithmetic as
// the fill, so the same bits -- into registers as {Pv after, Mv before, Eq} per column, then lets every lane walk
// while it stays inside that tile.  HBM sees 1.25 B per block-column instead of 16.
// Cells the fill did not compute read as P = 0, M = 0 (see oracle header).
// ===========================================================================
__global__ __launch_bounds__(512) void k_traceback(TraceArgs A) {
    const int g = QE_GROUP_INDEX(), lane = threadIdx.x & 63, t = g * 64 + lane;
    if (g * 64 >= A.T.ntasks) return;
    int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    if (A.only_if != nullptr && pair >= 0 && A.only_if[t] == 0) pair = -1;
    const bool valid = pair >= 0;
    if (!__any(valid)) return;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0;
    const u64* pp = A.P.pl_p; const u64* tp = A.P.pl_t;
    EqTest E; E.pp = pp; E.tp = tp; E.ap = nullptr; E.at = nullptr; E.raw = false;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t]; cut_in = A.T.cutoff[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair]; tp = A.P.pl_t + A.P.pl_t_off[pair];
        E.pp = pp; E.tp = tp;
        E.ap = A.P.asc_p + A.P.asc_p_off[pair]; E.at = A.P.asc_t + A.P.asc_t_off[pair];
        E.raw = (A.P.flags[pair] & FLAG_NONCANON) != 0;
    }
    const bool any_raw = __any(E.raw);
    const Geom G = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    const int gns = A.g_nslots[g], gnch = A.g_nch[g];
    const GroupWs W = group_ws(const_cast<uint8_t*>(A.ws), A.g_ws_off[g], gns, A.g_nrows[g], gnch);
    const int16_t* cf = W.cf + lane;
    const int16_t* cl = W.cl + lane;
    const uint4* cp = A.mat + A.g_mat_off[g] + lane;
    const int64_t cps = (int64_t)gns * 64;
    const uint4* hw = cp + (int64_t)QE_CPC * gnch * cps;
    RunSink R;
    {
        const int cap = valid ? A.g_runs_cap[g] : 0;
        R.init(valid ? A.runs + A.g_runs_off[g] + (A.runs_by_task ? (int64_t)lane * cap : (int64_t)lane) : nullptr, cap, A.runs_by_task ? 1 : 64);
    }
    int h = n - 1, v = m - 1;
    u32 steps = 0;
    int nmatch = 0;
    // what a round needs besides the checkpoint changes rarely: band-edge records and text planes per chunk (every 8
    // tiles), carry words per (chunk, slot), pattern planes per block row -- kept in registers, reloaded on change
    int ck = -1, cs = -1, cR = -1;
    int cf_a = 0, cf_b = 0, cl_b = -1;
    u64 T0 = 0, T1 = 0, TN = 0, hinP = 0, hinM = 0, pa = 0, pb = 0, pn = 0;
    while (__any(valid && v >= 0 && h >= 0)) {
        const bool act = valid && v >= 0 && h >= 0;
        constexpr int TW = QE_CP_COLS;     // tile width = the fill's checkpoint interval
        const int q = h / TW, Rb = v >> 6, k = q / (64 / TW);
        // the tile: per column {Pv after, Mv before, Eq}, in registers (the loops over its columns are unrolled)
        u64 tP[TW], tM[TW], tE[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) { tP[j] = 0; tM[j] = 0; tE[j] = 0; }
        u32 inb_same = 0, inb_7 = 0;
        if (act) {
            if (k != ck) {
                ck = k; cs = -1;
                if (QE_TB_ELIDE & 8) { cf_a = 0; cf_b = 0; cl_b = gns - 1; T0 = 0x9E3779B97F4A7C15ull * (u64)(k + lane + 1); T1 = T0 >> 7; TN = 0; }
                else {
                cf_a = cf[(int64_t)(k + 1) * 64]; cf_b = cf[(int64_t)k * 64]; cl_b = cl[(int64_t)k * 64];
                load_planes(tp, t0 + 64 * k, T0, T1, TN);
                }
            }
            const int s = Rb - (k - G.prolog);
            if (Rb != cR) { cR = Rb; if (QE_TB_ELIDE & 8) { pa = 0xD1B54A32D192ED03ull * (u64)(Rb + lane + 1); pb = pa >> 5; pn = 0; } else load_planes(pp, p0 + 64 * Rb, pa, pb, pn); }
            // a step at column h reads Pv of stored column h + 1: inside the band of THAT column's chunk or 0
            // (oracle header).  All but the last column of the tile share this chunk; the last may be the chunk's last.
            inb_same = (u32)((s >= 0) & (s >= cf_b) & (s <= cl_b));
            inb_7 = ((q & (64 / TW - 1)) == 64 / TW - 1) ? (u32)((s >= 1) & (s - 1 >= cf_a) & (s - 1 <= cl_b)) : inb_same;
        }
        const int c_first = (TW * q) & 63;
        u64 P = 0, M = 0;
        bool computed = false;
        if (act) {
            const int pos_v = k - G.prolog, s = Rb - pos_v;
            computed = s >= cf_b && s <= min(cl_b, nw - 1 - pos_v);
            if (computed) {
                const int se = (QE_TB_ELIDE & 14) ? min(max(s, 0), gns - 1) : s;        // (with made-up band edges any slot may come up)
                const uint4 c0 = (QE_TB_ELIDE & 2) ? make_uint4((u32)q * 2654435761u, (u32)lane, 0u, 0u) : cp[(int64_t)(q * (TW / QE_CP_COLS)) * cps + (int64_t)se * 64];
                if (s != cs) {
                    cs = s;
                    const uint4 w0 = (QE_TB_ELIDE & 4) ? make_uint4(~0u, ~0u, 0u, 0u) : hw[((int64_t)k * gns + se) * 64];
                    hinP = mk64(w0.x, w0.y); hinM = mk64(w0.z, w0.w);
                }
                P = mk64(c0.x, c0.y); M = mk64(c0.z, c0.w);
            }
        }
        // TW block steps from the checkpoint: the same arithmetic as the fill.  The step's Eq word is also
        // the traceback's match test for the 64 cells of the column (bpm_banded.c:1012: equal codes; raw
        // bytes only for non-canonical input, see EqTest)
        if (!__any(act && ((TN | pn) != 0))) {
            const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
            const u32 t0s = (u32)(T0 >> c_first), t1s = (u32)(T1 >> c_first);
            const u32 hp = (u32)(hinP >> c_first), hm = (u32)(hinM >> c_first);
            u32 Plo = lo32(P), Phi = hi32(P), Mlo = lo32(M), Mhi = hi32(M), gP = 0, gM = 0;
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, j, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, j, 1);
                const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                tE[j] = mk64(elo, ehi);
                tM[j] = mk64(Mlo, Mhi);
                block_step_fused(elo, ehi, Plo, Phi, Mlo, Mhi, __builtin_amdgcn_ubfe(hp, j, 1), __builtin_amdgcn_ubfe(hm, j, 1), gP, gM);
                tP[j] = mk64(Plo, Phi);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int c = c_first + j;
                const u64 m0 = (u64)0 - ((T0 >> c) & 1), m1 = (u64)0 - ((T1 >> c) & 1);
                const u64 acgt = ~(pa ^ m0) & ~(pb ^ m1) & ~pn;
                const u64 Eq = ((TN >> c) & 1) ? pn : acgt;
                tE[j] = Eq;
                tM[j] = M;
                u64 Ph, Mh;
                block_step(Eq, P, M, (u32)((hinP >> c) & 1), (u32)((hinM >> c) & 1), Ph, Mh);
                tP[j] = P;
            }
        }
        if (!computed) {
#pragma unroll
            for (int j = 0; j < TW; ++j) { tP[j] = 0; tM[j] = 0; }
            // the row the band bookkeeping creates at the end of a chunk: Pv = ~0, Mv = 0 (bpm_banded.c:910)
            if (act && (Rb - (k - G.prolog)) == cl_b + 1 && (q & (64 / TW - 1)) == 64 / TW - 1) tP[TW - 1] = QE_ONES;
        }
        // walk, column by column, straight-line: at its column a lane takes the whole run of deletions
        // (consecutive set Pv bits below its row) and then the one step that leaves the column
        if (any_raw) walk_tile<true, TW>(tP, tM, tE, act, inb_same, inb_7, Rb, v, h, steps, R, E, p0, t0);
        else walk_tile_lean<TW>(tP, tM, tE, act, inb_same, inb_7, Rb, v, h, steps, nmatch, R);
    }
    if (!valid) return;
    if (!any_raw) { R.nops = (int)steps; R.edits = (int)steps - nmatch; }      // what walk_tile_lean does not count per column
    R.push_n(OP_I, h + 1);
    R.push_n(OP_D, v + 1);
    R.flush();
    // the run buffer is sized from the cutoff (an alignment inside the parity domain has <= 2 cutoff + 1 runs); a path
    // with more runs (cutoff below the distance: the reference walks uninitialised memory there) is reported, not stored
    A.o_nruns[t] = (R.nruns <= R.cap) ? R.nruns : -1;
    A.o_nops[t] = R.nops;
    A.o_edits[t] = R.edits;
    A.o_steps[t] = steps;
}

// ---------------------------------------------------------------------------
// The walk of k_traceback_sys<3 / 4> (walk_round_diag; described for 16 lanes and a round of 128 columns -- with 8 lanes a
// round is 64 columns and the reductions run over half a DPP row): a path is mostly plain matches along a diagonal, so the walker does not
// visit them column by column, and it is not handed from lane to lane either: its state (v, h, the run under construction,
// the counters) is kept THE SAME IN ALL 16 LANES of its group, every lane runs the same code, and what a lane contributes
// is what its own tile says.  When a round's tiles are rebuilt, every lane lays three bit planes of its tile out ALONG THE
// DIAGONALS near the one the walker is on at the round's start (d0 = v - h): for o = 0 .. 15 a 16-bit word whose bit u
// (u = 15 - column: the walk's direction) is the cell of that column on diagonal d0 + o - 7 --
//   X: the cell is anything but a plain match (Pv | Mv inside the band, or Eq clear); cells outside the block row count as X
//   P: Pv inside the band (a deletion, bpm_banded.c:994-1003);   M: Mv inside the band (an insertion, 1004-1010)
// (diag_words: three windows per column, two 16 x 16 bit transposes).  A step of the walk: every lane looks up its X word
// for the walker's diagonal, finds the first X cell at or after the walker's column, and a min-reduction over the group
// (four row_ror DPP steps) gives the first such cell of the whole round's 128 columns together with its class bits; the run
// of matches up to it is taken in one go, then the cell's own step: a mismatch stays on the diagonal; a deletion takes the
// whole vertical run (the owner lane's Pv word of that column, count-leading-ones); an insertion one step, or, when the
// cell to its left is an insertion too, the whole horizontal run (every lane gathers its tile's row).  The round ends
// when the walker leaves the 128 columns or the predicted tiles, or has drifted more than 7 / 8 diagonals from d0.
// Same steps, runs and bytes as k_traceback (test_traceback_systolic_forced; priority D -> I -> M / X as there).
// ---------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void transpose16x2_stage(u32 (&r)[16]) {  // both 16-bit halves of every word at once
    constexpr u32 m16 = (S == 8) ? 0x00FFu : (S == 4) ? 0x0F0Fu : (S == 2) ? 0x3333u : 0x5555u, mask = m16 | (m16 << 16);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int i = ((n & ~(S - 1)) << 1) | (n & (S - 1));               // the rows with bit S clear
        const u32 a = r[i], b = r[i + S];
        const u32 t = ((a >> S) ^ b) & mask;          // (what a shift drags across the halves' border lands on masked-out bits)
        r[i] = a ^ (t << S); r[i + S] = b ^ t;
    }
}
__device__ __forceinline__ void transpose16x2(u32 (&r)[16]) {        // r[i] bit j <-> r[j] bit i, in the low and in the high halves
    transpose16x2_stage<8>(r); transpose16x2_stage<4>(r); transpose16x2_stage<2>(r); transpose16x2_stage<1>(r);
}

// 16 bits of `w` from bit sc on (sc may be negative or beyond the word); what comes from outside the word is 0.  Both shifts
// are always made -- one of them by 0 -- so that a column's three windows cost no branch: sr = clamp(sc, 0, 63),
// sl = clamp(-sc, 0, 31), live = sc <= 63
__device__ __forceinline__ u32 window16(u64 w, int sr, int sl, bool live) {
    const u32 t = (u32)(w >> sr) << sl;
    return (live ? t : 0u) & 0xFFFFu;
}

template <int TW>
__device__ __forceinline__ void diag_words(const u64 (&tP)[TW], const u64 (&tM)[TW], const u64 (&tE)[TW], u32 inb_same, u32 inb_7, int base,
                                           u32 (&X8)[8], u32 (&P8)[8], u32 (&M8)[8]) {
    static_assert(TW == 16, "sixteen columns per tile");
    u32 r1[16], r2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int sc = base + c - 7;                               // the window's first bit in the column's words
        const int sr = min(max(sc, 0), 63), sl = min(max(-sc, 0), 31);
        const bool live = sc <= 63;
        const bool inb = ((c == TW - 1) ? inb_7 : inb_same) != 0;
        const u64 Pc = inb ? tP[c] : (u64)0, Mc = inb ? tM[c] : (u64)0;
        const u32 wx = ~window16(~((Pc | Mc) | ~tE[c]), sr, sl, live) & 0xFFFFu;      // outside the word: 1
        r1[15 - c] = wx | (window16(Pc, sr, sl, live) << 16);
        r2[15 - c] = window16(Mc, sr, sl, live);
    }
    transpose16x2(r1); transpose16x2(r2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        X8[i] = (r1[2 * i] & 0xFFFFu) | (r1[2 * i + 1] << 16);
        P8[i] = (r1[2 * i] >> 16) | (r1[2 * i + 1] & 0xFFFF0000u);
        M8[i] = (r2[2 * i] & 0xFFFFu) | (r2[2 * i + 1] << 16);
    }
}

__device__ __forceinline__ u32 sel16(const u32 (&D8)[8], u32 o) {    // word o of the sixteen 16-bit words in eight registers
    const u32 i = o >> 1;
    const u32 a0 = (i & 1u) ? D8[1] : D8[0], a1 = (i & 1u) ? D8[3] : D8[2], a2 = (i & 1u) ? D8[5] : D8[4], a3 = (i & 1u) ? D8[7] : D8[6];
    const u32 b0 = (i & 2u) ? a1 : a0, b1 = (i & 2u) ? a3 : a2;
    const u32 w = (i & 4u) ? b1 : b0;
    return (o & 1u) ? (w >> 16) : (w & 0xFFFFu);
}
template <int TW>
__device__ __forceinline__ u64 sel_col(const u64 (&t)[TW], u32 c) {  // t[c], c a lane's own value
    // (a tree of selects on values, level by level: written over one array the optimizer turns it into an indexed load from scratch)
    const bool c8 = (c & 8u) != 0, c4 = (c & 4u) != 0, c2 = (c & 2u) != 0, c1 = (c & 1u) != 0;
    const u64 a0 = c8 ? t[8] : t[0], a1 = c8 ? t[9] : t[1], a2 = c8 ? t[10] : t[2], a3 = c8 ? t[11] : t[3];
    const u64 a4 = c8 ? t[12] : t[4], a5 = c8 ? t[13] : t[5], a6 = c8 ? t[14] : t[6], a7 = c8 ? t[15] : t[7];
    const u64 b0 = c4 ? a4 : a0, b1 = c4 ? a5 : a1, b2 = c4 ? a6 : a2, b3 = c4 ? a7 : a3;
    const u64 d0 = c2 ? b2 : b0, d1 = c2 ? b3 : b1;
    return c1 ? d1 : d0;
}
__device__ __forceinline__ u32 pair_swap(u32 x) { return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, false); }     // quad_perm:[1,0,3,2]
template <int LG>
__device__ __forceinline__ u32 grp_min(u32 x) {                      // the minimum over the 16 / 8 lanes of a group, in all of them
    if (LG == 4) {
        x = min(x, (u32)row_ror_n<1>((int)x)); x = min(x, (u32)row_ror_n<2>((int)x));
        x = min(x, (u32)row_ror_n<4>((int)x)); return min(x, (u32)row_ror_n<8>((int)x));
    }
    // eight lanes = half a DPP row: the quad's neighbour, the quad's other pair, the half row's mirror image
    x = min(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
    x = min(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
    return min(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
}
template <int LG>
__device__ __forceinline__ u32 grp_max(u32 x) {
    if (LG == 4) {
        x = max(x, (u32)row_ror_n<1>((int)x)); x = max(x, (u32)row_ror_n<2>((int)x));
        x = max(x, (u32)row_ror_n<4>((int)x)); return max(x, (u32)row_ror_n<8>((int)x));
    }
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
    x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
    return max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
}

// one round's walk.  act / q / Rb / base: this lane's tile (base = d0 + 16 q - 64 Rb); x = the tile's place in the round
// (0 = the walker's column tile at the round's start, 7 = the leftmost); col_r = the round's rightmost column (u = 0)
template <int TW, int LG>
__device__ __forceinline__ void walk_round_diag(const u64 (&tP)[TW], const u64 (&tM)[TW], const u32 (&X8)[8], const u32 (&P8)[8], const u32 (&M8)[8],
                                                bool live, bool act, int x, int Rb, int base, int d0, int col_r, u32 inb_same, u32 inb_7, bool writer,
                                                int& v, int& h, u32& steps, int& nmatch, RunSink& R) {
    const int u_base = 16 * x;
    bool going = live;
    auto switch_op = [&](int op) {                    // the run under construction ends where the operation changes (RunSink::emit_run)
        if (R.cur_op != op) {
            if (R.cur_len > 0) {
                if (writer && R.nruns < R.cap) R.runs[(int64_t)R.nruns * R.stride] = ((u32)R.cur_len << 2) | (u32)R.cur_op;
                ++R.nruns;
            }
            R.cur_len = 0; R.cur_op = op;
        }
    };
    while (__any(going)) {
        const u32 o = (u32)((v - h) - d0 + 7);
        const bool step = going && o <= 15u;          // (else: drifted off the diagonals the words cover -- the next round re-centres)
        const u32 oc = o & 15u;
        const int u0 = col_r - h;                     // the walker's place in the round's columns, 0 .. SPAN - 1
        // the first cell at or after the walker's that is not a plain match, with its class bits
        const int lo = u0 - u_base;
        const u32 keep = (lo <= 0) ? 0xFFFFu : ((lo > 15) ? 0u : ((0xFFFFu << lo) & 0xFFFFu));
        u32 wx = sel16(X8, oc);
        wx &= pair_swap(wx);                          // the pair's two block rows: the cell is in one of them, the other says 1
        const u32 wm = wx & keep;
        const u32 cu = wm ? (u32)__builtin_ctz(wm) : 0u;
        constexpr int SPAN = 8 << LG;                 // the round's columns: 16 per column tile, half as many tiles as lanes
        const u32 pcell = wm ? (u32)u_base + cu : (u32)SPAN;
        const int bitpos = base + (15 - (int)cu) + (int)o - 7;
        const bool inr = act && wm != 0 && (u32)bitpos < 64u;       // the cell is in THIS lane's block row
        const u32 pb = inr ? ((sel16(P8, oc) >> cu) & 1u) : 0u, mb = inr ? ((sel16(M8, oc) >> cu) & 1u) : 0u;
        u32 pay = (pcell << 3) | (inr ? 4u : 0u) | (pb << 1) | mb;
        pay |= pair_swap(pay) & 7u;
        pay = grp_min<LG>(pay);
        const int ub = (int)(pay >> 3);
        const u32 cls = pay & 7u;
        const int r = step ? ub - u0 : 0;             // plain matches up the diagonal
        if (r > 0) { switch_op((int)OP_M); R.cur_len += r; nmatch += r; steps += (u32)r; v -= r; h -= r; }
        const bool cont = step && v >= 0 && h >= 0 && ub < SPAN && (cls & 4u) != 0;       // (no owner: the path left the predicted tiles)
        const bool isD = cont && (cls & 2u) != 0, isI = cont && !isD && (cls & 1u) != 0, isX = cont && !isD && !isI;
        const bool owner = inr && (int)pcell == ub;
        if (isX) { switch_op((int)OP_X); R.cur_len += 1; steps += 1u; v -= 1; h -= 1; }
        if (__any(isD)) {                             // the vertical run: Pv bits from the cell's row upwards while set
            const int bit = v & 63;
            u32 rd = 0;
            if (isD && owner) {
                const u64 Pw = sel_col<TW>(tP, 15u - cu);
                rd = (u32)min(__clzll((long long)~(Pw << (63 - bit))), bit + 1);
            }
            rd = grp_max<LG>(rd);
            if (isD) { switch_op((int)OP_D); R.cur_len += (int)rd; steps += rd; v -= (int)rd; }
        }
        if (__any(isI)) {
            // one step -- or the whole horizontal run when the cell to the left is an insertion as well
            u32 more = 0;
            if (isI && owner && cu < 15u && o < 15u) {
                const u32 o1 = o + 1u;
                more = ((sel16(M8, o1) >> (cu + 1u)) & 1u) & ~((sel16(P8, o1) >> (cu + 1u)) & 1u);
            }
            more = grp_max<LG>(more);
            int ri = 1;
            if (__any(isI && more != 0)) {
                const int bit = v & 63;
                const bool match = act && Rb == (v >> 6);
                u32 go16 = 0;                         // bit u: the cell of column 15 - u in the walker's row is an insertion (Mv set, Pv clear)
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const u32 mm = (bit & 32) ? hi32(tM[c]) : lo32(tM[c]), pp = (bit & 32) ? hi32(tP[c]) : lo32(tP[c]);
                    go16 |= (((mm & ~pp) >> (bit & 31)) & 1u) << (15 - c);
                }
                go16 &= (inb_same ? 0xFFFEu : 0u) | (inb_7 ? 1u : 0u);
                const int un = u0 + r, ln = un - u_base;           // the walker's place now, after the run of matches
                const u32 keep_n = (ln <= 0) ? 0xFFFFu : ((ln > 15) ? 0u : ((0xFFFFu << ln) & 0xFFFFu));
                const u32 st = ~go16 & keep_n;
                u32 pi = (match && (isI && more != 0)) ? (st ? (u32)u_base + (u32)__builtin_ctz(st) : 1000u) : 2000u;
                pi = min(pi, pair_swap(pi));
                // no tile of this column pair is in the walker's row: the run's data ends at the pair's right edge
                if (pi == 2000u) pi = (u_base + 15 < un) ? 1000u : (u32)max(u_base, un + 1);
                pi = grp_min<LG>(pi);
                ri = (more != 0) ? (int)min(pi, (u32)SPAN) - un : 1;
            }
            if (isI) { switch_op((int)OP_I); R.cur_len += ri; steps += (u32)ri; h -= ri; }
        }
        going = cont;
        asm("" : "+v"(v), "+v"(h));                   // (see walk_tile_lean)
    }
}

// ===========================================================================
// BandEd traceback with G = 4 / 8 / 16 LANES PER ALIGNMENT (k_traceback_sys<log2 G>), for launches of few waves: there
// k_traceback's duration is one lane's chain of ~(n / 16 + m / 64) rounds, each a dependent load (the checkpoint the walk
// has just decided on), 16 block steps of recompute and the walk of 16 columns.  Here a round of the group rebuilds G tiles
// at once -- the G / 2 column tiles to the left of the walk's position, for each the block row the path is expected in
// (a path runs along its diagonal: 16 rows up per column tile) and the block row above it; every lane loads its tile's
// checkpoint / carry words / planes and runs the 16 block steps, all in the time one tile takes -- and then the walk
// visits them in path order: the walker's state (v, h, the run under construction) is the same in all G lanes, the lane
// whose tile the walker is in walks it (walk_tile, the same code and therefore the same steps, runs and bytes as
// k_traceback), its state is broadcast to the group, and the next owner takes over.  A path that leaves the predicted
// tiles (an indel run longer than the prediction's slack) just ends the round early; the next round starts from where
// the walker really is.  One memory latency and one recompute per ~8 G columns instead of per 16.
// Tasks with N or non-canonical symbols are flagged (o_abort) and left to k_traceback.
// ===========================================================================
template <int LG>
__global__ __launch_bounds__(256) void k_traceback_sys(TraceArgs A) {
    if (A.prio) __builtin_amdgcn_s_setprio(3);          // few waves, each a serial chain: first in line at the SIMD's issue arbiter
    constexpr int GL = 1 << LG, NT = 64 >> LG;                        // lanes per task, tasks per wave
    const int wv = QE_GROUP_INDEX(), lane = threadIdx.x & 63, j = lane & (GL - 1), gl = lane & ~(GL - 1);
    const int t = wv * NT + (lane >> LG);
    if (wv * NT >= A.T.ntasks) return;
    const int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    const bool valid = pair >= 0;
    int m = 1, n = 1, p0 = 0, t0 = 0, cut_in = 0;
    const u64* pp = A.P.pl_p; const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t]; cut_in = A.T.cutoff[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair]; tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const bool ok = valid && (fl & (FLAG_HAS_N | FLAG_NONCANON)) == 0;
    if (valid && j == 0) A.o_abort[t] = ok ? 0 : 1;
    if (!__any(ok)) return;
    const Geom G = band_geometry(m, n, cut_in);
    const int nw = (m + 63) >> 6;
    const int g = ok ? (t >> 6) : 0, col = t & 63;
    const int gns = A.g_nslots[g], gnch = A.g_nch[g];
    const GroupWs W = group_ws(const_cast<uint8_t*>(A.ws), A.g_ws_off[g], gns, A.g_nrows[g], gnch);
    const int16_t* cf = W.cf + col;
    const int16_t* cl = W.cl + col;
    const uint4* cp = A.mat + A.g_mat_off[g] + col;
    const int64_t cps = (int64_t)gns * 64;
    const uint4* hw = cp + (int64_t)QE_CPC * gnch * cps;
    RunSink R;
    {
        const int cap = ok ? A.g_runs_cap[g] : 0;
        R.init(ok ? A.runs + A.g_runs_off[g] + (A.runs_by_task ? (int64_t)col * cap : (int64_t)col) : nullptr, cap, A.runs_by_task ? 1 : 64);
    }
    int h = n - 1, v = m - 1;
    u32 steps = 0;
    int nmatch = 0;
    constexpr int TW = QE_CP_COLS;
#ifdef QE_TBS_PROF      // tools/tbs_prof.sh: where a round's cycles go (s_memtime at the phase boundaries, printed by the first lane)
    long long pf_t = __builtin_readcyclecounter(), pf_load = 0, pf_comp = 0, pf_walk = 0, pf_bcast = 0; int pf_rounds = 0, pf_iters = 0;
#define PF_MARK(acc) do { const long long n__ = __builtin_readcyclecounter(); acc += n__ - pf_t; pf_t = n__; } while (0)
#else
#define PF_MARK(acc) do { } while (0)
#endif
    // every load of a tile in one go: the checkpoint and the carry words are fetched for the slot clamped into the group's
    // range before the band-edge records say whether the slot was computed (one memory latency, not two); planes as raw words
    auto tbs_fetch = [&](int q, int Rb) {
        TbsFetch F;
        const int k = q / (64 / TW), se = min(max(Rb - (k - G.prolog), 0), gns - 1);
        F.q = q; F.Rb = Rb;
        F.cf_a = cf[(int64_t)(k + 1) * 64]; F.cf_b = cf[(int64_t)k * 64]; F.cl_b = cl[(int64_t)k * 64];
        F.c0 = cp[(int64_t)q * cps + (int64_t)se * 64];
        F.w0 = hw[((int64_t)k * gns + se) * 64];
        planes_ab_raw(tp, t0 + 64 * k, F.t);
        planes_ab_raw(pp, p0 + 64 * Rb, F.p);
        return F;
    };
    TbsFetch pre;
    pre.q = -1; pre.Rb = -1; pre.cf_a = 0; pre.cf_b = 0; pre.cl_b = -1; pre.c0 = make_uint4(0, 0, 0, 0); pre.w0 = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) { pre.t[c] = 0; pre.p[c] = 0; }
    while (__any(ok && v >= 0 && h >= 0)) {
        const bool live = ok && v >= 0 && h >= 0;
        // this lane's tile: column tile q0 - x, block row b0(x) or the one above
        const int x = j >> 1;
        const int q = (h / TW) - x;
        const int v_in = (x == 0) ? v : v - (h - (TW * q + TW - 1));          // where the diagonal meets the tile's right edge
        const int Rb = (max(v_in, 0) >> 6) - (j & 1);
        const bool act = live && q >= 0 && Rb >= 0;
        const int k = max(q, 0) / (64 / TW);
        u64 tP[TW], tM[TW], tE[TW];
#pragma unroll
        for (int c = 0; c < TW; ++c) { tP[c] = 0; tM[c] = 0; tE[c] = 0; }
        u32 inb_same = 0, inb_7 = 0;
        u64 P = 0, M = 0, T0 = 0, T1 = 0, hinP = 0, hinM = 0, pa = 0, pb = 0;
        bool computed = false;
        int cl_b = -1;
        // what the tile needs from memory: fetched a round ahead for the tile the walker is expected in next (pre), here only if
        // that guess was wrong (the first round, a walker that ended the round off its diagonal's block row)
        TbsFetch F = pre;
        if (act && !(pre.q == q && pre.Rb == Rb)) F = tbs_fetch(q, Rb);
        if (act) {
            const int pos_v = k - G.prolog, s = Rb - pos_v;
            const int cf_a = F.cf_a, cf_b = F.cf_b;
            cl_b = F.cl_b;
            const uint4 c0 = F.c0, w0 = F.w0;
            planes_ab_finish(F.t, t0 + 64 * k, T0, T1);
            planes_ab_finish(F.p, p0 + 64 * Rb, pa, pb);
            // a step at column h reads Pv of stored column h + 1: inside the band of THAT column's chunk or 0 (oracle header)
            inb_same = (u32)((s >= 0) & (s >= cf_b) & (s <= cl_b));
            inb_7 = ((q & (64 / TW - 1)) == 64 / TW - 1) ? (u32)((s >= 1) & (s - 1 >= cf_a) & (s - 1 <= cl_b)) : inb_same;
            computed = s >= cf_b && s <= min(cl_b, nw - 1 - pos_v);
            if (computed) {
                hinP = mk64(w0.x, w0.y); hinM = mk64(w0.z, w0.w);
                P = mk64(c0.x, c0.y); M = mk64(c0.z, c0.w);
            }
        }
#ifdef QE_TBS_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PF_MARK(pf_load); ++pf_rounds;
#endif
        {   // TW block steps from the checkpoint: the fill's arithmetic, so its bits
            const int c_first = (TW * max(q, 0)) & 63;
            const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
            const u32 t0s = (u32)(T0 >> c_first), t1s = (u32)(T1 >> c_first);
            const u32 hp = (u32)(hinP >> c_first), hm = (u32)(hinM >> c_first);
            u32 Plo = lo32(P), Phi = hi32(P), Mlo = lo32(M), Mhi = hi32(M), gP = 0, gM = 0;
#pragma unroll
            for (int c = 0; c < TW; ++c) {
                const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, c, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, c, 1);
                const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                tE[c] = mk64(elo, ehi);
                tM[c] = mk64(Mlo, Mhi);
                block_step_fused(elo, ehi, Plo, Phi, Mlo, Mhi, __builtin_amdgcn_ubfe(hp, c, 1), __builtin_amdgcn_ubfe(hm, c, 1), gP, gM);
                tP[c] = mk64(Plo, Phi);
            }
        }
        if (!computed) {
#pragma unroll
            for (int c = 0; c < TW; ++c) { tP[c] = 0; tM[c] = 0; }
            // the row the band bookkeeping creates at the end of a chunk: Pv = ~0, Mv = 0 (bpm_banded.c:910)
            if (act && (Rb - (k - G.prolog)) == cl_b + 1 && (q & (64 / TW - 1)) == 64 / TW - 1) tP[TW - 1] = QE_ONES;
        }
        {   // the next round's tile, if the walker keeps to its diagonal: GL / 2 column tiles to the left, 16 rows up per column tile
            const int qn = q - GL / 2, vn = v_in - TW * (GL / 2);
            const int Rbn = (max(vn, 0) >> 6) - (j & 1);
            pre.q = -1; pre.Rb = -1;
            if (live && qn >= 0 && vn >= 0 && Rbn >= 0) pre = tbs_fetch(qn, Rbn);
        }
        if (LG >= 3) {
            // 16 / 8 lanes per leaf (four / eight walkers per wave): the tiles' cells laid out along the diagonals near the walker's, and the
            // walk over the round's 128 columns in all lanes at once (walk_round_diag)
            u32 X8[8], P8[8], M8[8];
            const int d0 = v - h;
            diag_words<TW>(tP, tM, tE, inb_same, inb_7, d0 + TW * q - 64 * Rb, X8, P8, M8);
            if (!act) {
#pragma unroll
                for (int c = 0; c < 8; ++c) X8[c] = ~0u;
            }
            PF_MARK(pf_comp);
            walk_round_diag<TW, LG>(tP, tM, X8, P8, M8, live, act, x, Rb, d0 + TW * q - 64 * Rb, d0, TW * (h / TW) + TW - 1, inb_same, inb_7, j == 0,
                                v, h, steps, nmatch, R);
            PF_MARK(pf_walk);
            continue;
        }
        // the walk, tile by tile in path order
        bool first_phase = true;
        PF_MARK(pf_comp);
        while (true) {
            const bool mine = act && live && v >= 0 && h >= 0 && (h / TW) == q && (v >> 6) == Rb;
            const u64 bal = __ballot(mine);
            const u32 grp = (u32)(bal >> gl) & ((1u << GL) - 1u);
            if (!__any(grp != 0)) break;
            PF_MARK(pf_bcast);
            walk_tile_lean<TW>(tP, tM, tE, mine, inb_same, inb_7, Rb, v, h, steps, nmatch, R);
#ifdef QE_TBS_PROF
            PF_MARK(pf_walk); ++pf_iters;
#endif
            if (grp != 0) {                                         // (uniform over the lanes of a group)
                const int own = gl | (__ffs((int)grp) - 1);
                v = __shfl(v, own); h = __shfl(h, own); steps = (u32)__shfl((int)steps, own); nmatch = __shfl(nmatch, own);
                R.nruns = __shfl(R.nruns, own); R.cur_op = __shfl(R.cur_op, own); R.cur_len = __shfl(R.cur_len, own);
            }
            first_phase = false;
        }
        (void)first_phase;
        PF_MARK(pf_bcast);
    }
#ifdef QE_TBS_PROF
    if (wv == 0 && lane == 0) printf("k_traceback_sys<%d>: %d rounds, %d tile walks; cycles: loads %lld, recompute %lld, walks %lld, hand-over %lld\n", LG, pf_rounds, pf_iters, pf_load, pf_comp, pf_walk, pf_bcast);
#endif
    if (!ok || j != 0) return;
    R.nops = (int)steps; R.edits = (int)steps - nmatch;
    R.push_n(OP_I, h + 1);
    R.push_n(OP_D, v + 1);
    R.flush();
    A.o_nruns[t] = (R.nruns <= R.cap) ? R.nruns : -1;
    A.o_nops[t] = R.nops;
    A.o_edits[t] = R.edits;
    A.o_steps[t] = steps;
}

template __global__ void k_traceback_sys<2>(TraceArgs);
template __global__ void k_traceback_sys<3>(TraceArgs);
template __global__ void k_traceback_sys<4>(TraceArgs);

// In-window traceback over one recomputed 8-column tile of the window's last block row (W = 2, O = 1:
// the walk stays in window rows 64..127 and columns 64..127, bpm_windowed.c:448-561).  vw / hw are window
// coordinates.  SCORE_ONLY: D -> I -> match -> X, cost only (527-549); else match -> D -> I -> X (476-495).
template <bool SCORE_ONLY>
__device__ __forceinline__ void window_walk_tile(const u64 (&tP)[8], const u64 (&tM)[8], const u64 (&tE)[8],
                                                 bool& inr, int& vw, int& hw, int& wscore, RunSink& R) {
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const bool mine = inr && (hw & 7) == j;
        const int bit = vw & 63;
        const u64 del = SCORE_ONLY ? tP[j] : (tP[j] & ~tE[j]);       // cells that take a deletion
        int r = min(__clzll((long long)~(del << (63 - bit))), bit + 1);
        if (!mine) r = 0;
        const bool up = r == bit + 1;                                 // the run left the block row: window done
        const bool go = mine && !up;
        const int b1 = (bit - r) & 63;
        const u32 mb = (u32)((tM[j] >> b1) & 1), eq = (u32)((tE[j] >> b1) & 1);
        u32 isI;
        if (SCORE_ONLY) {
            isI = mb;
            wscore += r + (go ? (int)(isI | (eq ^ 1u)) : 0);
        } else {
            isI = mb & (eq ^ 1u);
            R.emit((int)OP_D, r, r > 0);
            R.emit(eq ? (int)OP_M : (isI ? (int)OP_I : (int)OP_X), 1, go);
        }
        vw -= r + ((go && !isI) ? 1 : 0);
        hw -= go ? 1 : 0;
        inr = inr && !(mine && up) && vw >= 64;
    }
}

// The same walk for any window shape (bpm_windowed.c:448-561): the tile is 8 columns of window block row Rb; the lane
// leaves it to the left, upwards (the next round recomputes the tile above) or by leaving the traceback region
// (v < v_ov or h < h_ov: the window is done).  v / h are alignment coordinates, v0 / h0 the window's origin.
template <bool SCORE_ONLY>
__device__ __forceinline__ void window_walk_tile_g(const u64 (&tP)[8], const u64 (&tM)[8], const u64 (&tE)[8], bool inw, int Rb,
                                                   int v0, int h0, int v_ov, int h_ov, int& v, int& h, int& wscore, RunSink& R) {
    bool in_tile = inw;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const bool mine = in_tile && ((h - h0) & 7) == j;
        const int bit = (v - v0) & 63;
        const u64 del = SCORE_ONLY ? tP[j] : (tP[j] & ~tE[j]);       // cells that take a deletion
        int r = min(__clzll((long long)~(del << (63 - bit))), bit + 1);
        r = min(r, v - v_ov + 1);                                      // the region ends at row v_ov
        if (!mine) r = 0;
        const bool up = r == bit + 1;                                 // the run left the block row
        const bool go = mine && !up && (v - r) >= v_ov;
        const int b1 = (bit - r) & 63;
        const u32 mb = (u32)((tM[j] >> b1) & 1), eq = (u32)((tE[j] >> b1) & 1);
        u32 isI;
        if (SCORE_ONLY) {
            isI = mb;
            wscore += r + (go ? (int)(isI | (eq ^ 1u)) : 0);
        } else {
            isI = mb & (eq ^ 1u);
            R.emit((int)OP_D, r, r > 0);
            R.emit(eq ? (int)OP_M : (isI ? (int)OP_I : (int)OP_X), 1, go);
        }
        v -= r + ((go && !isI) ? 1 : 0);
        h -= go ? 1 : 0;
        in_tile = in_tile && !(mine && up) && v >= v_ov && h >= h_ov && ((v - v0) >> 6) == Rb;
    }
}

// window_walk_tile_g<true> for a wave with few walkers (k_windowed_sys): where the walker's cell is a plain match -- tX =
// Pv | Mv | ~Eq has a zero at its row -- the step is v--, h-- at no cost; everything else takes the general column behind
// one branch.  Same steps, same score.
__device__ __forceinline__ void window_walk_tile_g_fast(const u64 (&tP)[8], const u64 (&tM)[8], const u64 (&tE)[8], const u64 (&tX)[8], bool inw, int Rb,
                                                        int v0, int h0, int v_ov, int h_ov, int& v, int& h, int& wscore) {
    bool in_tile = inw;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const bool mine = in_tile && ((h - h0) & 7) == j;
        const int bit = (v - v0) & 63;
        const bool slow = mine && ((u32)(tX[j] >> bit) & 1u) != 0;
        if (slow) {
            int r = min(__clzll((long long)~(tP[j] << (63 - bit))), bit + 1);
            r = min(r, v - v_ov + 1);                                  // the region ends at row v_ov
            const bool up = r == bit + 1;                             // the run left the block row
            const bool go = !up && (v - r) >= v_ov;
            const int b1 = (bit - r) & 63;
            const u32 mb = (u32)(tM[j] >> b1) & 1u, eq = (u32)(tE[j] >> b1) & 1u;
            wscore += r + (go ? (int)(mb | (eq ^ 1u)) : 0);
            v -= r + ((go && !mb) ? 1 : 0);
            h -= go ? 1 : 0;
            in_tile = !up;
        }
        const int f = (mine && !slow) ? 1 : 0;                        // a plain match
        v -= f; h -= f;
        in_tile = in_tile && v >= v_ov && h >= h_ov && ((v - v0) >> 6) == Rb;
        asm("" : "+v"(v), "+v"(h));                                   // (see walk_tile_lean)
    }
}

// K vertically adjacent blocks i .. i + K - 1 of a window over the 64 columns of chunk j (window_cp_* state: see the
// checkpointed general path of k_windowed)
template <int K>
__device__ __forceinline__ void window_cp_pass(int i, int j, int W, int steps_v, int blk_min, bool chunk_on, bool chunk_keep,
                                               u64* Pv, u64* Mv, const u64* pp, int prow0, u64 pinit, u64 T0, u64 T1,
                                               u64& hinP, u64& hinM, uint4* cpb, uint4* hwb) {
    u64 P[K], M[K], a[K], b[K], oP[K], oM[K];
    bool act[K], keep[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        act[k] = chunk_on && (i + k) < steps_v;
        keep[k] = act[k] && chunk_keep && (i + k) >= blk_min;
        P[k] = pinit; M[k] = 0; a[k] = 0; b[k] = 0;
        if (act[k]) {
            u64 nn;
            if (j > 0) { P[k] = Pv[(int64_t)(i + k) * 64]; M[k] = Mv[(int64_t)(i + k) * 64]; }
            load_planes(pp, prow0 + 64 * (i + k), a[k], b[k], nn);
        }
    }
    run64_skew_cp<K>(P, M, a, b, T0, T1, hinP, hinM, oP, oM, keep, cpb + ((int64_t)(8 * j) * W + i) * 64, (int64_t)W * 64);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (act[k]) { Pv[(int64_t)(i + k) * 64] = P[k]; Mv[(int64_t)(i + k) * 64] = M[k]; }
        if (keep[k]) {
            const u64 iP = (k == 0) ? hinP : oP[k > 0 ? k - 1 : 0], iM = (k == 0) ? hinM : oM[k > 0 ? k - 1 : 0];
            hwb[((int64_t)j * W + (i + k)) * 64] = make_uint4(lo32(iP), hi32(iP), lo32(iM), hi32(iM));
        }
    }
    hinP = oP[K - 1]; hinM = oM[K - 1];
}

// ===========================================================================
// WindowEd chain (bpm_windowed.c:563-628): windows of W x W blocks anchored at
// the current traceback position, filled (202-280; SSE semantics 283-445 when
// sse && W == 2), traced back inside the non-overlap region (448-561).
// ===========================================================================
// CP: the general path (any W, O; partial windows) keeps checkpoints + carry words instead of every column's history
template <bool CP>
__device__ __forceinline__ void windowed_body(const WindowArgs& A) {
    uint4 (*wck)[64] = (uint4 (*)[64])(qe_dyn_lds + QE_WAVE_IN_BLOCK() * 512);    // [8][64] per wave
    const int g = QE_GROUP_INDEX(), lane = threadIdx.x & 63, t = g * 64 + lane;
    if (g * 64 >= A.T.ntasks) return;
    int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    if (A.only_if != nullptr && pair >= 0 && A.only_if[t] == 0) pair = -1;      // done by k_windowed_sys
    const bool valid = pair >= 0;
    if (!__any(valid)) return;
    const int W = A.W, O = A.O;
    int m = 0, n = 0, p0 = 0, t0 = 0;
    const u64* pp = A.P.pl_p; const u64* tp = A.P.pl_t;
    u32 fl = 0;
    EqTest E; E.pp = pp; E.tp = tp; E.ap = nullptr; E.at = nullptr; E.raw = false;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair]; tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
        E.pp = pp; E.tp = tp;
        E.ap = A.P.asc_p + A.P.asc_p_off[pair]; E.at = A.P.asc_t + A.P.asc_t_off[pair];
        E.raw = (fl & FLAG_NONCANON) != 0;
        if (A.reversed) { E.rp = A.P.p_len[pair] - 1; E.rt = A.P.t_len[pair] - 1; }
    }
    const bool hasN = (fl & FLAG_HAS_N) != 0;
    const bool sse = A.sse && W == 2;                               // bpm_windowed.c:577
    uint8_t* wsb = A.ws + A.g_ws_off[g];
    u64* const Pv = (u64*)wsb + lane;                               // [W][64]
    u64* const Mv = (u64*)wsb + (int64_t)W * 64 + lane;
    uint4* const hist = (uint4*)(wsb + (int64_t)2 * W * 64 * 8) + lane * 8;   // tiled like the fill matrix, nslots = W
    const int64_t tstride = (int64_t)W * 512;
    RunSink R;
    R.init(A.score_only ? nullptr : A.runs + A.g_runs_off[g] + lane, A.score_only ? 0 : A.g_runs_cap[g]);
    int pos_v = m - 1, pos_h = n - 1;
    int score = 0, hew = 0;
    u32 steps = 0;
    if (A.state && valid) {                                         // the chain so far: k_windowed_quad's
        const int64_t nt = A.T.ntasks;
        pos_v = A.state[t]; pos_h = A.state[nt + t]; score = A.state[2 * nt + t]; hew = A.state[3 * nt + t];
        steps = (u32)A.state[4 * nt + t];
    }

    const bool w2 = W == 2 && O == 1 && !__any(valid && (hasN || E.raw));
    const bool cp_ok = CP && !sse && A.cp_path != 0 && !__any(valid && (hasN || E.raw));
    // the checkpointed general path has a loop of its own: nothing of the paths below (match test, history pointers)
    // stays live across its windows
    if (CP && cp_ok) {
        while (__any(valid && pos_v >= 0 && pos_h >= 0)) {
            const bool on = valid && pos_v >= 0 && pos_h >= 0;
            // ---- general window, checkpointed: the fill runs K = 3 blocks per skewed pass and leaves {Pv, Mv} before
            // every 8th column of every block the traceback may visit plus the carry-in words per (chunk, block) -- 2.25
            // instead of 16 B per block-column -- and the traceback recomputes the 8-column tile it is in (same
            // arithmetic, same bits as the stored history of the path below; tested against it: QE_WINDOWED_CP = 0)
            const int v_fi = pos_v, h_fi = pos_h;
            const int v0 = max(v_fi - 64 * W + 1, 0), h0 = max(h_fi - 64 * W + 1, 0);
            const int steps_v = on ? (v_fi - v0) / 64 + 1 : 0;
            const int ncols_total = on ? h_fi - h0 + 1 : 0;
            const u64 ph_first = (v0 == 0) ? QE_ONES : 0;
            const u64 pinit = (h0 == 0) ? QE_ONES : 0;
            const int v_ov = max(v_fi - 64 * (W - O) + 1, 0), h_ov = max(h_fi - 64 * (W - O) + 1, 0);
            const int blk_min = (v_ov - v0) >> 6;
            const int nchunk = wave_max((ncols_total + 63) >> 6), nblk = wave_max(steps_v);
            uint4* const cpb = (uint4*)(wsb + (int64_t)2 * W * 64 * 8) + lane;     // [8 x chunks][W][64]: {Pv, Mv} before column 8 q
            uint4* const hwb = cpb + (int64_t)8 * W * W * 64;                      // [chunks][W][64]: carry-in words
            for (int j = 0; j < nchunk; ++j) {
                const bool chunk_on = ncols_total > 64 * j;
                const bool chunk_keep = 64 * j + 63 >= h_ov - h0;
                u64 T0 = 0, T1 = 0, TN = 0;
                if (chunk_on) load_planes(tp, t0 + h0 + 64 * j, T0, T1, TN);
                u64 hinP = ph_first, hinM = 0;
                for (int i = 0; i < nblk;) {
                    const int rem = nblk - i;
                    if (rem >= 3) { window_cp_pass<3>(i, j, W, steps_v, blk_min, chunk_on, chunk_keep, Pv, Mv, pp, p0 + v0, pinit, T0, T1, hinP, hinM, cpb, hwb); i += 3; }
                    else if (rem == 2) { window_cp_pass<2>(i, j, W, steps_v, blk_min, chunk_on, chunk_keep, Pv, Mv, pp, p0 + v0, pinit, T0, T1, hinP, hinM, cpb, hwb); i += 2; }
                    else { window_cp_pass<1>(i, j, W, steps_v, blk_min, chunk_on, chunk_keep, Pv, Mv, pp, p0 + v0, pinit, T0, T1, hinP, hinM, cpb, hwb); i += 1; }
                }
            }
            if (on) steps += (u32)steps_v * (u32)ncols_total;
            // in-window traceback (bpm_windowed.c:448-561) over recomputed tiles
            int v = pos_v, h = pos_h, wscore = 0;
            int cRb = -1, cj = -1;
            u64 pa = 0, pb = 0, X0 = 0, X1 = 0, hP = 0, hM = 0;
            while (__any(on && v >= v_ov && h >= h_ov)) {
                const bool inw = on && v >= v_ov && h >= h_ov;
                const int Rb = inw ? (v - v0) >> 6 : 0, q = inw ? (h - h0) >> 3 : 0, j = q >> 3;
                u32 Plo = 0, Phi = 0, Mlo = 0, Mhi = 0;
                if (inw) {
                    u64 nn;
                    const bool nr = Rb != cRb, nj = j != cj;
                    if (nr) load_planes(pp, p0 + v0 + 64 * Rb, pa, pb, nn);
                    if (nj) load_planes(tp, t0 + h0 + 64 * j, X0, X1, nn);
                    if (nr | nj) {
                        const uint4 w0 = hwb[((int64_t)j * W + Rb) * 64];
                        hP = mk64(w0.x, w0.y); hM = mk64(w0.z, w0.w);
                    }
                    cRb = Rb; cj = j;
                    const uint4 c0 = cpb[((int64_t)q * W + Rb) * 64];
                    Plo = c0.x; Phi = c0.y; Mlo = c0.z; Mhi = c0.w;
                }
                const int sh = 8 * (q & 7);
                const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
                const u32 t0s = (u32)(X0 >> sh), t1s = (u32)(X1 >> sh);
                const u32 hp = (u32)(hP >> sh), hm = (u32)(hM >> sh);
                u32 aP = 0, aM = 0;
                u64 tP[8], tM[8], tE[8];
    #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, c, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, c, 1);
                    const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                    tE[c] = mk64(elo, ehi);
                    tM[c] = mk64(Mlo, Mhi);
                    block_step_fused(elo, ehi, Plo, Phi, Mlo, Mhi, __builtin_amdgcn_ubfe(hp, c, 1), __builtin_amdgcn_ubfe(hm, c, 1), aP, aM);
                    tP[c] = mk64(Plo, Phi);
                }
                if (A.score_only) window_walk_tile_g<true>(tP, tM, tE, inw, Rb, v0, h0, v_ov, h_ov, v, h, wscore, R);
                else window_walk_tile_g<false>(tP, tM, tE, inw, Rb, v0, h0, v_ov, h_ov, v, h, wscore, R);
            }
            if (on) {
                if (A.score_only) {
                    if (wscore > (W - O) * 64 * A.hew_threshold / 100) ++hew;
                    score += wscore;
                }
                pos_h = h; pos_v = v;
            }
        }
    } else
    while (__any(valid && pos_v >= 0 && pos_h >= 0)) {
        const bool on = valid && pos_v >= 0 && pos_h >= 0;
        if (w2 && !__any(on && (pos_v < 127 || pos_h < 127))) {
            // ---- every live lane has a full 128 x 128 window: the whole window stays on chip.  Fill the four
            // block-chunks in registers, keep {Pv, Mv} of the last one every 8 columns in LDS, and let the
            // traceback recompute 8-column tiles from those (same arithmetic, same bits as a stored history)
            const int v0 = on ? pos_v - 127 : 0, h0 = on ? pos_h - 127 : 0;
            u64 pl0[3] = {0, 0, 0}, pl1[3] = {0, 0, 0}, tx0[3] = {0, 0, 0}, tx1[3] = {0, 0, 0};
            if (on) { load_planes2(pp, p0 + v0, pl0, pl1); load_planes2(tp, t0 + h0, tx0, tx1); }
            const u64 ph_first = (v0 == 0) ? QE_ONES : 0;
            const u64 pinit = (h0 == 0) ? QE_ONES : 0;
            u64 P0 = pinit, M0 = 0, P1 = pinit, M1 = 0;
            u64 gP, gM, xP, xM;
            {   // first 64 columns: both blocks in one pass (shared text masks, register carries; nothing to collect)
                u64 Pw[2] = {P0, P1}, Mw[2] = {M0, M1};
                const u64 aw[2] = {pl0[0], pl1[0]}, bw[2] = {pl0[1], pl1[1]};
                run64_multi<2>(Pw, Mw, aw, bw, tx0[0], tx0[1], sse ? (0x5555555555555556ull | (ph_first & 1)) : ph_first, 0, xP, xM);
                P0 = Pw[0]; M0 = Mw[0]; P1 = Pw[1]; M1 = Mw[1];
            }
            {   // second 64 columns: both blocks in one skewed pass; block 1 leaves its checkpoints in LDS
                u64 Eq_x = 0;
                if (sse) {
                    // the SSE kernel runs block 0 one column past the window and feeds THAT column's carries to
                    // block 1's last column (bpm_windowed.c:428-444); 127 is odd, so always here
                    int tc = 4;                                              // text[tlen] reads as N (A.7(4))
                    if (on && pos_h + 1 < n) tc = plane_code(tp, t0 + pos_h + 1);
                    Eq_x = (tc == 4) ? pl0[2] : (~(pl0[0] ^ ((u64)0 - (u64)(tc & 1))) & ~(pl0[1] ^ ((u64)0 - (u64)((tc >> 1) & 1))) & ~pl0[2]);
                }
                run64_win2(P0, M0, P1, M1, pl0[0], pl0[1], pl1[0], pl1[1], tx1[0], tx1[1], sse ? 0x5555555555555555ull : ph_first,
                           sse, Eq_x, gP, gM, &wck[0][lane], 64);
            }
            if (on) steps += 256u;
            int vw = 127, hw = 127, wscore = 0;
            bool inr = on;
            const u32 alo = lo32(pl1[0]), ahi = hi32(pl1[0]), blo = lo32(pl1[1]), bhi = hi32(pl1[1]);
#pragma unroll 1
            for (int q = 7; q >= 0 && __any(inr); --q) {
                const uint4 c0 = wck[q][lane];
                const u32 t0s = (u32)(tx1[0] >> (8 * q)), t1s = (u32)(tx1[1] >> (8 * q));
                const u32 hp = (u32)(gP >> (8 * q)), hm = (u32)(gM >> (8 * q));
                u32 Plo = c0.x, Phi = c0.y, Mlo = c0.z, Mhi = c0.w, aP = 0, aM = 0;
                u64 tP[8], tM[8], tE[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, j, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, j, 1);
                    const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                    tE[j] = mk64(elo, ehi);
                    tM[j] = mk64(Mlo, Mhi);
                    block_step_fused(elo, ehi, Plo, Phi, Mlo, Mhi, __builtin_amdgcn_ubfe(hp, j, 1), __builtin_amdgcn_ubfe(hm, j, 1), aP, aM);
                    tP[j] = mk64(Plo, Phi);
                }
                if (A.score_only) window_walk_tile<true>(tP, tM, tE, inr, vw, hw, wscore, R);
                else window_walk_tile<false>(tP, tM, tE, inr, vw, hw, wscore, R);
            }
            if (on) {
                if (A.score_only) {
                    if (wscore > (W - O) * 64 * A.hew_threshold / 100) ++hew;
                    score += wscore;
                }
                pos_v = v0 + vw; pos_h = h0 + hw;
            }
            continue;
        }
        const int v_fi = pos_v, h_fi = pos_h;
        const int v0 = max(v_fi - 64 * W + 1, 0), h0 = max(h_fi - 64 * W + 1, 0);
        const int steps_v = on ? (v_fi - v0) / 64 + 1 : 0;
        const int steps_h = h_fi - h0;
        const int ncols_total = on ? steps_h + 1 : 0;
        const u64 ph_first = (v0 == 0) ? QE_ONES : 0;               // D[0][c] = c only on the real first row
        // left boundary: Pv = ~0 on the real first column, else free start (bpm_windowed.c:226-230)
        if (on) {
            for (int i = 0; i < W; ++i) {
                const u64 pinit = (h0 == 0) ? QE_ONES : 0;
                Pv[(int64_t)i * 64] = pinit;
                Mv[(int64_t)i * 64] = 0;
                hist[tile_elem(0, i, W)] = make_uint4(lo32(pinit), hi32(pinit), 0u, 0u);
            }
        }
        // the in-window traceback only visits rows >= v_ov and columns >= h_ov (448-561):
        // store the history of exactly those blocks / 64-column chunks
        const int blk_min = (max(v_fi - 64 * (W - O) + 1, 0) - v0) >> 6;
        const int col_min = max(h_fi - 64 * (W - O) + 1, 0) - h0 + 1;
        const int nchunk = wave_max((ncols_total + 63) >> 6);
        const int nblk = wave_max(steps_v);
        for (int j = 0; j < nchunk; ++j) {
            const int ncols = min(max(ncols_total - 64 * j, 0), 64);
            u64 T0 = 0, T1 = 0, TN = 0;
            if (ncols > 0) load_planes(tp, t0 + h0 + 64 * j, T0, T1, TN);
            u64 hinP, hinM = 0;
            if (sse) hinP = (j == 0) ? (0x5555555555555556ull | (ph_first & 1)) : 0x5555555555555555ull;
            else hinP = ph_first;
            for (int i = 0; i < nblk; ++i) {
                const bool act = ncols > 0 && i < steps_v;
                u64 P = 0, M = 0, a = 0, b = 0, nn = 0;
                if (act) {
                    P = Pv[(int64_t)i * 64];
                    M = Mv[(int64_t)i * 64];
                    load_planes(pp, p0 + v0 + 64 * i, a, b, nn);    // bit-unaligned window rows (237-244)
                }
                uint4* st = hist + tile_elem(64 * j, i, W);
                u64 houtP, houtM, sP, sM;
                const bool slow = act && (ncols != 64 || hasN);
                const bool keep = act && i >= blk_min && 64 * j + 64 >= col_min;
                if (!__any(slow)) {
                    run64_fast<1>(P, M, a, b, T0, T1, hinP, hinM, houtP, houtM, keep, st, tstride, st + 8 * tstride);
                } else {
                    run64_general<1>(P, M, a, b, nn, T0, T1, TN, hinP, hinM, houtP, houtM, sP, sM,
                                        63, act ? ncols : 0, keep, st, tstride, st + 8 * tstride);
                }
                if (act) { Pv[(int64_t)i * 64] = P; Mv[(int64_t)i * 64] = M; steps += (u32)ncols; }
                hinP = houtP; hinM = houtM;
            }
        }
        if (sse && on && steps_v == 2 && (steps_h & 1)) {
            // the SSE kernel runs block 0 one column past the window and redoes block 1's last
            // column with the carries of that extra column (bpm_windowed.c:428-444)
            u64 a, b, nn, P = Pv[0], M = Mv[0];
            load_planes(pp, p0 + v0, a, b, nn);
            const int tc = (h_fi + 1 < n) ? plane_code(tp, t0 + h_fi + 1) : 4;      // text[tlen] reads as N (A.7(4))
            u64 Eq = (tc == 4) ? nn : (~(a ^ ((u64)0 - (u64)(tc & 1))) & ~(b ^ ((u64)0 - (u64)((tc >> 1) & 1))) & ~nn);
            u64 Ph, Mh;
            block_step(Eq, P, M, 1u, 0u, Ph, Mh);                   // column steps_h + 1 is even: PHin = 1
            const u32 cP = (u32)(Ph >> 63), cM = (u32)(Mh >> 63);
            load_planes(pp, p0 + v0 + 64, a, b, nn);
            const uint4 q = hist[tile_elem(steps_h, 1, W)], q1 = hist[tile_elem(steps_h + 1, 1, W)];
            P = mk64(q.x, q.y); M = mk64(q1.z, q1.w);            // state before column steps_h: {Pv after the previous, Mv before this}
            const int tl = plane_code(tp, t0 + h_fi);
            Eq = (tl == 4) ? nn : (~(a ^ ((u64)0 - (u64)(tl & 1))) & ~(b ^ ((u64)0 - (u64)((tl >> 1) & 1))) & ~nn);
            block_step(Eq, P, M, cP, cM, Ph, Mh);
            hist[tile_elem(steps_h + 1, 1, W)] = make_uint4(lo32(P), hi32(P), q1.z, q1.w);
        }
        // in-window traceback (bpm_windowed.c:448-561)
        if (on) {
            int h = pos_h, v = pos_v;
            const int h_min = max(pos_h - 64 * W + 1, 0), h_ov = max(pos_h - 64 * (W - O) + 1, 0);
            const int v_min = max(pos_v - 64 * W + 1, 0), v_ov = max(pos_v - 64 * (W - O) + 1, 0);
            int wscore = 0;
            constexpr int LOOK = 8;      // mostly-diagonal path: fetch the next LOOK diagonal cells at once
            while (v >= v_ov && h >= h_ov) {
                uint4 Q[LOOK];
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    const int hj = max(h - j, h_min), vj = max(v - j, v_min);
                    Q[j] = hist[tile_elem(hj - h_min + 1, ((vj - v_min) >> 6) & 0xff, W)];
                }
                bool go = true;
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    if (go && v >= v_ov && h >= h_ov) {
                        const int bit = (v - v_min) & 63;                   // A.7(1)
                        const u32 pb = (u32)((mk64(Q[j].x, Q[j].y) >> bit) & 1);
                        const u32 mb = (u32)((mk64(Q[j].z, Q[j].w) >> bit) & 1);
                        const u32 eq = E.eq(p0 + v, t0 + h) ? 1u : 0u;
                        u32 isD, isI;
                        if (A.score_only) {                                 // D -> I -> match -> X (527-549)
                            isD = pb; isI = mb & (pb ^ 1u);
                            wscore += (int)(isD | isI | (eq ^ 1u));
                        } else {                                            // match -> D -> I -> X (476-495)
                            isD = pb & (eq ^ 1u); isI = mb & (eq ^ 1u) & (pb ^ 1u);
                            R.push(isD ? (int)OP_D : (isI ? (int)OP_I : (eq ? (int)OP_M : (int)OP_X)));
                        }
                        v -= (int)(isI ^ 1u);
                        h -= (int)(isD ^ 1u);
                        go = (isD | isI) == 0;
                    }
                }
            }
            if (A.score_only) {
                if (wscore > (W - O) * 64 * A.hew_threshold / 100) ++hew;
                score += wscore;
            }
            pos_h = h; pos_v = v;
        }
    }
    if (valid) {
        if (A.score_only) {
            if (pos_h >= 0) score += pos_h + 1;
            if (pos_v >= 0) score += pos_v + 1;
            A.o_score[t] = score;
            A.o_hew[t] = hew;
        } else {
            R.push_n(OP_I, pos_h + 1);
            R.push_n(OP_D, pos_v + 1);
            R.flush();
            A.o_nruns[t] = R.nruns;
            A.o_nops[t] = R.nops;
            A.o_edits[t] = R.edits;
            A.o_score[t] = R.edits;
            A.o_hew[t] = 0;
        }
        A.o_steps[t] = steps;
    }
}
__global__ __launch_bounds__(512) void k_windowed(WindowArgs A) { windowed_body<false>(A); }
__global__ __launch_bounds__(512) void k_windowed_cp(WindowArgs A) { windowed_body<true>(A); }

// ===========================================================================
// WindowEd(2, 1) with FOUR LANES PER ALIGNMENT (k_windowed_quad): the cooperative form of the on-chip window above, for
// launches of few waves whose duration is one lane's serial chain (a batch of a few thousand pairs, QuickEd's stage 1 of
// config 4, a single pair).  Lane j of a quad owns rows 32 j .. 32 j + 31 of the 128 x 128 window as ONE 32-bit block:
// the Myers step is exact for any partition of a column into blocks (bpm_commons.h:82-101 evaluates the recurrence cell
// by cell; the carries between 32-bit blocks are what the 64-bit add carries inside a block), so every Pv / Mv bit equals
// the reference's.  The four blocks form a systolic array skewed by one column per lane: at step s lane j works on
// column s - j, its carry-in is lane j - 1's carry-out of the step before (one v_mov_dpp quad_perm:[3,0,1,2] per carry
// bit; lane 0 takes the window's top boundary instead) -- 132 steps of one 32-bit block step instead of 256 64-bit
// ones, and every step's text bit is a literal position of the lane's own text words (loaded j bases early).  Lanes 2 and
// 3 leave {Pv after, Mv before} of columns 64 .. 127 in LDS; all four lanes then walk the traceback (bpm_windowed.c:504-
// 561) redundantly from those words, so the quad agrees on the next window's anchor without another exchange.
// x86 SSE semantics (bpm_windowed.c:283-445): the boundary pattern of the top carries, and block 1's last
// column taking the carries of block 0's column one past the window -- here lanes 0 and 1 run that extra column and lanes
// 2 and 3 run their last column two steps later than the skew alone would have them.
// Only full windows: a task's chain stops at its first clamped window (or at once: N / non-canonical symbols) and
// k_windowed takes it up from WindowArgs::state.
// ===========================================================================
__device__ __forceinline__ u32 quad_ror1(u32 x) {           // lane 4 q + j <- lane 4 q + (j + 3) % 4
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x93, 0xf, 0xf, false);
}
// One column of a lane's 32-bit block: block_step on 32-bit words (bitop3 forms as in block_step_core), text bit `bit` of the
// lane's words (t0w, t1w), Eq masked by emask.  The carries travel as the RAW pre-shift delta words (bit 31 = the carry): the
// producer extracts nothing, the consumer's "(Ph << 1) | PHin" is one v_alignbit with the incoming word; MHin arrives as a clean 0 / 1
template <bool STORE>
__device__ __forceinline__ void quad_col_w(u32 t0w, u32 t1w, int bit, u32 emask, u32 a, u32 b, u32& P, u32& M, u32 inPw, u32 MHin,
                                           u32& oPw, u32& oMw, uint2* st) {
    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0w, bit, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1w, bit, 1);
    const u32 e = bitop3<0x90>(~(a ^ m0), b, m1) & emask;
    const u32 Mb = M;
    const u32 xv = e | M;
    const u32 ec = e | MHin;
    const u32 sum = (ec & P) + P;
    const u32 ph = bitop3<0xF3>(M, bitop3<0xFE>(sum, P, ec), 0u);     // M | ~(sum | P | Eqc)
    const u32 mh = bitop3<0xB0>(P, sum, ec);                          // P & ((sum ^ P) | Eqc)
    const u32 phs = __builtin_amdgcn_alignbit(ph, inPw, 31);          // (Ph << 1) | (inPw >> 31)
    const u32 mhs = (mh << 1) | MHin;
    P = bitop3<0xF1>(mhs, xv, phs);                                   // Mhs | ~(Xv | Phs)
    M = phs & xv;
    oPw = ph; oMw = mh;
    if (STORE) *st = make_uint2(P, Mb);
}
// window_walk_tile<true> with the deletion run behind a branch the wave skips when no lane starts one at this column
__device__ __forceinline__ void window_walk_tile_lean(const u64 (&tP)[8], const u64 (&tM)[8], const u64 (&tE)[8],
                                                      bool& inr, int& vw, int& hw, int& wscore) {
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const bool mine = inr && (hw & 7) == j;
        const int bit = vw & 63;
        const u32 pb = (u32)(tP[j] >> bit) & 1u;
        int r = 0, b1 = bit;
        if (mine && pb != 0) {
            r = min(__clzll((long long)~(tP[j] << (63 - bit))), bit + 1);
            b1 = (bit - r) & 63;
        }
        const bool up = r == bit + 1;                                 // the run left the block row: window done
        const bool go = mine && !up;
        const u32 mb = (u32)(tM[j] >> b1) & 1u, eq = (u32)(tE[j] >> b1) & 1u;
        wscore += r + (go ? (int)(mb | (eq ^ 1u)) : 0);
        vw -= r + ((go && !mb) ? 1 : 0);
        hw -= go ? 1 : 0;
        inr = inr && !(mine && up) && vw >= 64;
        asm("" : "+v"(vw), "+v"(hw));                                 // (see walk_tile_lean)
    }
}

__global__ __launch_bounds__(256) void k_windowed_quad(WindowArgs A) {
    if (A.prio) __builtin_amdgcn_s_setprio(3);          // few waves, each a serial chain: first in line at the SIMD's issue arbiter
    const int wv = QE_GROUP_INDEX(), lane = threadIdx.x & 63, j = lane & 3;
    const int t = wv * 16 + (lane >> 2);
    if (wv * 16 >= A.T.ntasks) return;
    uint2* const hist = (uint2*)((char*)qe_dyn_lds + (size_t)QE_WAVE_IN_BLOCK() * QE_WQ_LDS_PER_WAVE);   // [QE_WQ_SLOTS][64]
    uint2* const hw_ = hist + lane;                                   // this lane's column of the slots
    const uint2* const r2 = hist + ((lane & ~3) | 2);                 // lane 2's / lane 3's of this quad
    const uint2* const r3 = hist + ((lane & ~3) | 3);
    const int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    const bool valid = pair >= 0;
    int m = 0, n = 0, p0 = 0, t0 = 0;
    const u64* pp = A.P.pl_p; const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair]; tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const bool ok = valid && (fl & (FLAG_HAS_N | FLAG_NONCANON)) == 0;
    const bool sse = A.sse != 0;
    const bool is0 = j == 0;
    int pos_v = m - 1, pos_h = n - 1;
    int score = 0, hew = 0;
    u32 steps = 0;
    while (__any(ok && pos_v >= 127 && pos_h >= 127)) {
        const bool on = ok && pos_v >= 127 && pos_h >= 127;
        const int v0 = on ? pos_v - 127 : 0, h0 = on ? pos_h - 127 : 0;
        // pattern: this lane's 32 rows; rows 64 .. 127 (the traceback's) for the walk's Eq words
        // every load of the window first, then the funnel shifts (branch-free: a conditional second load per call made the
        // three groups wait for each other -- three memory round trips per window where one does)
        u64 pa = 0, pb = 0, a1 = 0, b1 = 0;
        u64 SA[3] = {0, 0, 0}, SB[3] = {0, 0, 0};
        if (on) {
            const int bpj = p0 + v0 + 32 * j, bp1 = p0 + v0 + 64;
            const u64* qj = pp + 3 * (int64_t)(bpj >> 6);
            const u64* q1 = pp + 3 * (int64_t)(bp1 >> 6);
            // text: 192 bases from t0 + h0 - j, so that bit s of these words is column s - j (what this lane does at step s)
            const int ts = t0 + h0 - j, tsc = max(ts, 0), lsh = tsc - ts;
            const u64* q = tp + 3 * (int64_t)(tsc >> 6);
            const u64 ja0 = qj[0], jb0 = qj[1], ja1 = qj[3], jb1 = qj[4];
            const u64 ra0 = q1[0], rb0 = q1[1], ra1 = q1[3], rb1 = q1[4];
            const u64 x0 = q[0], x1 = q[3], x2 = q[6], x3 = q[9], y0 = q[1], y1 = q[4], y2 = q[7], y3 = q[10];
            const int shj = bpj & 63, sh1 = bp1 & 63, sh = tsc & 63;
            pa = (ja0 >> shj) | ((ja1 << 1) << (63 - shj)); pb = (jb0 >> shj) | ((jb1 << 1) << (63 - shj));
            a1 = (ra0 >> sh1) | ((ra1 << 1) << (63 - sh1)); b1 = (rb0 >> sh1) | ((rb1 << 1) << (63 - sh1));
            SA[0] = (x0 >> sh) | ((x1 << 1) << (63 - sh)); SA[1] = (x1 >> sh) | ((x2 << 1) << (63 - sh)); SA[2] = (x2 >> sh) | ((x3 << 1) << (63 - sh));
            SB[0] = (y0 >> sh) | ((y1 << 1) << (63 - sh)); SB[1] = (y1 >> sh) | ((y2 << 1) << (63 - sh)); SB[2] = (y2 >> sh) | ((y3 << 1) << (63 - sh));
            if (lsh) {                                                // the window starts at the text's first bases: j - ts bits of nothing first
                SA[2] = (SA[2] << lsh) | (SA[1] >> (64 - lsh)); SA[1] = (SA[1] << lsh) | (SA[0] >> (64 - lsh)); SA[0] <<= lsh;
                SB[2] = (SB[2] << lsh) | (SB[1] >> (64 - lsh)); SB[1] = (SB[1] << lsh) | (SB[0] >> (64 - lsh)); SB[0] <<= lsh;
            }
        }
        const u32 a = lo32(pa), b = lo32(pb);
        const u32 tw0[5] = {lo32(SA[0]), hi32(SA[0]), lo32(SA[1]), hi32(SA[1]), lo32(SA[2])};
        const u32 tw1[5] = {lo32(SB[0]), hi32(SB[0]), lo32(SB[1]), hi32(SB[1]), lo32(SB[2])};
        // columns 64 .. 127 un-skewed, for the walk
        const u64 X0 = j ? ((SA[1] >> j) | (SA[2] << (64 - j))) : SA[1];
        const u64 X1 = j ? ((SB[1] >> j) | (SB[2] << (64 - j))) : SB[1];
        // the window's boundaries (bpm_windowed.c:226-230, 260; SSE: 348, 393, 424)
        const u64 ph_first = (v0 == 0) ? QE_ONES : 0;
        const u32 pinit = (h0 == 0) ? ~0u : 0u;
        // lane 0's carry-in words (bit 31 = the boundary's PHin): the real first row, or the SSE kernel's pattern -- column 0
        // Ph_first, column 1 one, then one at even columns (and at the column one past the window)
        const u32 Bf = (u32)(ph_first & 1) << 31;
        const u32 BWe = sse ? 0x80000000u : Bf, BWo = sse ? 0u : Bf;
        const u32 w0 = is0 ? 0u : 1u;                               // width of the MHin extract: lane 0's is always 0
        // the extra column's Eq (lanes 0 and 1, SSE): text[tlen] reads as N, which matches nothing here (A.7(4))
        const u32 eqx = (on && pos_h + 1 < n) ? ~0u : 0u;
        u32 P = pinit, M = 0, oPw = 0, oMw = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the walk of the window before has read its slots
#pragma unroll
        for (int s = 0; s < 129; ++s) {
            u32 inPw = quad_ror1(oPw);
            const u32 inMw = quad_ror1(oMw);
            inPw = is0 ? ((s == 0) ? Bf : ((s & 1) && s != 1 ? BWo : BWe)) : inPw;
            const u32 MHin = __builtin_amdgcn_ubfe(inMw, 31, w0);
            const u32 em = (s == 128 && is0) ? eqx : ~0u;
            if (s < 3) {
                if (j <= s) quad_col_w<false>(tw0[s >> 5], tw1[s >> 5], s & 31, em, a, b, P, M, inPw, MHin, oPw, oMw, hw_);
            } else if (s < 66) {
                quad_col_w<false>(tw0[s >> 5], tw1[s >> 5], s & 31, em, a, b, P, M, inPw, MHin, oPw, oMw, hw_);
            } else {
                quad_col_w<true>(tw0[s >> 5], tw1[s >> 5], s & 31, em, a, b, P, M, inPw, MHin, oPw, oMw, hw_ + (s - 66) * 64);
            }
        }
        {   // step 129: lane 1 runs the extra column (its carries reach lane 2 only under SSE semantics), lane 3 column 126
            const u32 inPw = quad_ror1(oPw), MHin = __builtin_amdgcn_ubfe(quad_ror1(oMw), 31, 1u);
            const u32 kP = oPw, kM = oMw;
            if (j & 1) quad_col_w<true>(tw0[4], tw1[4], 1, (j == 1) ? eqx : ~0u, a, b, P, M, inPw, MHin, oPw, oMw, hw_ + 63 * 64);
            if (j == 1 && !sse) { oPw = kP; oMw = kM; }
        }
        {   // step 130: lane 2's column 127 (text bit 127 + 2)
            const u32 inPw = quad_ror1(oPw), MHin = __builtin_amdgcn_ubfe(quad_ror1(oMw), 31, 1u);
            if (j == 2) quad_col_w<true>(tw0[4], tw1[4], 1, ~0u, a, b, P, M, inPw, MHin, oPw, oMw, hw_ + 63 * 64);
        }
        {   // step 131: lane 3's column 127 (text bit 127 + 3)
            const u32 inPw = quad_ror1(oPw), MHin = __builtin_amdgcn_ubfe(quad_ror1(oMw), 31, 1u);
            if (j == 3) quad_col_w<true>(tw0[4], tw1[4], 2, ~0u, a, b, P, M, inPw, MHin, oPw, oMw, hw_ + 64 * 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // lanes 2 / 3 wrote, every lane of the quad reads
        if (on) steps += 256u;
        // in-window traceback over the stored columns: window rows / columns 64 .. 127 (cf. k_windowed's on-chip path)
        int vw = 127, hw = 127, wscore = 0;
        bool inr = on;
        const u32 alo = lo32(a1), ahi = hi32(a1), blo = lo32(b1), bhi = hi32(b1);
#pragma unroll 1
        for (int q = 7; q >= 0 && __any(inr); --q) {
            const u32 t0s = (u32)(X0 >> (8 * q)), t1s = (u32)(X1 >> (8 * q));
            u64 tP[8], tM[8], tE[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint2 l2 = r2[(8 * q + c) * 64], l3 = r3[(8 * q + c + 1) * 64];
                const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, c, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, c, 1);
                tE[c] = mk64(bitop3<0x90>(~(alo ^ m0), blo, m1), bitop3<0x90>(~(ahi ^ m0), bhi, m1));
                tP[c] = mk64(l2.x, l3.x);
                tM[c] = mk64(l2.y, l3.y);
            }
            window_walk_tile_lean(tP, tM, tE, inr, vw, hw, wscore);
        }
        if (on) {
            if (wscore > 64 * A.hew_threshold / 100) ++hew;         // (W - O) * 64 * hew_threshold / 100, bpm_windowed.c:556-558
            score += wscore;
            pos_v = v0 + vw; pos_h = h0 + hw;
        }
    }
    if (is0 && t < A.T.ntasks) {
        const int64_t nt = A.T.ntasks;
        A.state[t] = pos_v; A.state[nt + t] = pos_h; A.state[2 * nt + t] = score; A.state[3 * nt + t] = hew;
        A.state[4 * nt + t] = (int32_t)steps;
    }
}

// ===========================================================================
// WindowEd for ANY window shape of up to 15 blocks, SIXTEEN LANES PER ALIGNMENT (k_windowed_sys): the cooperative form of
// k_windowed_cp for launches of few waves -- QuickEd's stage 2 (WindowEd(9, 1), forward and reversed) on the pairs a run
// left, the WINDOWED algorithm on a few hundred pairs (bpm_windowed.c:202-280, 504-628; score only).
// Fill: lane i of a group is block row i of the window, the rows a systolic array skewed by one column per row (row i works
// on column s - i at step s, carries by v_mov_dpp row_ror:1, the lane above row 0 holds the window's top boundary carry):
// 64 W + W - 1 steps of one block step instead of W x 64 W.  A lane's text bit at step s is bit s mod 64 of a per-lane word
// funnelled from the current and the previous chunk's plane words, so that every lane reads the same literal position.
// What the in-window traceback needs is left exactly as k_windowed_cp leaves it ({Pv, Mv} before every 8th column of every
// block it may visit, the carry-in words per (chunk, block), in the group's workspace).
// Traceback: a round rebuilds SIXTEEN 8-column tiles at once -- the 8 column tiles to the left of the walk's position, for
// each the block row the path's diagonal predicts and the one above -- and the walk then visits them in path order, its
// state (v, h, the window's score) broadcast from the tile's owner after every tile (cf. k_traceback_sys).
// Same windows, same anchors, same scores and HEW counts as k_windowed_cp; tasks with N or non-canonical symbols are
// flagged (o_abort) and left to it.
// ===========================================================================
__global__ __launch_bounds__(256) void k_windowed_sys(WindowArgs A) {
    if (A.prio) __builtin_amdgcn_s_setprio(3);          // few waves, each a serial chain: first in line at the SIMD's issue arbiter
    const int wv = QE_GROUP_INDEX(), lane = threadIdx.x & 63, j = lane & 15, gl = lane & ~15;
    const int t = wv * 4 + (lane >> 4);
    if (wv * 4 >= A.T.ntasks) return;
    const int pair = (t < A.T.ntasks) ? A.T.pair[t] : -1;
    const bool valid = pair >= 0;
    int m = 0, n = 0, p0 = 0, t0 = 0;
    const u64* pp = A.P.pl_p; const u64* tp = A.P.pl_t;
    u32 fl = 0;
    if (valid) {
        m = A.T.m[t]; n = A.T.n[t]; p0 = A.T.p0[t]; t0 = A.T.t0[t];
        pp = A.P.pl_p + A.P.pl_p_off[pair]; tp = A.P.pl_t + A.P.pl_t_off[pair];
        fl = A.P.flags[pair];
    }
    const int W = A.W, O = A.O;
    const bool ok = valid && (fl & (FLAG_HAS_N | FLAG_NONCANON)) == 0;
    if (valid && j == 0) A.o_abort[t] = ok ? 0 : 1;
    if (!__any(ok)) return;
    const int g = ok ? (t >> 6) : 0, col = t & 63;
    uint8_t* wsb = A.ws + A.g_ws_off[g];
    uint4* const cpb = (uint4*)(wsb + (int64_t)2 * W * 64 * 8) + col;      // [8 x chunks][W][64]: {Pv, Mv} before column 8 q
    uint4* const hwb = cpb + (int64_t)8 * W * W * 64;                      // [chunks][W][64]: carry-in words
    int pos_v = m - 1, pos_h = n - 1;
    int score = 0, hew = 0;
    u32 steps = 0;
    while (__any(ok && pos_v >= 0 && pos_h >= 0)) {
        const bool on = ok && pos_v >= 0 && pos_h >= 0;
        const int v_fi = pos_v, h_fi = pos_h;
        const int v0 = max(v_fi - 64 * W + 1, 0), h0 = max(h_fi - 64 * W + 1, 0);
        const int steps_v = on ? (v_fi - v0) / 64 + 1 : 0;
        const int ncols_total = on ? h_fi - h0 + 1 : 0;
        const u32 ph_first = (v0 == 0) ? 1u : 0u;
        const u32 pinit = (h0 == 0) ? ~0u : 0u;
        const int v_ov = max(v_fi - 64 * (W - O) + 1, 0), h_ov = max(h_fi - 64 * (W - O) + 1, 0);
        const int blk_min = (v_ov - v0) >> 6;
        // ---- fill: lane i = block row i
        const int i = j;
        const bool rowon = on && i < steps_v;
        u64 pa = 0, pb = 0, pn;
        if (rowon) load_planes(pp, p0 + v0 + 64 * i, pa, pb, pn);
        const u32 alo = lo32(pa), ahi = hi32(pa), blo = lo32(pb), bhi = hi32(pb);
        u32 Plo = pinit, Phi = pinit, Mlo = 0, Mhi = 0;
        u32 oP = 0, oM = 0;
        if (i == 15) { oP = ph_first; oM = 0u; }                   // the lane above row 0: the window's top boundary carry
        const u32 len = rowon ? (u32)ncols_total : 0u;
        const bool rkeep = rowon && i >= blk_min;
        const int nJ = __builtin_amdgcn_readfirstlane(wave_max((ncols_total + steps_v - 1 + 63) >> 6));       // 64-step blocks
        u64 Tp0 = 0, Tp1 = 0;                                      // the previous chunk's text planes
        u64 gP = 0, gM = 0;
#pragma unroll 1
        for (int J = 0; J < nJ; ++J) {
            u64 Tc0 = 0, Tc1 = 0;
            if (on && 64 * J < ncols_total) load_planes_ab(tp, t0 + h0 + 64 * J, Tc0, Tc1);
            // bit s mod 64 of these is the text's column 64 J + (s mod 64) - i: this lane's column at step s
            const u64 R0 = i ? ((Tc0 << i) | (Tp0 >> (64 - i))) : Tc0, R1 = i ? ((Tc1 << i) | (Tp1 >> (64 - i))) : Tc1;
            Tp0 = Tc0; Tp1 = Tc1;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const u32 w0 = half ? hi32(R0) : lo32(R0), w1 = half ? hi32(R1) : lo32(R1);
#pragma unroll
                for (int sb = 0; sb < 32; ++sb) {
                    const int s = 64 * J + 32 * half + sb;
                    const u32 inP = grp_ror1<4>(oP), inM = grp_ror1<4>(oM);
                    const u32 c = (u32)(s - i);
                    if (c < len) {
                        // {Pv, Mv} before every 8th column and the chunk's carry-in words, where the traceback may come
                        const bool ckeep = rkeep && (int)(c | 63u) >= h_ov - h0;
                        if ((c & 7u) == 0 && ckeep) cpb[((int64_t)(c >> 3) * W + i) * 64] = make_uint4(Plo, Phi, Mlo, Mhi);
                        const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)w0, sb, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)w1, sb, 1);
                        const u32 elo = bitop3<0x90>(~(alo ^ m0), blo, m1), ehi = bitop3<0x90>(~(ahi ^ m0), bhi, m1);
                        u32 phhi, mhhi;
                        block_step_core(elo, ehi, Plo, Phi, Mlo, Mhi, inP, inM, phhi, mhhi);
                        oP = phhi >> 31; oM = mhhi >> 31;
                        gP = shl1_add_u64(gP, (u64)inP); gM = shl1_add_u64(gM, (u64)inM);
                        if ((c & 63u) == 63u || c + 1 == len) {
                            // the chunk's carry-in words, bit c' = column c' of the chunk
                            const int nc = (int)(c & 63u) + 1;
                            const u64 xP = __builtin_bitreverse64(gP << (64 - nc)), xM = __builtin_bitreverse64(gM << (64 - nc));
                            if (ckeep) hwb[((int64_t)(c >> 6) * W + i) * 64] = make_uint4(lo32(xP), hi32(xP), lo32(xM), hi32(xM));
                            gP = 0; gM = 0;
                        }
                    }
                }
            }
        }
        if (on && j == 0) steps += (u32)steps_v * (u32)ncols_total;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the rows' lanes stored, the tiles' lanes load (one wave, program order)
        // ---- in-window traceback (bpm_windowed.c:504-561) over recomputed tiles, sixteen at a time
        int v = pos_v, h = pos_h, wscore = 0;
        while (__any(on && v >= v_ov && h >= h_ov)) {
            const bool live = on && v >= v_ov && h >= h_ov;
            const int x = j >> 1;
            const int q = ((h - h0) >> 3) - x;
            const int v_in = (x == 0) ? v : v - ((h - h0) - (8 * q + 7));            // where the diagonal meets the tile's right edge
            const int Rb = ((max(v_in, v0) - v0) >> 6) - (j & 1);
            const bool act = live && q >= 0 && Rb >= 0;
            u64 tP[8], tM[8], tE[8];
            {
                u64 qa = 0, qb = 0, qn, X0 = 0, X1 = 0, hP = 0, hM = 0;
                u32 cPlo = 0, cPhi = 0, cMlo = 0, cMhi = 0;
                if (act) {
                    const int jc = q >> 3;
                    load_planes(pp, p0 + v0 + 64 * Rb, qa, qb, qn);
                    load_planes_ab(tp, t0 + h0 + 64 * jc, X0, X1);
                    const uint4 w0 = hwb[((int64_t)jc * W + Rb) * 64];
                    const uint4 c0 = cpb[((int64_t)q * W + Rb) * 64];
                    hP = mk64(w0.x, w0.y); hM = mk64(w0.z, w0.w);
                    cPlo = c0.x; cPhi = c0.y; cMlo = c0.z; cMhi = c0.w;
                }
                const int sh = 8 * (max(q, 0) & 7);
                const u32 xlo = lo32(qa), xhi = hi32(qa), ylo = lo32(qb), yhi = hi32(qb);
                const u32 t0s = (u32)(X0 >> sh), t1s = (u32)(X1 >> sh);
                const u32 hp = (u32)(hP >> sh), hm = (u32)(hM >> sh);
                u32 aP = 0, aM = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)t0s, c, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)t1s, c, 1);
                    const u32 elo = bitop3<0x90>(~(xlo ^ m0), ylo, m1), ehi = bitop3<0x90>(~(xhi ^ m0), yhi, m1);
                    tE[c] = mk64(elo, ehi);
                    tM[c] = mk64(cMlo, cMhi);
                    block_step_fused(elo, ehi, cPlo, cPhi, cMlo, cMhi, __builtin_amdgcn_ubfe(hp, c, 1), __builtin_amdgcn_ubfe(hm, c, 1), aP, aM);
                    tP[c] = mk64(cPlo, cPhi);
                }
            }
            u64 tX[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) tX[c] = tP[c] | tM[c] | ~tE[c];
            while (true) {
                const bool mine = act && v >= v_ov && h >= h_ov && ((h - h0) >> 3) == q && ((v - v0) >> 6) == Rb;
                const u64 bal = __ballot(mine);
                const u32 grp = (u32)(bal >> gl) & 0xffffu;
                if (!__any(grp != 0)) break;
                window_walk_tile_g_fast(tP, tM, tE, tX, mine, Rb, v0, h0, v_ov, h_ov, v, h, wscore);
                if (grp != 0) {
                    const int own = gl | (__ffs((int)grp) - 1);
                    v = __shfl(v, own); h = __shfl(h, own); wscore = __shfl(wscore, own);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // ... and the next window's stores come after these loads
        if (on) {
            if (wscore > (W - O) * 64 * A.hew_threshold / 100) ++hew;
            score += wscore;
            pos_v = v; pos_h = h;
        }
    }
    if (ok && j == 0) {
        if (pos_h >= 0) score += pos_h + 1;
        if (pos_v >= 0) score += pos_v + 1;
        A.o_score[t] = score;
        A.o_hew[t] = hew;
        A.o_steps[t] = steps;
    }
}

// ===========================================================================
// QuickEd without a host round trip after stage 1 (the common case: no pair leaves stage 1).  k_stage1_decide applies
// the stage-1 rule of run_quicked (quicked.c:201-202) to k_windowed's outputs; k_apply_cutoffs hands the bounds to the
// align step's task list as cutoffs and takes the pairs that need stages 2 / 3 (or more room than was planned) out
// of it -- the host aligns those afterwards through the classic flow.
// ===========================================================================
__global__ __launch_bounds__(256) void k_stage1_decide(Stage1Args A) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.nt) return;
    int cut = 0, skip = 1;
    u32 steps = 0;
    if (A.pair[t] >= 0) {
        const u32 mx = (u32)max(A.m[t], A.n[t]);
        const bool stage2 = (u64)(u32)A.hew[t] * 64u > (u64)(mx * A.hew_percentage / 100u);     // unsigned arithmetic as in quicked.c:201
        cut = A.score[t];
        skip = (stage2 ? 1 : 0) | ((cut > A.est[t]) ? 2 : 0);
        if (A.flags != nullptr && (A.flags[A.pair[t]] & FLAG_NONCANON)) skip |= 4;
        steps = A.steps[t];
    }
    A.o_cut[t] = cut; A.o_skip[t] = skip; A.o_steps[t] = steps;
}
__global__ __launch_bounds__(256) void k_apply_cutoffs(int nt, int32_t* cutoff, int32_t* pair, const int32_t* cut, const int32_t* skip) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    cutoff[t] = cut[t];
    if (skip[t] != 0) pair[t] = -1;
}

// ===========================================================================
// CIGAR formatting (cigar_sprint, cigar.c:453-488): runs are stored back to
// front, the string goes front to back: "<len><op>" per run.
// ===========================================================================
__device__ __forceinline__ int dec_digits(u32 x) {
    int d = 1;
    while (x >= 10) { x /= 10; ++d; }
    return d;
}



// ===========================================================================
// Hirschberg midpoint join.  D[i][column] for pattern prefix length i follows
// from the stopped band: scores[r] is D at the bottom row of block r (A.8) and
// rows inside a block from the Pv/Mv vertical deltas.  The split row is the
// FIRST i minimising D_fwd[i] + D_rev[m - i] over the rows both bands cover;
// the two summands are the children's exact distances = their cutoffs.
// ===========================================================================
struct ColDist {
    const u64* Pv; const u64* Mv; const int32_t* S;
    int first, last, posv, maxrow, m, nw, column;
    int blk = -1, bottom = 0;
    u64 P = 0, M = 0;
    int64_t stride = 64;
    __device__ __forceinline__ void init(const BandState& B, int t, int m_, int column_) {
        if (B.G <= 1) {
            const int g = t >> 6, lane = t & 63;
            const GroupWs W = group_ws(const_cast<uint8_t*>(B.ws), B.g_ws_off[g], B.g_nslots[g], B.g_nrows[g], B.g_nch[g]);
            Pv = W.Pv + 64 + lane; Mv = W.Mv + 64 + lane; S = W.S + lane;
            stride = 64;
        } else {
            // k_banded_coop layout: Pv[(ns+1)][NA] | Mv | S[2][nrows][NA]; the row values after the last
            // processed chunk are in parity (chunks - 1) & 1
            const int NA = 64 / B.G, w = t / NA, q = t - w * NA;
            const uint8_t* base = B.ws + B.g_ws_off[w];
            const int ns = B.g_nslots[w], nr = B.g_nrows[w];
            Pv = (const u64*)base + NA + q;                      base += (int64_t)(ns + 1) * NA * 8;
            Mv = (const u64*)base + NA + q;                      base += (int64_t)(ns + 1) * NA * 8;
            const int chunks = (column_ >> 6) + ((column_ & 63) ? 1 : 0);
            S = (const int32_t*)base + q + (((chunks - 1) & 1) ? (int64_t)nr * NA : 0);
            stride = NA;
        }
        first = B.first[t]; last = B.last[t]; posv = B.posv[t]; maxrow = B.maxrow[t];
        m = m_; nw = (m_ + 63) >> 6; column = column_;
    }
    __device__ __forceinline__ bool row_ok(int r) const {
        const int s = r - posv;
        return s >= first && s <= last && r >= 0 && r < nw && r <= maxrow;
    }
    // -1 when prefix length i is not covered by the band
    __device__ __forceinline__ int get(int i) {
        if (i == 0) return row_ok(0) ? column : -1;         // exact only on the real first row (PHin = 1 there)
        const int r = (i - 1) >> 6;
        if (!row_ok(r)) return -1;
        if (r != blk) {
            blk = r;
            const int s = r - posv;
            P = Pv[(int64_t)s * stride]; M = Mv[(int64_t)s * stride];
            const int sc = S[(int64_t)r * stride];
            const int full = 64 * (r + 1);
            if (full <= m) bottom = sc;
            else {                                          // partial last block: scores tracks row m (A.8)
                bottom = sc - (full - m);
                const u64 mask = (((u64)1) << (m - 64 * r)) - 1;
                P &= mask; M &= mask;
            }
        }
        const int k = i - 64 * r;                           // 1..64
        if (k >= 64) return bottom;
        return bottom - (__popcll(P >> k) - __popcll(M >> k));
    }
};

// One wave per node: lane l scans the prefix lengths [l c, (l + 1) c) and the wave keeps the lexicographic minimum of
// (sum, i) -- the FIRST minimising i, as a single scan would
__global__ __launch_bounds__(64) void k_join(JoinArgs A) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= A.nnodes) return;
    const int m = A.m[j];
    ColDist F, R;
    F.init((A.F.G > 1 && A.F.abort[j]) ? A.Ffb : A.F, j, m, A.n1[j]);
    R.init((A.R.G > 1 && A.R.abort[j]) ? A.Rfb : A.R, j, m, A.n2[j]);
    const int per = (m + 64) / 64;                              // ceil((m + 1) / 64)
    const int lo = lane * per, hi = min(lo + per - 1, m);
    int best = 0x7fffffff, best_i = 0x7fffffff, sl = 0, sr = 0;
    for (int i = lo; i <= hi; ++i) {
        const int df = F.get(i);
        if (df < 0) continue;
        const int dr = R.get(m - i);
        if (dr < 0) continue;
        const int s = df + dr;
        if (s < best) { best = s; best_i = i; sl = df; sr = dr; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ob = __shfl_xor(best, o), oi = __shfl_xor(best_i, o), ol = __shfl_xor(sl, o), orr = __shfl_xor(sr, o);
        if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; sl = ol; sr = orr; }
    }
    if (lane == 0) {
        const bool ok = best != 0x7fffffff;
        A.o_best[j] = ok ? best_i : -1;
        A.o_score_l[j] = ok ? sl : 0;
        A.o_score_r[j] = ok ? sr : 0;
        A.o_ok[j] = ok ? 1 : 0;
    }
}

// ===========================================================================
// Segment formatter: one lane per pair walks its segments front to back (a
// leaf's runs were emitted back to front), merging equal neighbours.
// ===========================================================================
// where the runs of leaf task t are (see RunSink): [idx][lane] rows of its 64-task group, or a stretch of its own
struct RunView {
    const u32* base; int64_t stride;
    __device__ __forceinline__ u32 at(int64_t k) const { return base[k * stride]; }
};
__device__ __forceinline__ RunView run_view(const SegFormatArgs& A, int t) {
    const int g = t >> 6, lane = t & 63;
    RunView v;
    if (A.runs_by_task) { v.base = A.runs + A.g_runs_off[g] + (int64_t)lane * A.g_runs_cap[g]; v.stride = 1; }
    else { v.base = A.runs + A.g_runs_off[g] + lane; v.stride = 64; }
    return v;
}

struct RunMerger {
    int style = 0;                  // SegFormatArgs::style
    int op = -1, len = 0, total = 0, edits = 0, nops = 0;
    char* out = nullptr;
    u64 buf = 0; int nb = 0;        // up to 7 characters waiting for an 8-byte store
    __device__ __forceinline__ void put(u64 tok, int n) {            // n <= 8 characters, first one in the low byte
        buf |= tok << (8 * nb);                                      // nb <= 7
        if (nb + n >= 8) {
            __builtin_memcpy(out, &buf, 8);                          // one (unaligned) 8-byte store
            out += 8;
            buf = nb ? (tok >> (8 * (8 - nb))) : 0;
            nb += n - 8;
        } else nb += n;
    }
    template <bool WRITE> __device__ __forceinline__ void emit() {
        if (len <= 0) return;
        const int d = dec_digits((u32)len);
        if (WRITE) {
            const u32 letters = (style == 1) ? 0x4449583Du : 0x4449584Du;     // "=XID" : "MXID"
            const u64 opc = (u64)((letters >> (8 * (op & 3))) & 0xFFu);
            u32 x = (u32)len;
            if (d <= 7) {
                u64 tok = opc;
                for (int k = 0; k < d; ++k) { tok = (tok << 8) | (u64)('0' + x % 10); x /= 10; }
                put(tok, d + 1);
            } else {                                                 // 8..10 digits: two tokens
                u64 lo = opc; int k = 0;
                for (; k < 3; ++k) { lo = (lo << 8) | (u64)('0' + x % 10); x /= 10; }     // last 3 digits + op
                u64 hi = 0;
                for (; k < d; ++k) { hi = (hi << 8) | (u64)('0' + x % 10); x /= 10; }
                put(hi, d - 3);
                put(lo, 4);
            }
        }
        total += d + 1;
    }
    template <bool WRITE> __device__ __forceinline__ void push(int o, int n) {
        if (n <= 0) return;
        nops += n;
        if (o != (int)OP_M) edits += n;
        if (style == 2 && o == (int)OP_X) {                           // SAM "M" covers matches and mismatches ...
            if (nops == n && op < 0) {                                // ... but the reference reads the alignment's very first
                op = (int)OP_X; len = 1;                              // operation before its mapping step (cigar.c:211 vs 217):
                if (--n == 0) return;                                 // a leading X stays "1X"
            }
            o = (int)OP_M;
        }
        if (o == op) len += n;
        else { emit<WRITE>(); op = o; len = n; }
    }
    __device__ __forceinline__ void finish() {                       // the waiting characters and the terminator
        for (int k = 0; k < nb; ++k) out[k] = (char)(buf >> (8 * k));
        out[nb] = '\0';
    }
};

template <bool WRITE>
__global__ __launch_bounds__(64) void k_format_segs(SegFormatArgs A) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.npairs) return;
    RunMerger Mg;
    Mg.style = A.style;
    bool bad = false;
    if (WRITE) Mg.out = A.pool + A.str_off[i];
    for (int64_t sidx = A.seg_off[i]; sidx < A.seg_off[i + 1]; ++sidx) {
        if (A.seg_kind[sidx] == 1) { Mg.push<WRITE>(A.seg_a[sidx], A.seg_b[sidx]); continue; }
        const int t = A.seg_a[sidx];
        const RunView runs = run_view(A, t);
        if (A.nruns[t] < 0) bad = true;                                   // run buffer overflow (see k_traceback)
        for (int k = A.nruns[t] - 1; k >= 0 && !bad; --k) {
            const u32 r = runs.at(k);
            Mg.push<WRITE>((int)(r & 3), (int)(r >> 2));
        }
    }
    if (bad) {
        if (WRITE) A.pool[A.str_off[i]] = '\0';
        else { A.o_len[i] = 0; A.o_edits[i] = -1; A.o_nops[i] = 0; }
        return;
    }
    Mg.emit<WRITE>();
    if (WRITE) Mg.finish();
    else { A.o_len[i] = Mg.total; A.o_edits[i] = Mg.edits; A.o_nops[i] = Mg.nops; }
}
template __global__ void k_format_segs<false>(SegFormatArgs);
template __global__ void k_format_segs<true>(SegFormatArgs);

// ---------------------------------------------------------------------------
// The same formatter with one WAVE per alignment (strings of more than a few runs: a 100 kb pair has ~30 k runs, and one
// lane walking them twice is a 15 ms launch).  The alignment's runs in string order are one sequence (segment after
// segment, a leaf's runs back to front); the wave takes it 64 consecutive runs at a time -- lane l run 64 j + l: with the
// traceback's by-task layout one 256-byte row per step, every line of the run buffer read once (round 5's form gave every
// lane a slice of its own to walk: 64 streams per wave, 20-27 x the run bytes fetched from HBM once the waves of a chip
// together outgrew its L2) -- and merges equal neighbours with a segmented scan: inside a leaf neighbouring runs differ, so
// groups of more than one run only form across segment borders, but any grouping is handled.  A group's text is written
// by the lane that holds its last run; a group that is still open at the end of a step is carried into the next.
// Styles 0 and 1 only (style 2 folds X into M, which regroups inside leaves).
// ---------------------------------------------------------------------------
struct SegCursor {
    const SegFormatArgs* A; int64_t s1;          // end of this alignment's segments
    int64_t seg, first;                          // current segment; sequence index of its first run
    int n = 0, kind = 0, la = 0, lb = 0, nr = 0;
    RunView rv;
    __device__ __forceinline__ void load() {
        n = 0;
        if (seg >= s1) return;
        kind = A->seg_kind[seg]; la = A->seg_a[seg];
        if (kind == 1) { lb = A->seg_b[seg]; n = lb > 0 ? 1 : 0; }
        else { nr = max(A->nruns[la], 0); n = nr; rv = run_view(*A, la); }
    }
    // run `idx` of the sequence (idx only ever grows from call to call); false past the end
    __device__ __forceinline__ bool get(int64_t idx, int& op, int& len) {
        while (seg < s1 && idx >= first + n) { first += n; ++seg; load(); }
        if (seg >= s1) return false;
        if (kind == 1) { op = la; len = lb; return true; }
        const u32 r = rv.at(nr - 1 - (idx - first));
        op = (int)(r & 3); len = (int)(r >> 2);
        return true;
    }
};

template <bool WRITE>
__global__ __launch_bounds__(64) void k_format_segs_wave(SegFormatArgs A) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= A.npairs) return;
    const int64_t s0 = A.seg_off[i], s1 = A.seg_off[i + 1];
    int64_t total_runs = 0; bool bad = false;
    for (int64_t sg = s0 + lane; sg < s1; sg += 64) {
        if (A.seg_kind[sg] == 1) total_runs += A.seg_b[sg] > 0 ? 1 : 0;
        else { const int nr = A.nruns[A.seg_a[sg]]; if (nr < 0) bad = true; total_runs += max(nr, 0); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total_runs += __shfl_xor(total_runs, o);
    if (__any(bad)) {
        if (lane == 0) { if (WRITE) A.pool[A.str_off[i]] = '\0'; else { A.o_len[i] = 0; A.o_edits[i] = -1; A.o_nops[i] = 0; } }
        return;
    }
    SegCursor C; C.A = &A; C.s1 = s1; C.seg = s0; C.first = 0; C.load();
    const u32 letters = (A.style == 1) ? 0x4449583Du : 0x4449584Du;      // "=XID" : "MXID"
    char* const out = WRITE ? A.pool + A.str_off[i] : nullptr;
    int64_t written = 0;                 // characters of the groups closed so far (the same in every lane)
    int edits = 0, nops = 0;             // per lane; summed at the end
    int carry_op = -1, carry_len = 0;    // the group that was still open at the end of the previous step (the same in every lane)
    int nop = -1, nlen = 0;              // my run of the NEXT step (one step of look-ahead: lane 63 needs its right neighbour's op)
    bool nvalid = C.get(lane, nop, nlen);
    for (int64_t base = 0; base < total_runs; base += 64) {
        const bool valid = nvalid;
        const int op = valid ? nop : -1;
        int x = valid ? nlen : 0;
        nop = -1; nlen = 0;
        nvalid = (base + 64 + lane < total_runs) && C.get(base + 64 + lane, nop, nlen);
        const int right = __shfl_down(op, 1), first_next = __shfl(nvalid ? nop : -1, 0);
        const int op_next = (lane == 63) ? first_next : right;
        const int left = __shfl_up(op, 1);
        const int op_prev = (lane == 0) ? carry_op : left;
        bool flag = valid && op != op_prev;                              // my run opens a group
        const bool tail = valid && op != op_next;                        // ... closes one
        // segmented inclusive scan of the run lengths: x = length of my group up to and including my run
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            const int f = __shfl_up((int)flag, d);
            if (lane >= d && !flag) { x += y; flag = f != 0; }
        }
        if (valid && !flag) x += carry_len;                              // my group was opened in an earlier step
        // what the step leaves open
        const int last = (int)min((int64_t)63, total_runs - 1 - base);
        const int l_tail = __shfl((int)tail, last), l_x = __shfl(x, last), l_op = __shfl(op, last);
        carry_op = l_op; carry_len = l_tail ? 0 : l_x;
        // the groups that close here: "<length><letter>"
        const int digits = tail ? dec_digits((u32)x) : 0;
        const int chars = tail ? digits + 1 : 0;
        if (tail) { nops += x; if (op != (int)OP_M) edits += x; }
        int off = chars;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(off, d); if (lane >= d) off += y; }
        const int step_chars = __shfl(off, 63);
        if (WRITE && tail) {
            char* q = out + written + (off - chars);
            u32 v = (u32)x;
            for (int k = digits - 1; k >= 0; --k) { q[k] = (char)('0' + v % 10); v /= 10; }
            q[digits] = (char)((letters >> (8 * (op & 3))) & 0xFFu);
        }
        written += step_chars;
    }
    if (!WRITE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { edits += __shfl_xor(edits, o); nops += __shfl_xor(nops, o); }
        if (lane == 0) { A.o_len[i] = (int)written; A.o_edits[i] = edits; A.o_nops[i] = nops; }
        return;
    }
    if (lane == 0) out[written] = '\0';
}
template __global__ void k_format_segs_wave<false>(SegFormatArgs);
template __global__ void k_format_segs_wave<true>(SegFormatArgs);

// ===========================================================================
// Validator (cigar_check_alignment, cigar.c:363-434): one lane per alignment walks its operations front
// to back over the RAW bytes of the pair: M needs equal bytes, X different ones, I consumes text, D
// pattern; both sequences must be consumed exactly.
// ===========================================================================
struct AlignCheck {
    const uint8_t* ap; const uint8_t* at; int m, n, v = 0, h = 0; bool ok = true;
    __device__ __forceinline__ void apply(int op, int cnt) {
        if (!ok || cnt <= 0) return;
        if (op == (int)OP_I) { h += cnt; return; }
        if (op == (int)OP_D) { v += cnt; return; }
        if (v + cnt > m || h + cnt > n) { ok = false; return; }
        int k = 0;
        if (op == (int)OP_M) {
            for (; k + 8 <= cnt; k += 8) {
                u64 x, y; __builtin_memcpy(&x, ap + v + k, 8); __builtin_memcpy(&y, at + h + k, 8);
                if (x != y) { ok = false; return; }
            }
            for (; k < cnt; ++k) if (ap[v + k] != at[h + k]) { ok = false; return; }
        } else {
            for (; k < cnt; ++k) if (ap[v + k] == at[h + k]) { ok = false; return; }
        }
        v += cnt; h += cnt;
    }
    __device__ __forceinline__ int verdict() const { return (ok && v == m && h == n) ? 1 : 0; }
};

__global__ __launch_bounds__(64) void k_check_segs(SegCheckArgs C) {
    const SegFormatArgs& A = C.F;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.npairs) return;
    const int pair = C.root_pair[i];
    AlignCheck K;
    K.ap = C.P.asc_p + C.P.asc_p_off[pair]; K.at = C.P.asc_t + C.P.asc_t_off[pair];
    K.m = C.P.p_len[pair]; K.n = C.P.t_len[pair];
    for (int64_t sidx = A.seg_off[i]; sidx < A.seg_off[i + 1]; ++sidx) {
        if (A.seg_kind[sidx] == 1) { K.apply(A.seg_a[sidx], A.seg_b[sidx]); continue; }
        const int t = A.seg_a[sidx];
        const RunView runs = run_view(A, t);
        if (A.nruns[t] < 0) K.ok = false;
        for (int k = A.nruns[t] - 1; k >= 0; --k) {
            const u32 r = runs.at(k);
            K.apply((int)(r & 3), (int)(r >> 2));
        }
    }
    C.o_ok[i] = K.verdict();
}

// The same walk over caller-supplied CIGAR strings ("<len><op>" with op in MXID, or '=' for M): pair i's string
// starts at pool + off[i] and is NUL-terminated; off[i] < 0 = no string (verdict -1)
__global__ __launch_bounds__(64) void k_check_strings(PairView P, int npairs, const char* pool, const int64_t* off, int32_t* o_ok) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= npairs) return;
    if (off[i] < 0) { o_ok[i] = -1; return; }
    AlignCheck K;
    K.ap = P.asc_p + P.asc_p_off[i]; K.at = P.asc_t + P.asc_t_off[i];
    K.m = P.p_len[i]; K.n = P.t_len[i];
    const char* q = pool + off[i];
    int64_t num = 0; bool have = false;
    for (;; ++q) {
        const char c = *q;
        if (c == 0) break;
        if (c >= '0' && c <= '9') { num = num * 10 + (c - '0'); have = true; if (num > 0x7fffffff) { K.ok = false; break; } continue; }
        int op = -1;
        if (c == 'M' || c == '=') op = (int)OP_M; else if (c == 'X') op = (int)OP_X;
        else if (c == 'I') op = (int)OP_I; else if (c == 'D') op = (int)OP_D;
        if (op < 0 || !have || num == 0) { K.ok = false; break; }
        K.apply(op, (int)num);
        num = 0; have = false;
    }
    if (have) K.ok = false;                 // digits without an operation
    o_ok[i] = K.verdict();
}

// Small in-stream copies as kernels (16 bytes per lane): task lists from pinned host memory (the device reads it over the
// link), results into the batch's result arena.  As hipMemcpyAsync they would go to the DMA engines with a dependency on
// the kernels before them in their stream, and a bulk upload of another thread (quicked_batch_reload) then queues behind
// them on the same engine until that run has executed: 9 ms of transfer took 80 (tools/probe_reload.py).
struct CopyTable {
    enum { MAX = 16 };
    int32_t n;
    uint4* dst[MAX]; const uint4* src[MAX]; int64_t n_u4[MAX];
};
__global__ __launch_bounds__(256) void k_copy_multi(CopyTable T) {
    const int e = blockIdx.y;
    if (e >= T.n) return;
    uint4* __restrict__ d = T.dst[e]; const uint4* __restrict__ q = T.src[e];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.n_u4[e]; i += (int64_t)gridDim.x * blockDim.x) d[i] = q[i];
}

// gathers word spans from anywhere on the device into one buffer: span e = nwords[e] 64-bit words at address src_addr[e] ->
// dst + dst_off[e] (the bit-planes of pairs that live in several batch objects' arenas, qe_driver.hip merged early finish);
// one wave per span
__global__ __launch_bounds__(256) void k_gather_words(int nspans, const int64_t* __restrict__ src_addr, const int64_t* __restrict__ dst_off,
                                                      const int32_t* __restrict__ nwords, u64* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= nspans) return;
    const u64* __restrict__ q = reinterpret_cast<const u64*>(src_addr[e]);
    u64* __restrict__ d = dst + dst_off[e];
    for (int i = lane; i < nwords[e]; i += 64) d[i] = q[i];
}

// copies the first *total bytes (a count only the device knows: the string pool of a run) from src to dst, 16 bytes per lane
__global__ __launch_bounds__(256) void k_copy_total(uint4* __restrict__ dst, const uint4* __restrict__ src, const int64_t* __restrict__ total, int64_t cap_u4) {
    const int64_t n = min((*total + 15) >> 4, cap_u4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// exclusive scan of (len + 1) over tasks -> string offsets; one block, tiles of 1024 consecutive elements (coalesced
// loads and stores), wave scan by shuffles, the 16 wave totals through LDS, a running total carried from tile to tile
__global__ __launch_bounds__(1024) void k_scan_offsets(const int32_t* len, const int32_t* pair, int64_t* off, int64_t* total, int n) {
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int64_t v = (i < n && pair[i] >= 0) ? (int64_t)len[i] + 1 : 0;
        int64_t x = v;                                      // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int64_t t = wsum[w]; if (w < wave) before += t; all += t; }
        if (i < n) off[i] = carry + before + x - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

}  // namespace qe
