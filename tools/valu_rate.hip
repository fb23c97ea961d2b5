// valu_rate.hip -- measures the issue rate of the integer VALU ops the block step is made of
// (one gfx950 chip, every SIMD loaded with 8 waves).  Prints lane-ops/s per instruction kind.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, e = a * 11, f = a * 13, g = a * 17, h = a * 19;
    float fa = a, fb = b, fc = c, fd = d, fe = e, ff = f, fg = g, fh = h;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { a ^= b; b ^= c; c ^= d; d ^= e; e ^= f; f ^= g; g ^= h; h ^= a; }
            if (KIND == 1) {
                a = __builtin_amdgcn_bitop3_b32(a, b, c, 0xB0); b = __builtin_amdgcn_bitop3_b32(b, c, d, 0xF1);
                c = __builtin_amdgcn_bitop3_b32(c, d, e, 0xB0); d = __builtin_amdgcn_bitop3_b32(d, e, f, 0xF1);
                e = __builtin_amdgcn_bitop3_b32(e, f, g, 0xB0); f = __builtin_amdgcn_bitop3_b32(f, g, h, 0xF1);
                g = __builtin_amdgcn_bitop3_b32(g, h, a, 0xB0); h = __builtin_amdgcn_bitop3_b32(h, a, b, 0xF1);
            }
            if (KIND == 2) {
                a = __builtin_amdgcn_alignbit(a, b, 31); b = __builtin_amdgcn_alignbit(b, c, 31);
                c = __builtin_amdgcn_alignbit(c, d, 31); d = __builtin_amdgcn_alignbit(d, e, 31);
                e = __builtin_amdgcn_alignbit(e, f, 31); f = __builtin_amdgcn_alignbit(f, g, 31);
                g = __builtin_amdgcn_alignbit(g, h, 31); h = __builtin_amdgcn_alignbit(h, a, 31);
            }
            if (KIND == 3) { fa = fa * fb + fc; fb = fb * fc + fd; fc = fc * fd + fe; fd = fd * fe + ff; fe = fe * ff + fg; ff = ff * fg + fh; fg = fg * fh + fa; fh = fh * fa + fb; }
            if (KIND == 4) { a += b; b += c; c += d; d += e; e += f; f += g; g += h; h += a; }
            if (KIND == 8) {   // v_or3_b32
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 9) {   // v_lshl_or_b32
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 10) {  // v_add_co_u32 + v_addc_co_u32 (one 64-bit add = 2 instructions, counted as 2 ops)
                uint64_t x = ((uint64_t)b << 32) | a, y = ((uint64_t)d << 32) | c, z = ((uint64_t)f << 32) | e, w = ((uint64_t)h << 32) | g;
                x += y; y += z; z += w; w += x;
                a = (uint32_t)x; b = (uint32_t)(x >> 32); c = (uint32_t)y; d = (uint32_t)(y >> 32); e = (uint32_t)z; f = (uint32_t)(z >> 32); g = (uint32_t)w; h = (uint32_t)(w >> 32);
            }
            if (KIND == 11) {  // v_or_b32 via asm (cannot be simplified away)
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 12) {   // v_bfe_u32
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(a) : "v"(b));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(b) : "v"(c));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(c) : "v"(d));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(d) : "v"(e));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(e) : "v"(f));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(f) : "v"(g));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(g) : "v"(h));
                asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(h) : "v"(a));
            }
            if (KIND == 13) {   // v_bfe_i32
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(a) : "v"(b));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(b) : "v"(c));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(c) : "v"(d));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(d) : "v"(e));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(e) : "v"(f));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(f) : "v"(g));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(g) : "v"(h));
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(h) : "v"(a));
            }
            if (KIND == 14) {   // v_lshlrev_b32
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(a) : "v"(b));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(b) : "v"(c));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(c) : "v"(d));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(d) : "v"(e));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(e) : "v"(f));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(f) : "v"(g));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(g) : "v"(h));
                asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 15) {   // v_ashrrev_i32
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(a) : "v"(b));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(b) : "v"(c));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(c) : "v"(d));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(d) : "v"(e));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(e) : "v"(f));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(f) : "v"(g));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(g) : "v"(h));
                asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 16) {   // v_and_or_b32
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 17) {   // v_xad_u32
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 18) {   // v_bfi_b32
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 19) {   // v_perm_b32
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 20) {   // v_and_b32
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 21) {   // v_cndmask_b32
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 22) {   // v_bfrev_b32
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(a) : "v"(b));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(b) : "v"(c));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(c) : "v"(d));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(d) : "v"(e));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(e) : "v"(f));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(f) : "v"(g));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(g) : "v"(h));
                asm volatile("v_bfrev_b32 %0, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 23) {   // v_add3_u32
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(c), "v"(d));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(c) : "v"(c), "v"(d), "v"(e));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(d), "v"(e), "v"(f));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(e) : "v"(e), "v"(f), "v"(g));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(f) : "v"(f), "v"(g), "v"(h));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(g) : "v"(g), "v"(h), "v"(a));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(h) : "v"(h), "v"(a), "v"(b));
            }
            if (KIND == 24) {   // v_xnor_b32
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 25) {  // v_lshrrev_b64 by a register amount (one 64-bit shift = one op)
                uint64_t x = ((uint64_t)b << 32) | a, y = ((uint64_t)d << 32) | c, z = ((uint64_t)f << 32) | e, w = ((uint64_t)h << 32) | g;
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(x) : "v"(g), "v"(x));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(y) : "v"(g), "v"(y));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(z) : "v"(a), "v"(z));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(w) : "v"(a), "v"(w));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(x) : "v"(c), "v"(x));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(y) : "v"(c), "v"(y));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(z) : "v"(e), "v"(z));
                asm volatile("v_lshrrev_b64 %0, %1, %2" : "=v"(w) : "v"(e), "v"(w));
                a = (uint32_t)x | 1; b = (uint32_t)(x >> 32); c = (uint32_t)y | 1; d = (uint32_t)(y >> 32); e = (uint32_t)z | 1; f = (uint32_t)(z >> 32); g = (uint32_t)w | 1; h = (uint32_t)(w >> 32);
            }
            if (KIND == 26) {  // v_lshlrev_b64
                uint64_t x = ((uint64_t)b << 32) | a, y = ((uint64_t)d << 32) | c, z = ((uint64_t)f << 32) | e, w = ((uint64_t)h << 32) | g;
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(x) : "v"(g), "v"(x));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(y) : "v"(g), "v"(y));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(z) : "v"(a), "v"(z));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(w) : "v"(a), "v"(w));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(x) : "v"(c), "v"(x));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(y) : "v"(c), "v"(y));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(z) : "v"(e), "v"(z));
                asm volatile("v_lshlrev_b64 %0, %1, %2" : "=v"(w) : "v"(e), "v"(w));
                a = (uint32_t)x | 1; b = (uint32_t)(x >> 32); c = (uint32_t)y | 1; d = (uint32_t)(y >> 32); e = (uint32_t)z | 1; f = (uint32_t)(z >> 32); g = (uint32_t)w | 1; h = (uint32_t)(w >> 32);
            }
            if (KIND == 27) {   // v_lshrrev_b32
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a) : "v"(b));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(b) : "v"(c));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(c) : "v"(d));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(d) : "v"(e));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(e) : "v"(f));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(f) : "v"(g));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(g) : "v"(h));
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 28) {   // v_mov_b32
                asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
                asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(c));
                asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(d));
                asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(e));
                asm volatile("v_mov_b32 %0, %1" : "=v"(e) : "v"(f));
                asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(g));
                asm volatile("v_mov_b32 %0, %1" : "=v"(g) : "v"(h));
                asm volatile("v_mov_b32 %0, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 29) {   // v_sub_u32
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 30) {   // v_min_u32
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_min_u32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 31) {   // v_ffbh_u32
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(a) : "v"(b));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(b) : "v"(c));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(c) : "v"(d));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(d) : "v"(e));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(e) : "v"(f));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(f) : "v"(g));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(g) : "v"(h));
                asm volatile("v_ffbh_u32 %0, %1" : "=v"(h) : "v"(a));
            }
            if (KIND == 32) {   // v_bcnt_u32_b32
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(c));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(e));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(e) : "v"(e), "v"(f));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(f) : "v"(f), "v"(g));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(g) : "v"(g), "v"(h));
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(h) : "v"(h), "v"(a));
            }
            if (KIND == 5) {     // 64-bit add as one v_lshl_add_u64 (counted as ONE op per 64-bit add)
                uint64_t x = ((uint64_t)b << 32) | a, y = ((uint64_t)d << 32) | c, z = ((uint64_t)f << 32) | e, w = ((uint64_t)h << 32) | g;
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(x) : "v"(x), "v"(y));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(y) : "v"(y), "v"(z));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(z) : "v"(z), "v"(w));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(w) : "v"(w), "v"(x));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(x) : "v"(x), "v"(y));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(y) : "v"(y), "v"(z));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(z) : "v"(z), "v"(w));
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(w) : "v"(w), "v"(x));
                a = (uint32_t)x; b = (uint32_t)(x >> 32); c = (uint32_t)y; d = (uint32_t)(y >> 32); e = (uint32_t)z; f = (uint32_t)(z >> 32); g = (uint32_t)w; h = (uint32_t)(w >> 32);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(fa + fb + fc + fd + fe + ff + fg + fh);
}

template <int KIND> void run(const char* name, uint32_t* out) {
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 1, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16 * 8;
    printf("%-14s %8.2f Tlane-ops/s  (%.3f ms)\n", name, ops / ms / 1e9, ms);
}

int main() {
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", out); run<1>("v_bitop3_b32", out); run<2>("v_alignbit_b32", out);
    run<3>("v_fma_f32", out); run<4>("v_add_u32", out); run<5>("v_lshl_add_u64", out);
    run<8>("v_or3_b32", out); run<9>("v_lshl_or_b32", out); run<10>("add_co+addc", out); run<11>("v_or_b32", out);
    run<12>("v_bfe_u32", out);
    run<13>("v_bfe_i32", out);
    run<14>("v_lshlrev_b32", out);
    run<15>("v_ashrrev_i32", out);
    run<16>("v_and_or_b32", out);
    run<17>("v_xad_u32", out);
    run<18>("v_bfi_b32", out);
    run<19>("v_perm_b32", out);
    run<20>("v_and_b32", out);
    run<21>("v_cndmask_b32", out);
    run<22>("v_bfrev_b32", out);
    run<23>("v_add3_u32", out);
    run<24>("v_xnor_b32", out);
    run<25>("v_lshrrev_b64", out);
    run<26>("v_lshlrev_b64", out);
    run<27>("v_lshrrev_b32", out);
    run<28>("v_mov_b32", out);
    run<29>("v_sub_u32", out);
    run<30>("v_min_u32", out);
    run<31>("v_ffbh_u32", out);
    run<32>("v_bcnt_u32_b32", out);
    return 0;
}
