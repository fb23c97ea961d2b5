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
    run<3>("v_fma_f32", out); run<4>("v_add_u32", out);
    return 0;
}
