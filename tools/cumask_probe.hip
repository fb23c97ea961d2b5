// Does a CU mask isolate a latency-bound launch from a chip-filling one on gfx950?  (hipExtStreamCreateWithCUMask)
//   small: 16 one-wave workgroups, each a long dependent VALU chain (what an early-finish flow's kernels are)
//   big:   a launch that keeps two 4-wave workgroups on every CU busy for the whole time (what a 100 k-pair run is)
// Prints small's duration alone, beside big on ordinary streams, and with big masked to the complement of small's CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void chain(unsigned* out, int iters) {
    unsigned x = threadIdx.x + 1, y = blockIdx.x;
    for (int i = 0; i < iters; ++i) { x = x * 1664525u + y; y ^= x >> 7; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y;
}
int main(int argc, char** argv) {
    const int reserve = argc > 1 ? atoi(argv[1]) : 32;                 // CUs for the small launch
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("CUs %d, reserving %d for the small launch\n", ncu, reserve);
    unsigned* buf; CK(hipMalloc(&buf, (size_t)1 << 26));
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> m_small(words, 0), m_big(words, 0);
    // every (ncu / reserve)-th CU goes to the small launch: spreads the reservation over the XCDs whatever the numbering
    const int stride = ncu / reserve;
    for (int c = 0; c < ncu; ++c) { if (c % stride == 0) m_small[c / 32] |= 1u << (c % 32); else m_big[c / 32] |= 1u << (c % 32); }
    hipStream_t s_small, s_big, m_s, m_b;
    CK(hipStreamCreateWithFlags(&s_small, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_big, hipStreamNonBlocking));
    CK(hipExtStreamCreateWithCUMask(&m_s, words, m_small.data())); CK(hipExtStreamCreateWithCUMask(&m_b, words, m_big.data()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_small = [&](hipStream_t ss, hipStream_t sb, bool with_big) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            if (with_big) hipLaunchKernelGGL(chain, dim3(ncu * 8 * 6), dim3(256), 54 * 1024, sb, buf + (1 << 20), 60000);   // ~6 rounds of two workgroups per CU
            CK(hipEventRecord(e0, ss));
            hipLaunchKernelGGL(chain, dim3(16), dim3(64), 0, ss, buf, 400000);
            CK(hipEventRecord(e1, ss));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            CK(hipDeviceSynchronize());
        }
        return best;
    };
    auto time_big = [&](hipStream_t sb) {
        CK(hipEventRecord(e0, sb));
        hipLaunchKernelGGL(chain, dim3(ncu * 8 * 6), dim3(256), 54 * 1024, sb, buf + (1 << 20), 60000);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    CK(hipFuncSetAttribute((const void*)chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    time_small(s_small, s_big, false);
    printf("small alone:                          %.3f ms\n", time_small(s_small, s_big, false));
    printf("small alone on its masked stream:     %.3f ms\n", time_small(m_s, m_b, false));
    printf("small beside big, ordinary streams:   %.3f ms\n", time_small(s_small, s_big, true));
    printf("small beside big, both masked:        %.3f ms\n", time_small(m_s, m_b, true));
    printf("small masked beside big unmasked:     %.3f ms\n", time_small(m_s, s_big, true));
    printf("big alone, ordinary / masked stream:  %.3f / %.3f ms\n", time_big(s_big), time_big(m_b));
    return 0;
}
