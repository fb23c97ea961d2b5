// pmc_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the kernels use
// (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of a 16 B/lane streaming read; "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams a buffer far
// larger than the 256 MiB Infinity Cache exactly once, as [row][lane] rows like the BandEd kernels' state:
//   k_read<4|8|16>   one 4 / 8 / 16-byte load per lane per row      k_write<8|16>   the same with stores
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/bin/pmc_calib      (and --pmc WRITE_SIZE)
// and prints the bytes every kernel moved; tools/summarise_pmc.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T> __global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, uint64_t* sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T v = src[i];
        const unsigned char* b = (const unsigned char*)&v;
        acc += b[0] + b[sizeof(T) - 1];
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
template <typename T> __global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v; __builtin_memset(&v, 1, sizeof(T));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = v;
}
template __global__ void k_read<uint32_t>(const uint32_t*, uint64_t*, size_t);
template __global__ void k_read<uint64_t>(const uint64_t*, uint64_t*, size_t);
template __global__ void k_read<uint4>(const uint4*, uint64_t*, size_t);
template __global__ void k_write<uint64_t>(uint64_t*, size_t);
template __global__ void k_write<uint4>(uint4*, size_t);

int main() {
    const size_t bytes = (size_t)4 << 30;
    void* buf; uint64_t* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 3, bytes);
    hipDeviceSynchronize();
    const int blocks = 256 * 8;
    k_read<uint32_t><<<blocks, 256>>>((const uint32_t*)buf, sink, bytes / 4);
    k_read<uint64_t><<<blocks, 256>>>((const uint64_t*)buf, sink, bytes / 8);
    k_read<uint4><<<blocks, 256>>>((const uint4*)buf, sink, bytes / 16);
    k_write<uint64_t><<<blocks, 256>>>((uint64_t*)buf, bytes / 8);
    k_write<uint4><<<blocks, 256>>>((uint4*)buf, bytes / 16);
    hipDeviceSynchronize();
    printf("every kernel moved %zu bytes (%.1f KiB)\n", bytes, bytes / 1024.0);
    return 0;
}
