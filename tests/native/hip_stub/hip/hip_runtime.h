// A header-compatible FAKE of the ~35 HIP runtime calls the host side of libquicked_hip.so uses, for building that host side
// (qe_driver.hip's host half, qe_stages.hip, qe_pool.h, qe_batch.h, qe_capi.cpp) with g++ under ThreadSanitizer /
// AddressSanitizer + UBSan on a machine without a GPU (tests/test_host_sanitizers.py; the reference offers ASAN / UBSAN for
// its whole library, CMakeLists.txt:43-49).  TEST INFRASTRUCTURE: nothing of the product includes this file.
//
// Model: "device memory" is host memory (calloc) under a byte budget that can be made to fail (hipErrorOutOfMemory) --
// QE_STUB_HBM_BYTES, default 8 GiB; a kernel launch runs the kernel's host stand-in (qe_kernels_stub.h: real code for the
// copy / scan / decide kernels the host logic depends on, no-ops for the alignment kernels) at once in the launching thread,
// so a stream is always drained and an event is complete as soon as it is recorded.  What the sanitizers then see is the
// host layer itself: contexts on lease, the per-device book, the rotation, queued runs and fetches, the early-finish
// threads and their hand-over of batch objects, reclaim under a budget -- with real threads.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <unordered_set>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorInvalidDevice = 101, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };

struct ihipStream_t { int id; };
struct ihipEvent_t { std::atomic<int> recorded{0}; };
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace hipstub {
struct State {
    std::mutex mu;
    std::unordered_map<void*, size_t> dev;           // device allocations -> bytes
    std::unordered_set<void*> pinned;
    size_t used = 0, budget = 0;
    std::atomic<long> launches{0}, mallocs{0}, oom{0};
    State() {
        const char* e = getenv("QE_STUB_HBM_BYTES");
        budget = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)8 << 30);
    }
};
inline State& st() { static State s; return s; }
inline thread_local int tl_dev = 0;
inline thread_local hipError_t tl_last = hipSuccess;
inline hipError_t fail(hipError_t e) { tl_last = e; return e; }
}  // namespace hipstub

static inline const char* hipGetErrorString(hipError_t e) {
    switch (e) { case hipSuccess: return "hipSuccess"; case hipErrorOutOfMemory: return "hipErrorOutOfMemory"; case hipErrorNotReady: return "hipErrorNotReady";
                 case hipErrorInvalidValue: return "hipErrorInvalidValue"; case hipErrorInvalidDevice: return "hipErrorInvalidDevice"; default: return "hipErrorUnknown"; }
}
static inline hipError_t hipGetLastError() { const hipError_t e = hipstub::tl_last; hipstub::tl_last = hipSuccess; return e; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d != 0) return hipstub::fail(hipErrorInvalidDevice); hipstub::tl_dev = d; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

static inline hipError_t hipMalloc(void** p, size_t bytes) {
    auto& S = hipstub::st();
    std::lock_guard<std::mutex> lk(S.mu);
    ++S.mallocs;
    if (S.used + bytes > S.budget) { ++S.oom; *p = nullptr; return hipstub::fail(hipErrorOutOfMemory); }
    // the product sizes pools for a 288 GB device; the stand-in backs an allocation with zero pages the kernel only
    // materialises when touched (calloc of a large block is an mmap)
    void* q = calloc(1, bytes ? bytes : 1);
    if (!q) { ++S.oom; *p = nullptr; return hipstub::fail(hipErrorOutOfMemory); }
    S.dev[q] = bytes; S.used += bytes; *p = q;
    return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
static inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    auto& S = hipstub::st();
    std::lock_guard<std::mutex> lk(S.mu);
    auto it = S.dev.find(p);
    if (it == S.dev.end()) return hipstub::fail(hipErrorInvalidValue);
    S.used -= it->second; S.dev.erase(it);
    free(p);
    return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    auto& S = hipstub::st();
    std::lock_guard<std::mutex> lk(S.mu);
    *total_b = S.budget; *free_b = S.budget > S.used ? S.budget - S.used : 0;
    return hipSuccess;
}
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) {
    void* q = calloc(1, bytes ? bytes : 1);
    if (!q) return hipstub::fail(hipErrorOutOfMemory);
    auto& S = hipstub::st();
    std::lock_guard<std::mutex> lk(S.mu);
    S.pinned.insert(q); *p = q;
    return hipSuccess;
}
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned f = 0) { return hipHostMalloc((void**)p, bytes, f); }
static inline hipError_t hipHostFree(void* p) {
    if (!p) return hipSuccess;
    auto& S = hipstub::st();
    { std::lock_guard<std::mutex> lk(S.mu); S.pinned.erase(p); }
    free(p);
    return hipSuccess;
}
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* at, const void* p) {
    auto& S = hipstub::st();
    std::lock_guard<std::mutex> lk(S.mu);
    at->device = 0; at->devicePointer = const_cast<void*>(p); at->hostPointer = const_cast<void*>(p);
    at->type = S.pinned.count(const_cast<void*>(p)) ? hipMemoryTypeHost : hipMemoryTypeUnregistered;
    return at->type == hipMemoryTypeHost ? hipSuccess : hipstub::fail(hipErrorInvalidValue);
}

static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static std::atomic<int> n{0}; *s = new ihipStream_t{++n}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->recorded.store(1, std::memory_order_release); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t e) { (void)e->recorded.load(std::memory_order_acquire); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { (void)e->recorded.load(std::memory_order_acquire); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { (void)e->recorded.load(std::memory_order_acquire); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }

static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// a launch = the kernel's host stand-in, run at once by the launching thread
template <typename K, typename... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    (void)grid; (void)block;
    ++hipstub::st().launches;
    kernel(args...);
}
