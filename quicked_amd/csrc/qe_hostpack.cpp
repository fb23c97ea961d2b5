// qe_hostpack.cpp -- host-side serializer of whole sequence pools into the packed wire formats of
// include/quicked_batch.h (QUICKED_WIRE_2BIT / QUICKED_WIRE_PLANES3), SIMD and multi-threaded.
//
// Why it exists: the reference consumes the caller's ASCII directly (quicked.c:405-437; its harness keeps the reads in
// sequence_buffer_t, quicked_utils/include/sequence_buffer.h:30-50).  Here ASCII over PCIe Gen5 bounds a GPU at
// ~2.7 M alignments/s of 10 kb (2 GB per 100 k pairs at ~54 GB/s); the 2-bit form is 4x smaller, so a caller that holds
// ASCII packs on the host cores -- inside the end-to-end clock -- and ships 0.5 GB.  The code table is the reference's
// (dna_text.c:41-46: A 0, C 1, G 2, T 3, everything else 4) restricted to the symbols whose raw-byte and encoded
// comparisons agree (upper-case ACGT, N in PLANES3), the same contract as quicked_wire_pack.
//
// Kernel: 64 (AVX-512BW) or 32 (AVX2) bases per iteration.  For the four letters, ASCII bit 2 is code bit 1 and
// ASCII bit 1 ^ bit 2 is code bit 0 ('A' 0x41, 'C' 0x43, 'G' 0x47, 'T' 0x54); a byte-to-mask move of the shifted
// vector yields 64 code bits at once; the 2-bit words interleave the two masks with BMI2 pdep; validation is four
// byte compares on the same vector.  Chosen at run time (__builtin_cpu_supports); a scalar table loop otherwise.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "quicked.h"
#include "quicked_batch.h"

#define QE_API extern "C" __attribute__((visibility("default")))

namespace {

typedef uint64_t u64;

// one block of up to 64 bases -> lo = code bit 0, hi = code bit 1, nn = 'N', ok = representable symbol; bit i = base i
struct Masks { u64 lo, hi, nn, ok; };

static inline Masks masks_scalar(const uint8_t* s, int cnt) {
    Masks m{0, 0, 0, 0};
    for (int i = 0; i < cnt; ++i) {
        const uint8_t c = s[i];
        const u64 bit = (u64)1 << i;
        switch (c) {
            case 'A': m.ok |= bit; break;
            case 'C': m.ok |= bit; m.lo |= bit; break;
            case 'G': m.ok |= bit; m.hi |= bit; break;
            case 'T': m.ok |= bit; m.lo |= bit; m.hi |= bit; break;
            case 'N': m.nn |= bit; break;
            default: break;
        }
    }
    return m;
}

__attribute__((target("avx2"))) static inline Masks masks_avx2(const uint8_t* s) {
    Masks m{0, 0, 0, 0};
    for (int h = 0; h < 2; ++h) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(s + 32 * h));
        const u64 b2 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 5));        // ASCII bit 2 -> bit 7 of its byte
        const u64 b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 6));
        const __m256i eq = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(v, _mm256_set1_epi8('C'))),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(v, _mm256_set1_epi8('T'))));
        const u64 ok = (uint32_t)_mm256_movemask_epi8(eq);
        const u64 nn = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('N')));
        m.lo |= ((b1 ^ b2) & ok) << (32 * h); m.hi |= (b2 & ok) << (32 * h); m.ok |= ok << (32 * h); m.nn |= nn << (32 * h);
    }
    return m;
}

__attribute__((target("avx512f,avx512bw"))) static inline Masks masks_avx512(const uint8_t* s) {
    const __m512i v = _mm512_loadu_si512((const void*)s);
    const u64 b2 = _mm512_movepi8_mask(_mm512_slli_epi16(v, 5));
    const u64 b1 = _mm512_movepi8_mask(_mm512_slli_epi16(v, 6));
    const u64 ok = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('A')) | _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('C')) |
                   _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('G')) | _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('T'));
    const u64 nn = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('N'));
    return Masks{(b1 ^ b2) & ok, b2 & ok, nn, ok};
}

// bits of x spread to the even bit positions of the result (bit i -> bit 2 i), 32 bits in
static inline u64 spread_scalar(u64 x) {
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__attribute__((target("bmi2"))) static inline u64 spread_pdep(u64 x) { return _pdep_u64(x, 0x5555555555555555ull); }

enum Isa { ISA_SCALAR = 0, ISA_AVX2 = 1, ISA_AVX512 = 2 };
static int detect_isa() {
    __builtin_cpu_init();
    const bool bmi2 = __builtin_cpu_supports("bmi2");
    if (bmi2 && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f")) return ISA_AVX512;
    if (bmi2 && __builtin_cpu_supports("avx2")) return ISA_AVX2;
    return ISA_SCALAR;
}
static int g_isa = -1;
static int g_force_isa = -1;          // tests: quicked_wire_pack_isa(k) pins the kernel

// one sequence; returns false on an unrepresentable symbol.  ISA is a template parameter so that each instance is
// compiled for its target as a whole (the block loop inlines its mask kernel)
template <int ISA> struct Pack;
#define QE_PACK_BODY(MASKS_FULL, SPREAD)                                                                                   \
    static bool run(const uint8_t* s, int32_t len, int wire, u64* out) {                                                   \
        const int32_t nfull = len >> 6, tail = len & 63;                                                                   \
        u64 bad = 0;                                                                                                       \
        for (int32_t k = 0; k <= nfull; ++k) {                                                                             \
            Masks m;                                                                                                       \
            int cnt = 64;                                                                                                  \
            if (k < nfull) m = MASKS_FULL(s + 64 * (int64_t)k);                                                            \
            else {                                                                                                         \
                if (!tail) break;                                                                                          \
                uint8_t buf[64];                                                                                           \
                memset(buf, 'A', 64);             /* 'A' packs to zero bits */                                             \
                memcpy(buf, s + 64 * (int64_t)k, (size_t)tail);                                                            \
                m = MASKS_FULL(buf);                                                                                       \
                cnt = tail;                                                                                                \
            }                                                                                                              \
            if (wire == QUICKED_WIRE_2BIT) {                                                                               \
                bad |= ~m.ok;                                                                                              \
                out[2 * (int64_t)k] = SPREAD(m.lo) | (SPREAD(m.hi) << 1);                                                  \
                if (cnt > 32) out[2 * (int64_t)k + 1] = SPREAD(m.lo >> 32) | (SPREAD(m.hi >> 32) << 1);                    \
            } else {                                                                                                       \
                bad |= ~(m.ok | m.nn);                                                                                     \
                u64* row = out + 3 * (int64_t)k;                                                                           \
                row[0] = m.lo; row[1] = m.hi; row[2] = m.nn;                                                               \
            }                                                                                                              \
        }                                                                                                                  \
        return bad == 0;                                                                                                   \
    }

static inline Masks masks_scalar64(const uint8_t* s) { return masks_scalar(s, 64); }
template <> struct Pack<ISA_SCALAR> { QE_PACK_BODY(masks_scalar64, spread_scalar) };
template <> struct Pack<ISA_AVX2> { __attribute__((target("avx2,bmi2"))) QE_PACK_BODY(masks_avx2, spread_pdep) };
template <> struct Pack<ISA_AVX512> { __attribute__((target("avx512f,avx512bw,bmi2"))) QE_PACK_BODY(masks_avx512, spread_pdep) };

static bool pack_one(int isa, const uint8_t* s, int32_t len, int wire, u64* out) {
    switch (isa) {
        case ISA_AVX512: return Pack<ISA_AVX512>::run(s, len, wire, out);
        case ISA_AVX2: return Pack<ISA_AVX2>::run(s, len, wire, out);
        default: return Pack<ISA_SCALAR>::run(s, len, wire, out);
    }
}

static int usable_threads() {
    cpu_set_t set;
    int n = (int)std::thread::hardware_concurrency();
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    return std::max(1, std::min(n, 32));
}

}  // namespace

// 0 scalar, 1 AVX2 + BMI2, 2 AVX-512BW + BMI2: the kernel quicked_wire_pack_pool uses on this host; force >= 0 pins it
// (tests run every kernel the CPU has), force < 0 restores the detection.  Returns the kernel in use, -1 if the CPU
// lacks the one asked for.
QE_API int quicked_wire_pack_isa(int force) {
    if (g_isa < 0) g_isa = detect_isa();
    if (force >= 0) {
        if (force > g_isa) return -1;
        g_force_isa = force;
    } else g_force_isa = -1;
    return g_force_isa >= 0 ? g_force_isa : g_isa;
}

QE_API int64_t quicked_wire_offsets(int64_t n, const int32_t* len, int wire, int64_t* out_off) {
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t w = quicked_wire_words(len[i], wire);
        if (w < 0) return -1;
        if (out_off) out_off[i] = total;
        total += w;
    }
    return total;
}

QE_API quicked_status_t quicked_wire_pack_pool(int64_t n, const char* pool, const int64_t* off, const int32_t* len, int wire,
                                               uint64_t* out_words, const int64_t* out_off, int threads, int64_t* bad_seq) {
    if (bad_seq) *bad_seq = -1;
    if (n < 0 || (wire != QUICKED_WIRE_2BIT && wire != QUICKED_WIRE_PLANES3)) return QUICKED_ERROR;
    if (n == 0) return QUICKED_OK;
    if (!pool || !off || !len || !out_words || !out_off) return QUICKED_ERROR;
    const int isa = quicked_wire_pack_isa(g_force_isa);
    int64_t bytes = 0;
    for (int64_t i = 0; i < n; ++i) { if (len[i] < 0) return QUICKED_ERROR; bytes += len[i]; }
    int nt = threads > 0 ? std::min(threads, 64) : usable_threads();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, bytes / (1 << 20) + 1));        // a thread per MB at least
    std::atomic<int64_t> first_bad{INT64_MAX};
    auto work = [&](int w) {
        // contiguous shares of the BYTES, cut at sequence borders
        const int64_t lo_b = bytes * w / nt, hi_b = bytes * (w + 1) / nt;
        int64_t acc = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t start = acc;
            acc += len[i];
            if (start < lo_b || start >= hi_b) { if (!(len[i] == 0 && w == 0)) continue; }
            if (len[i] == 0) continue;
            if (!pack_one(isa, (const uint8_t*)pool + off[i], len[i], wire, out_words + out_off[i])) {
                int64_t cur = first_bad.load();
                while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
            }
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        th.reserve((size_t)nt);
        for (int w = 0; w < nt; ++w) th.emplace_back(work, w);
        for (auto& t : th) t.join();
    }
    if (first_bad.load() != INT64_MAX) {
        if (bad_seq) *bad_seq = first_bad.load();
        return QUICKED_ERROR;
    }
    return QUICKED_OK;
}
