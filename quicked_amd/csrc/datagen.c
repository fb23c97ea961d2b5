/*
 * datagen.c -- seeded synthetic read-pair generator (libqe_datagen.so).
 *
 * Re-states the DISTRIBUTION of the reference's tools/generate_dataset
 * (generate_dataset.c:52-63 random text, 108-199 sequential mismatch /
 * deletion / insertion edits, 203-246 large deletions) with a fixed-seed
 * counter-based PRNG instead of srand(time(0)) (generate_dataset.c:375), so
 * that pair i of a dataset depends only on (seed, i) and shards can be
 * generated independently.  Host-side tooling: not on the
 * alignment path, not part of the oracle.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t s; } rng_t;

static inline uint64_t rng_next(rng_t* r) {            /* splitmix64 */
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* uniform integer in [0, n), n < 2^32, unbiased (multiply-shift with rejection) */
static inline uint32_t rng_below(rng_t* r, uint32_t n) {
    uint64_t m = (uint64_t)(uint32_t)rng_next(r) * n;
    uint32_t lo = (uint32_t)m;
    if (lo < n) {
        const uint32_t t = (uint32_t)(-n) % n;
        while (lo < t) { m = (uint64_t)(uint32_t)rng_next(r) * n; lo = (uint32_t)m; }
    }
    return (uint32_t)(m >> 32);
}

static const char ALPHABET[4] = {'A', 'C', 'G', 'T'};

/* capacity one pair needs: text L bytes, pattern up to L + n_errors bytes */
int64_t qe_gen_pattern_capacity(int64_t length, double error, int64_t indels_num, int64_t indels_len) {
    (void)indels_num; (void)indels_len;
    const int64_t e = (error >= 1.0) ? (int64_t)error : (int64_t)ceil((double)length * error);
    return length + e + 1;
}

/* Generates pair `index` of dataset `seed`.  text_out gets `length` bytes,
 * pattern_out up to qe_gen_pattern_capacity() bytes; returns the pattern
 * length.  error < 1: fraction (ceil(length*error) edits); error >= 1: that
 * many edits (generate_dataset.c:370).  indels_num/indels_len: up to
 * indels_num deletions of indels_len bases (the stage-2/3 trigger data). */
int64_t qe_gen_pair(uint64_t seed, uint64_t index, int64_t length, double error,
                    int64_t indels_num, int64_t indels_len,
                    char* pattern_out, char* text_out) {
    rng_t r;
    r.s = seed ^ (index * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
    rng_next(&r);
    for (int64_t i = 0; i < length; ++i) text_out[i] = ALPHABET[rng_next(&r) >> 62];
    memcpy(pattern_out, text_out, (size_t)length);
    int64_t len = length;
    const int64_t e = (error >= 1.0) ? (int64_t)error : (int64_t)ceil((double)length * error);
    for (int64_t k = 0; k < e; ++k) {
        const uint32_t kind = rng_below(&r, 3);
        if (kind == 0 && len > 0) {                              /* mismatch to a different base */
            uint32_t pos; char c;
            do { pos = rng_below(&r, (uint32_t)len); c = ALPHABET[rng_below(&r, 4)]; } while (pattern_out[pos] == c);
            pattern_out[pos] = c;
        } else if (kind == 1 && len > 1) {                       /* delete one base */
            const uint32_t pos = rng_below(&r, (uint32_t)len);
            memmove(pattern_out + pos, pattern_out + pos + 1, (size_t)(len - 1 - pos));
            --len;
        } else {                                                 /* insert one base */
            const uint32_t pos = rng_below(&r, (uint32_t)(len > 0 ? len : 1));
            memmove(pattern_out + pos + 1, pattern_out + pos, (size_t)(len - pos));
            pattern_out[pos] = ALPHABET[rng_below(&r, 4)];
            ++len;
        }
    }
    if (indels_num > 0) {
        const uint32_t n = rng_below(&r, (uint32_t)indels_num + 1);
        for (uint32_t k = 0; k < n; ++k) {
            if (indels_len >= len) break;
            const uint32_t pos = rng_below(&r, (uint32_t)(len - indels_len + 1));
            memmove(pattern_out + pos, pattern_out + pos + indels_len, (size_t)(len - indels_len - pos));
            len -= indels_len;
        }
    }
    return len;
}

/* Batch form: pairs [first, first+count) written back to back into the two
 * byte pools; offsets/lengths arrays have count entries.  Pools must hold
 * count*qe_gen_pattern_capacity() and count*length bytes.  OpenMP-parallel. */
void qe_gen_batch(uint64_t seed, uint64_t first, int64_t count, int64_t length, double error,
                  int64_t indels_num, int64_t indels_len,
                  char* pattern_pool, int64_t* pattern_off, int32_t* pattern_len,
                  char* text_pool, int64_t* text_off, int32_t* text_len) {
    const int64_t cap = qe_gen_pattern_capacity(length, error, indels_num, indels_len);
    /* a team only where there is work for one (a team of 256 threads costs 0.1 s to wake on a 256-thread box: callers
     * that generate one pair at a time paid that per pair) */
    #pragma omp parallel for schedule(dynamic, 64) if (count >= 256)
    for (int64_t i = 0; i < count; ++i) {
        pattern_off[i] = i * cap;
        text_off[i] = i * length;
        pattern_len[i] = (int32_t)qe_gen_pair(seed, first + (uint64_t)i, length, error, indels_num, indels_len,
                                              pattern_pool + i * cap, text_pool + i * length);
        text_len[i] = (int32_t)length;
    }
}
