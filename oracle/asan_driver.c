/*
 * asan_driver.c -- the oracle and the seeded generator under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU
 * (TEST INFRASTRUCTURE; `make -C oracle asan` builds and runs it).  The reference's own build offers the same for
 * its sources (CMakeLists.txt:43-49: ASAN / UBSAN switches); GPU sanitizers are not available on the MI355X pool, so
 * the host-side restatement of every kernel's arithmetic is what gets sanitised.
 *
 * Walks shapes that stress the index arithmetic: lengths around block borders, length mismatch, empty-ish inputs,
 * N / lower-case symbols, large indels (all QuickEd stages), forced deep Hirschberg splits; every alignment is
 * checked (valid CIGAR, edit count == score, QuickEd / BandEd / Hirschberg score == full-height exact distance).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quicked_oracle.h"

int64_t qe_gen_pattern_capacity(int64_t length, double error, int64_t indels_num, int64_t indels_len);
int64_t qe_gen_pair(uint64_t seed, uint64_t index, int64_t length, double error, int64_t indels_num, int64_t indels_len,
                    char* pattern_out, char* text_out);

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); ++failures; } } while (0)

static void one(const char* p, int m, const char* t, int n, int exact_algos) {
    static const struct { int algo, only_score; unsigned bw, w, o; int scalar; } RUNS[] = {
        {QO_QUICKED, 0, 15, 9, 1, 0}, {QO_QUICKED, 1, 15, 9, 1, 1}, {QO_BANDED, 1, 15, 9, 1, 0}, {QO_BANDED, 1, 1, 9, 1, 0},
        {QO_BANDED, 0, 15, 9, 1, 0}, {QO_HIRSCHBERG, 0, 15, 9, 1, 0}, {QO_WINDOWED, 0, 15, 2, 1, 0}, {QO_WINDOWED, 1, 15, 2, 1, 1},
        {QO_WINDOWED, 0, 15, 4, 2, 0}, {QO_WINDOWED, 1, 15, 9, 1, 0},
    };
    const int64_t exact = (m && n) ? qo_exact_distance(p, m, t, n) : -1;
    for (size_t r = 0; r < sizeof(RUNS) / sizeof(RUNS[0]); ++r) {
        qo_params_t prm;
        qo_default_params(&prm);
        prm.algo = RUNS[r].algo; prm.only_score = RUNS[r].only_score; prm.bandwidth = RUNS[r].bw;
        prm.window_size = RUNS[r].w; prm.overlap_size = RUNS[r].o; prm.force_scalar = RUNS[r].scalar;
        int score = -1; char* cg = NULL; qo_trace_t tr;
        const int st = qo_align(&prm, p, m, t, n, &score, &cg, &tr);
        if (m == 0 || n == 0) { CHECK(st == QO_EMPTY_SEQUENCE, "empty input: status %d", st); qo_free(cg); continue; }
        CHECK(st >= 0, "algo %d: status %d (m %d n %d)", prm.algo, st, m, n);
        if (cg) {
            const int64_t nops_max = (int64_t)m + n + 1;
            char* ops = (char*)malloc((size_t)nops_max + 1);
            const int64_t nops = qo_rle_to_ops(cg, ops, nops_max);
            CHECK(nops > 0 && qo_cigar_check(p, m, t, n, ops, nops), "algo %d: invalid CIGAR (m %d n %d)", prm.algo, m, n);
            CHECK(qo_cigar_score(ops, nops) == score, "algo %d: CIGAR edits != score", prm.algo);
            free(ops);
        }
        /* exact on upper-case ACGT input with a band that holds the distance (bandwidth 15 %, <= 10 % divergence here) */
        if (exact_algos && prm.bandwidth == 15 && (prm.algo == QO_QUICKED || ((prm.algo == QO_BANDED || prm.algo == QO_HIRSCHBERG) && exact * 100 <= (int64_t)15 * (m > n ? m : n))))
            CHECK(score == exact, "algo %d only_score %d: score %d != exact %lld (m %d n %d)", prm.algo, prm.only_score, score, (long long)exact, m, n);
        qo_free(cg);
    }
}

int main(void) {
    static const struct { int64_t count, length; double error; int64_t in, il; } SETS[] = {
        {24, 1, 0, 0, 0}, {24, 5, 2, 0, 0}, {24, 63, 0.1, 0, 0}, {24, 64, 0.1, 0, 0}, {24, 65, 0.1, 0, 0}, {24, 130, 0.08, 0, 0},
        {16, 1024, 0.05, 0, 0}, {8, 3000, 0.1, 0, 0}, {4, 10000, 0.05, 0, 0}, {4, 10000, 0.05, 4, 800}, {6, 2000, 0.3, 0, 0},
    };
    for (size_t s = 0; s < sizeof(SETS) / sizeof(SETS[0]); ++s) {
        const int64_t cap = qe_gen_pattern_capacity(SETS[s].length, SETS[s].error, SETS[s].in, SETS[s].il);
        char* p = (char*)malloc((size_t)cap + 1); char* t = (char*)malloc((size_t)SETS[s].length + 1);
        for (int64_t i = 0; i < SETS[s].count; ++i) {
            const int64_t m = qe_gen_pair(0xA5A5 + s, (uint64_t)i, SETS[s].length, SETS[s].error, SETS[s].in, SETS[s].il, p, t);
            /* exact-size heap copies: a one-byte over-read is a sanitizer hit */
            char* pp = (char*)malloc((size_t)(m > 0 ? m : 1)); char* tt = (char*)malloc((size_t)SETS[s].length);
            memcpy(pp, p, (size_t)m); memcpy(tt, t, (size_t)SETS[s].length);
            one(pp, (int)m, tt, (int)SETS[s].length, SETS[s].error <= 0.1);
            if (i % 4 == 1 && m > 8) {                      /* N, IUPAC and lower-case symbols, ragged lengths */
                pp[m / 3] = 'N'; tt[SETS[s].length / 2] = 'n'; pp[m / 2] = 'R'; tt[0] = 'a';
                one(pp, (int)(m - m / 5), tt, (int)SETS[s].length, 0);
                one(pp, (int)m, tt, (int)(SETS[s].length - SETS[s].length / 7), 0);
            }
            free(pp); free(tt);
        }
        free(p); free(t);
    }
    one("ACGT", 4, "", 0, 0); one("", 0, "ACGT", 4, 0);
    /* forced deep Hirschberg recursion (qo_hirschberg with a 4 KiB split threshold) */
    {
        const int64_t len = 2500, cap = qe_gen_pattern_capacity(len, 0.08, 0, 0);
        char* p = (char*)malloc((size_t)cap + 1); char* t = (char*)malloc((size_t)len + 1);
        for (int i = 0; i < 6; ++i) {
            const int64_t m = qe_gen_pair(77, (uint64_t)i, len, 0.08, 0, 0, p, t);
            const int64_t exact = qo_exact_distance(p, (int)m, t, (int)len);
            char* ops = (char*)malloc((size_t)(m + len + 1)); int64_t nops = 0; qo_trace_t tr;
            const int st = qo_hirschberg(p, (int)m, t, (int)len, exact, 1 << 12, ops, &nops, &tr);
            CHECK(st == QO_OK && tr.hirschberg_splits > 3, "forced splits: status %d, %lld splits", st, (long long)tr.hirschberg_splits);
            CHECK(qo_cigar_check(p, (int)m, t, (int)len, ops, nops) && qo_cigar_score(ops, nops) == exact, "forced splits: not optimal");
            free(ops);
        }
        free(p); free(t);
    }
    if (failures) { fprintf(stderr, "asan_driver: %d failure(s)\n", failures); return 1; }
    printf("asan_driver: all checks passed under ASAN + UBSAN\n");
    return 0;
}
