/*
 * cpu_bench.c -- times the CPU side of bench.py's `cpu_baseline` leg (TEST
 * INFRASTRUCTURE, see quicked_oracle.h).  One aligner per OpenMP thread over
 * disjoint pair ranges -- the reference's own parallel model
 * (tools/align_benchmark/align_benchmark.c:246-284).
 *
 * kind "reference": dlopen()s oracle/_ref/libquicked_ref.so (the reference
 * compiled from its own sources) and drives it through its public C-ABI
 * (quicked.h:81-96).  kind "port": the oracle restatement (qo_align).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "quicked_oracle.h"

/* quicked_params_t / quicked_aligner_t of the reference (quicked.h:43-67), layout only */
typedef struct {
    int algo; unsigned bandwidth, window_size, overlap_size, hew_threshold[2], hew_percentage[2];
    _Bool only_score, force_scalar, external_timer; void* external_allocator;
} ref_params_t;
typedef struct { ref_params_t* params; void* mm_allocator; char* cigar; int score; void* timers[5]; } ref_aligner_t;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

/* returns wall seconds, or -1 on error; scores_out[n] */
double cpu_bench_run(const char* ref_so, int n,
                     const char* ppool, const int64_t* poff, const int32_t* plen,
                     const char* tpool, const int64_t* toff, const int32_t* tlen,
                     int algo, int only_score, unsigned bandwidth, int threads, int32_t* scores_out) {
    ref_params_t (*f_default)(void) = NULL;
    int (*f_new)(ref_aligner_t*, ref_params_t*) = NULL;
    int (*f_align)(ref_aligner_t*, const char*, int, const char*, int) = NULL;
    int (*f_free)(ref_aligner_t*) = NULL;
    if (ref_so) {
        void* h = dlopen(ref_so, RTLD_NOW | RTLD_LOCAL);
        if (!h) return -1.0;
        f_default = (ref_params_t (*)(void))dlsym(h, "quicked_default_params");
        f_new = (int (*)(ref_aligner_t*, ref_params_t*))dlsym(h, "quicked_new");
        f_align = (int (*)(ref_aligner_t*, const char*, int, const char*, int))dlsym(h, "quicked_align");
        f_free = (int (*)(ref_aligner_t*))dlsym(h, "quicked_free");
        if (!f_default || !f_new || !f_align || !f_free) return -1.0;
    }
    if (threads > 0) omp_set_num_threads(threads);
    double t0 = 0;
    #pragma omp parallel
    {
        if (ref_so) {
            ref_params_t p = f_default();
            p.algo = algo; p.only_score = only_score ? 1 : 0; p.bandwidth = bandwidth;
            ref_aligner_t a;
            f_new(&a, &p);
            /* NUL-terminated copies: the reference's SSE window kernel reads text[tlen] */
            char* pb = NULL; char* tb = NULL; size_t pc = 0, tc = 0;
            /* aligner construction (128 MiB arena, quicked.c:331) stays outside the timed region */
            #pragma omp barrier
            #pragma omp master
            t0 = now_s();
            #pragma omp barrier
            #pragma omp for schedule(dynamic, 16)
            for (int i = 0; i < n; ++i) {
                if ((size_t)plen[i] + 1 > pc) { pc = (size_t)plen[i] + 64; pb = (char*)realloc(pb, pc); }
                if ((size_t)tlen[i] + 1 > tc) { tc = (size_t)tlen[i] + 64; tb = (char*)realloc(tb, tc); }
                memcpy(pb, ppool + poff[i], (size_t)plen[i]); pb[plen[i]] = 0;
                memcpy(tb, tpool + toff[i], (size_t)tlen[i]); tb[tlen[i]] = 0;
                f_align(&a, pb, plen[i], tb, tlen[i]);
                scores_out[i] = a.score;
            }
            free(pb); free(tb);
            f_free(&a);
        } else {
            qo_params_t p;
            qo_default_params(&p);
            p.algo = algo; p.only_score = only_score; p.bandwidth = bandwidth;
            #pragma omp barrier
            #pragma omp master
            t0 = now_s();
            #pragma omp barrier
            #pragma omp for schedule(dynamic, 16)
            for (int i = 0; i < n; ++i) {
                int sc = -1; char* cg = NULL;
                qo_align(&p, ppool + poff[i], plen[i], tpool + toff[i], tlen[i], &sc, &cg, NULL);
                if (cg) qo_free(cg);
                scores_out[i] = sc;
            }
        }
    }
    return now_s() - t0;
}
