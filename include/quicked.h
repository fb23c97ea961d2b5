/*
 * quicked.h -- the drop-in C-ABI of libquicked_hip.so (MI355X / gfx950).
 *
 * Binary-compatible with the reference's public header quicked/quicked.h:36-96
 * (x86-64 SysV): same six entry points, same enum values, same layouts of
 * quicked_params_t (48 B) and quicked_aligner_t (72 B).  A caller built against
 * the reference header can be relinked against libquicked_hip.so unchanged;
 * a caller built against THIS header does not need the reference's
 * quicked_utils headers -- the two helper types the reference's ABI leaks
 * (mm_allocator_t, profiler_timer_t) are declared here with the same layout:
 *
 *   profiler_counter_t   quicked_utils/include/profiler_counter.h:34-43 (64 B)
 *   profiler_timer_t     quicked_utils/include/profiler_timer.h:51-57   (88 B)
 *   mm_allocator_t       quicked_utils/include/mm_allocator.h:43-54     (56 B)
 *
 * Each prototype cites the reference definition it replaces.
 */
#ifndef QUICKED_H
#define QUICKED_H

#include <stdbool.h>
#include <stdint.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUICKED_WINDOW_STAGES 2        /* quicked.h:32 */
#define QUICKED_FAST_WINDOW_SIZE 2     /* quicked.h:33 */
#define QUICKED_FAST_WINDOW_OVERLAP 1  /* quicked.h:34 */

/* ---- helper types visible through the ABI (skipped when the reference's own
 *      quicked_utils headers were included first) ---------------------------- */
#ifndef PROFILER_COUNTER_H_
#define PROFILER_COUNTER_H_
typedef struct {
    uint64_t total;
    uint64_t samples;
    uint64_t min;
    uint64_t max;
    double m_oldM;
    double m_newM;
    double m_oldS;
    double m_newS;
} profiler_counter_t;
#endif

#ifndef PROFILER_TIMER_H
#define PROFILER_TIMER_H
typedef struct {
    struct timespec begin_timer;   /* CLOCK_REALTIME stamp of the running lap */
    profiler_counter_t time_ns;    /* total / samples / min / max / Welford   */
    uint64_t accumulated;          /* ns of the running lap                   */
} profiler_timer_t;
#endif

#ifndef MM_ALLOCATOR_H_
#define MM_ALLOCATOR_H_
/* Host arena of the reference.  libquicked_hip.so never allocates from it (all
 * scratch lives in the HIP device pool); the type exists so that
 * quicked_params_t.external_allocator and quicked_aligner_t.mm_allocator keep
 * their meaning for callers that pass one through. */
typedef struct {
    uint64_t request_ticker;
    uint64_t segment_size;
    void* segments;
    void* segments_free;
    uint64_t current_segment_idx;
    void* malloc_requests;
    uint64_t malloc_requests_freed;
} mm_allocator_t;
#endif

/* ---- quicked.h:36-41 ------------------------------------------------------- */
typedef enum {
    QUICKED,
    WINDOWED,
    BANDED,
    HIRSCHBERG,
} quicked_algo_t;

/* ---- quicked.h:43-54 ------------------------------------------------------- */
typedef struct quicked_params_t {
    quicked_algo_t algo;
    unsigned int bandwidth;      /* BandEd / Hirschberg cutoff, % of max(plen, tlen)        */
    unsigned int window_size;    /* WindowEd(L) window, in 64-row blocks                     */
    unsigned int overlap_size;   /* WindowEd(L) overlap, in blocks                           */
    unsigned int hew_threshold[QUICKED_WINDOW_STAGES];   /* % errors that make a window "high error" */
    unsigned int hew_percentage[QUICKED_WINDOW_STAGES];  /* % of HEWs that escalates to the next stage */
    bool only_score;
    bool force_scalar;           /* false: x86-default WindowEd(2,1) semantics (SSE kernel); true: scalar */
    bool external_timer;
    mm_allocator_t* external_allocator;
} quicked_params_t;

/* ---- quicked.h:56-67 ------------------------------------------------------- */
typedef struct quicked_aligner_t {
    quicked_params_t* params;    /* the caller's object, not a copy (quicked.c:327)          */
    mm_allocator_t* mm_allocator;
    char* cigar;                 /* RLE string "12M1X3I..." or NULL; owned until quicked_free */
    int score;
    profiler_timer_t* timer;
    profiler_timer_t* timer_windowed_s;
    profiler_timer_t* timer_windowed_l;
    profiler_timer_t* timer_banded;
    profiler_timer_t* timer_align;
} quicked_aligner_t;

/* ---- quicked.h:69-79 ------------------------------------------------------- */
typedef enum quicked_status_t {
    QUICKED_OK                   = 0,
    QUICKED_ERROR                = -1,
    QUICKED_FAIL_NON_CONVERGENCE = -2,
    QUICKED_UNKNOWN_ALGO         = -3,
    QUICKED_EMPTY_SEQUENCE       = -4,
    QUICKED_UNIMPLEMENTED        = -10,
    QUICKED_WIP                  = 1,   /* "not an error": what new/free/most aligns return   */
} quicked_status_t;

/* quicked.c:380 */
bool quicked_check_error(quicked_status_t status);
/* quicked.c:382-403 (same strings) */
const char* quicked_status_msg(quicked_status_t status);
/* quicked.c:308-321 */
quicked_params_t quicked_default_params(void);
/* quicked.c:323-352 */
quicked_status_t quicked_new(quicked_aligner_t* aligner, quicked_params_t* params);
/* quicked.c:354-378 */
quicked_status_t quicked_free(quicked_aligner_t* aligner);
/* quicked.c:405-437 -- one pair, synchronous; runs on the current HIP device */
quicked_status_t quicked_align(quicked_aligner_t* aligner,
                               const char* pattern, const int pattern_len,
                               const char* text, const int text_len);

#ifdef __cplusplus
}
#endif
#endif /* QUICKED_H */
