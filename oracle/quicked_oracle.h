/*
 * quicked_oracle.h -- CPU restatement (ORACLE) of the QuickEd hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quicked_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it -- as the checker, never as the thing
 * being measured or shipped.
 *
 * Every function states which part of the reference (maxdoblas/QuickEd @
 * 2024-10-22) it follows, as file:line relative to the reference root.  The
 * text is a from-scratch restatement (plain column-by-column C, no SIMD, no
 * arena allocator); parity with the compiled reference (oracle/_ref) is pinned
 * by tests/test_oracle_vs_ref.py in the build container and by the golden
 * vectors under tests/golden/ everywhere else.
 */
#ifndef QUICKED_ORACLE_H
#define QUICKED_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* quicked_algo_t / quicked_status_t values (quicked/quicked.h:36-41,69-79) */
enum { QO_QUICKED = 0, QO_WINDOWED = 1, QO_BANDED = 2, QO_HIRSCHBERG = 3 };
enum {
    QO_OK = 0, QO_ERROR = -1, QO_FAIL_NON_CONVERGENCE = -2, QO_UNKNOWN_ALGO = -3,
    QO_EMPTY_SEQUENCE = -4, QO_UNIMPLEMENTED = -10, QO_WIP = 1
};

/* mirrors quicked_params_t (quicked/quicked.h:43-54) minus the host pointers */
typedef struct {
    int32_t  algo;
    uint32_t bandwidth;
    uint32_t window_size;
    uint32_t overlap_size;
    uint32_t hew_threshold[2];
    uint32_t hew_percentage[2];
    int32_t  only_score;
    int32_t  force_scalar;   /* 0: x86-default semantics (SSE W==2 window kernel); 1: scalar */
} qo_params_t;

/* what the bound-and-align driver did (for stage-level parity tests) */
typedef struct {
    int64_t ws_score, ws_hew;         /* stage 1  WindowEd(2,1)                */
    int64_t wl_score, wl_hew;         /* stage 2  WindowEd(W,O) fwd/rev merged */
    int64_t wl_fwd_score, wl_rev_score;
    int32_t stage;                    /* 1, 2 or 3 = last bounding stage run   */
    int32_t banded_calls;             /* stage 3 score-only BandEd calls       */
    int64_t bound;                    /* cutoff handed to the align step       */
    int64_t hirschberg_splits;        /* number of split nodes                 */
    int64_t leaves;                   /* number of leaf alignments             */
    int64_t score_block_advances;     /* sum over score-only BandEd passes     */
    int64_t fill_block_advances;      /* sum over leaf fills                   */
    int64_t window_block_steps;       /* sum over WindowEd windows             */
    int64_t traceback_steps;          /* leaf traceback steps                  */
} qo_trace_t;

void qo_default_params(qo_params_t* p);               /* quicked.c:308-321 */
const char* qo_status_msg(int status);                /* quicked.c:382-403 */

/* quicked_align (quicked.c:405-437).  *cigar_out is malloc()ed (NUL-terminated
 * RLE string, cigar.c:453-488) or NULL; caller frees with qo_free(). */
int qo_align(const qo_params_t* params,
             const char* pattern, int plen, const char* text, int tlen,
             int* score_out, char** cigar_out, qo_trace_t* trace /* may be NULL */);
void qo_free(void* p);

/* BandEd score-only (bpm_banded.c:791-964 == _avx 349-788).  Returns the score,
 * or -1 when the band never reached the last pattern block (reference: read of
 * uninitialised memory).  tfin = text_finish_pos. */
int64_t qo_banded_score(const char* pattern, int plen, const char* text, int tlen,
                        int64_t cutoff_in, int tfin,
                        int64_t* first_out, int64_t* last_out, int64_t* block_advances);

/* BandEd fill + traceback (bpm_banded.c:199-316, 967-1036).  ops[] receives the
 * edit operations front-to-back (at most plen+tlen), returns their number.
 * score_out = the fill's score read-out (A.8). */
int64_t qo_banded_align(const char* pattern, int plen, const char* text, int tlen,
                        int64_t cutoff_in, char* ops, int64_t* score_out,
                        int64_t* block_advances, int64_t* tb_steps);

/* WindowEd (bpm_windowed.c:563-628).  sse_compat!=0 and W==2 selects the x86
 * SSE kernel's semantics (bpm_windowed.c:283-445).  score-only returns the
 * bound in *score_out and the HEW count; otherwise ops/n_ops (front-to-back). */
int qo_windowed(const char* pattern, int plen, const char* text, int tlen,
                int W, int O, int hew_threshold, int score_only, int sse_compat,
                int64_t* score_out, int64_t* hew_out, char* ops, int64_t* n_ops,
                int64_t* block_steps);

/* Hirschberg (bpm_hirschberg.c:33-270) with a clean midpoint join (SURVEY
 * A.5/A.7(12)).  split_bytes = the footprint threshold (reference: 1<<24). */
int qo_hirschberg(const char* pattern, int plen, const char* text, int tlen,
                  int64_t cutoff_in, uint64_t split_bytes,
                  char* ops, int64_t* n_ops, qo_trace_t* trace);

/* cigar_sprint(print_matches=true) (cigar.c:453-488); buf >= 2*n+10 bytes */
int64_t qo_cigar_rle(const char* ops, int64_t n, char* buf);
/* cigar_score_edit (cigar.c:274-289) */
int64_t qo_cigar_score(const char* ops, int64_t n);
/* alignment validity: ops transform pattern into text (cigar.c:363-434) */
int qo_cigar_check(const char* pattern, int plen, const char* text, int tlen,
                   const char* ops, int64_t n);
/* SAM CIGAR string ("=XID" with show_mismatches, else "MID" with X folded into M; cigar.c:194-240, 504-529) */
int64_t qo_cigar_sam(const char* ops, int64_t n, int show_mismatches, char* buf);
/* parse an RLE string back into ops (cigar.c:252-270); returns count */
int64_t qo_rle_to_ops(const char* rle, char* ops, int64_t max_ops);

/* independent exact edit distance (plain Myers bit-parallel full matrix, not
 * from the reference's hot path) -- second opinion for optimality checks */
int64_t qo_exact_distance(const char* pattern, int plen, const char* text, int tlen);

#ifdef __cplusplus
}
#endif
#endif
