/*
 * quicked_oracle.c -- CPU restatement (ORACLE) of the QuickEd hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see quicked_oracle.h).  Plain C, column-by-column,
 * no SIMD, no arena: the reference's 2/4/8-way skewed schedules are pure ILP
 * scheduling and do not change results, so they are not restated.
 *
 * Citations are file:line in the reference tree (maxdoblas/QuickEd @ 2024-10-22).
 *
 * Behaviours the reference leaves undefined and this file DEFINES:
 *   - bytes >= 0x80 are encoded through an unsigned index (A.7(7));
 *   - band rows >= ceil(plen/64) are never computed (A.7(2));
 *   - a score-only band that never reaches the last pattern block returns -1
 *     instead of uninitialised memory (A.7(3));
 *   - traceback reads outside the stored band see P=0, M=0 (reference: stale
 *     or uninitialised arena words) -- only reachable when cutoff < distance;
 *   - window traceback bit index is (v - v_min) & 63 (A.7(1), x86 shl);
 *   - PEQ blocks past the pattern are zero (A.7(5)); text[tlen] reads as N (A.7(4));
 *   - the Hirschberg midpoint join is re-derived (A.5, A.7(12)): full band
 *     overlap, first minimum, exact child scores;
 *   - QUICKED with only_score returns the edit count of the alignment it
 *     computed (reference: uninitialised cigar_out.score, quicked.c:283-299);
 *   - QUICKED stage 3 with a cutoff of 0 (bandwidth % of a read shorter than
 *     100 / bandwidth rounds to 0): the reference doubles 0 to 0 and never leaves
 *     the loop (quicked.c:248-278); here the doubling starts from 1.
 */
#include "quicked_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define W64 64
#define ALPHA 5
#define ONES (~(uint64_t)0)

static inline int64_t div_ceil(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t imax(int64_t a, int64_t b) { return a >= b ? a : b; }
static inline int64_t imin(int64_t a, int64_t b) { return a <= b ? a : b; }
static inline int64_t iabs(int64_t a) { return a >= 0 ? a : -a; }

/* dna_encode_table (quicked_utils/src/dna_text.c:41-46): A/a 0, C/c 1, G/g 2, T/t 3, else 4 */
static inline int enc(char c) {
    switch ((unsigned char)c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

/* ------------------------------------------------------------------------- */
/* Pattern equality bitmaps (bpm_banded.c:40-103 == bpm_windowed.c:41-122)   */
/* ------------------------------------------------------------------------- */
typedef struct {
    const char* pattern;
    int64_t plen, nw;
    uint64_t* peq;        /* [(nw+2)*5], blocks nw, nw+1 are zero */
    uint64_t* level_mask; /* [nw] */
} pat_t;

static void pat_compile(pat_t* p, const char* pattern, int64_t plen) {
    p->pattern = pattern;
    p->plen = plen;
    p->nw = div_ceil(plen, W64);
    p->peq = (uint64_t*)calloc((size_t)(p->nw + 2) * ALPHA, sizeof(uint64_t));
    p->level_mask = (uint64_t*)calloc((size_t)p->nw + 1, sizeof(uint64_t));
    for (int64_t i = 0; i < plen; ++i)
        p->peq[(i / W64) * ALPHA + enc(pattern[i])] |= (uint64_t)1 << (i % W64);
    /* padding rows of the last block match every symbol (bpm_banded.c:77-86) */
    for (int64_t i = plen; i < p->nw * W64; ++i)
        for (int c = 0; c < ALPHA; ++c)
            p->peq[(i / W64) * ALPHA + c] |= (uint64_t)1 << (i % W64);
    /* level_mask: bit 63, last block bit (plen%64)-1 (bpm_banded.c:88-102) */
    for (int64_t b = 0; b + 1 < p->nw; ++b) p->level_mask[b] = (uint64_t)1 << 63;
    if (p->nw > 0) {
        const int64_t mod = plen % W64;
        p->level_mask[p->nw - 1] = mod ? (uint64_t)1 << (mod - 1) : (uint64_t)1 << 63;
    }
}
static void pat_free(pat_t* p) { free(p->peq); free(p->level_mask); }

/* ------------------------------------------------------------------------- */
/* One 64-row x 1-column Myers/Hyyro step (bpm_commons.h:49-68 / 82-101)     */
/* ------------------------------------------------------------------------- */
static inline void block_step(uint64_t Eq, uint64_t out_mask, uint64_t* Pv, uint64_t* Mv,
                              uint64_t PHin, uint64_t MHin, uint64_t* PHout, uint64_t* MHout) {
    const uint64_t P = *Pv, M = *Mv;
    const uint64_t Xv = Eq | M;
    const uint64_t Eqc = Eq | MHin;
    const uint64_t Xh = (((Eqc & P) + P) ^ P) | Eqc;
    uint64_t Ph = M | ~(Xh | P);
    uint64_t Mh = P & Xh;
    *PHout = (Ph & out_mask) != 0;
    *MHout = (Mh & out_mask) != 0;
    Ph = (Ph << 1) | PHin;
    Mh = (Mh << 1) | MHin;
    *Pv = Mh | ~(Xv | Ph);
    *Mv = Ph & Xv;
}

/* ------------------------------------------------------------------------- */
/* Band geometry (bpm_banded.c:121-135; duplicated bpm_hirschberg.c:46-61)   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int64_t cutoff, diff, rel, prolog, ebb, fin;
} geom_t;

static void band_geometry(int64_t plen, int64_t tlen, int64_t cutoff_in, geom_t* g) {
    const int64_t k_end = iabs(tlen - plen) + 1;
    g->cutoff = imax(imax(k_end, cutoff_in), 65);
    g->diff = plen - tlen;
    g->rel = div_ceil(g->cutoff - iabs(g->diff), 2);
    if (g->diff >= 0) {
        g->prolog = div_ceil(g->rel, W64);
        g->ebb = div_ceil(g->rel + g->diff, W64) + 1 + g->prolog;
    } else {
        g->prolog = div_ceil(g->rel - g->diff, W64);
        g->ebb = div_ceil(g->rel, W64) + 1 + g->prolog;
    }
    g->fin = g->prolog * W64 + g->diff;
}

/* running band: slot i holds absolute block row i + pos_v */
typedef struct {
    int64_t first, last, pos_v, pos_h;
    int64_t nslots;
    int64_t* scores;      /* indexed by absolute block row */
    int64_t nscores;
    int64_t max_row_init; /* highest scores[] index ever written */
    int64_t advances;
} band_t;

static void band_init(band_t* b, const geom_t* g, int64_t nslots, int64_t nw) {
    b->first = g->prolog;
    b->last = nslots - 1;
    b->pos_v = -g->prolog;
    b->pos_h = 0;
    b->nslots = nslots;
    b->nscores = nw + nslots + 2 * g->prolog + 8;
    b->scores = (int64_t*)calloc((size_t)b->nscores, sizeof(int64_t));
    /* bpm_reset_search (bpm_banded.c:180-197) */
    for (int64_t i = 0; i < nslots; ++i) b->scores[i] = W64 * (i + 1);
    b->max_row_init = nslots - 1;
    b->advances = 0;
}

/* one text column over band slots first..last; P/M are the band's slot arrays
 * (bpm_banded.c:318-346 compute_advance_block, 232-262 fill inner loop) */
static inline void band_column(band_t* b, const pat_t* pat, int code, uint64_t* P, uint64_t* M,
                               uint64_t* Pnext, uint64_t* Mnext) {
    uint64_t PHin = 1, MHin = 0, PHout, MHout;
    for (int64_t i = b->first; i <= b->last; ++i) {
        const int64_t r = i + b->pos_v;
        if (r >= pat->nw) break;               /* A.7(2): row past the pattern, never computed */
        uint64_t Pv = P[i], Mv = M[i];
        block_step(pat->peq[r * ALPHA + code], pat->level_mask[r], &Pv, &Mv, PHin, MHin, &PHout, &MHout);
        Pnext[i] = Pv;
        Mnext[i] = Mv;
        PHin = PHout;
        MHin = MHout;
        b->scores[r] += (int64_t)PHout - (int64_t)MHout;
        b->advances++;
    }
}

/* every-64-columns bookkeeping (bpm_banded.c:889-922 score-only, 264-301 fill);
 * P/M = the column the shift applies to; stop_row = nw (score) or nw-1 (fill) */
static void band_chunk_end(band_t* b, const geom_t* g, uint64_t* P, uint64_t* M, int64_t stop_row) {
    int64_t first = b->first, last = b->last;
    const int64_t pos_v = b->pos_v;
    const int cut_lo = (first + 2 < last) && (g->fin > W64 * (first + 1)) &&
                       (b->scores[first + pos_v + 1] + (g->fin - W64 * (first + 1)) > g->cutoff);
    if (cut_lo && b->pos_h >= g->prolog) first++;
    else if (!cut_lo && b->pos_h < g->prolog) first--;
    for (int64_t j = first; j < last; ++j) { P[j] = P[j + 1]; M[j] = M[j + 1]; }
    P[last] = ONES;
    M[last] = 0;
    const int64_t pos = last + pos_v;
    b->scores[pos + 1] = b->scores[pos] + W64;
    if (pos + 1 > b->max_row_init) b->max_row_init = pos + 1;
    const int cut_hi = (first + 2 < last) && (W64 * (last - 1) > g->fin) &&
                       (b->scores[last + pos_v - 1] + (W64 * (last - 1) - g->fin) > g->cutoff);
    if (cut_hi || (pos_v + last >= stop_row)) last--;
    b->first = first;
    b->last = last;
    b->pos_v++;
    b->pos_h++;
}

/* final score read-out (bpm_banded.c:952-961) */
static int64_t band_final_score(const band_t* b, int64_t plen) {
    const int64_t row = (plen % W64) ? plen / W64 : (plen - 1) / W64;
    if (row > b->max_row_init) return -1;          /* A.7(3) */
    return (plen % W64) ? b->scores[row] - (W64 - plen % W64) : b->scores[row];
}

/* ------------------------------------------------------------------------- */
/* BandEd score-only (bpm_banded.c:791-964)                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    geom_t g;
    band_t b;
    uint64_t *P, *M;
    int64_t score;
} score_pass_t;

static void score_pass_run(score_pass_t* sp, const pat_t* pat, const char* text, int64_t tlen,
                           int64_t cutoff_in, int64_t tfin) {
    band_geometry(pat->plen, tlen, cutoff_in, &sp->g);
    /* the score-only kernels use their own, narrower band (bpm_banded.c:801-803) */
    const int64_t ebb_local = div_ceil(sp->g.cutoff, W64) + 1;
    band_init(&sp->b, &sp->g, ebb_local, pat->nw);
    sp->P = (uint64_t*)malloc((size_t)ebb_local * sizeof(uint64_t));
    sp->M = (uint64_t*)malloc((size_t)ebb_local * sizeof(uint64_t));
    for (int64_t i = 0; i < ebb_local; ++i) { sp->P[i] = ONES; sp->M[i] = 0; }
    int64_t col = 0;
    const int64_t chunks = tfin / W64;
    for (int64_t k = 0; k < chunks; ++k) {
        for (int64_t c = 0; c < W64; ++c, ++col)
            band_column(&sp->b, pat, enc(text[col]), sp->P, sp->M, sp->P, sp->M);
        band_chunk_end(&sp->b, &sp->g, sp->P, sp->M, pat->nw);
    }
    for (; col < tfin; ++col)
        band_column(&sp->b, pat, enc(text[col]), sp->P, sp->M, sp->P, sp->M);
    sp->score = band_final_score(&sp->b, pat->plen);
}
static void score_pass_free(score_pass_t* sp) { free(sp->P); free(sp->M); free(sp->b.scores); }

int64_t qo_banded_score(const char* pattern, int plen, const char* text, int tlen,
                        int64_t cutoff_in, int tfin,
                        int64_t* first_out, int64_t* last_out, int64_t* block_advances) {
    pat_t pat;
    pat_compile(&pat, pattern, plen);
    score_pass_t sp;
    score_pass_run(&sp, &pat, text, tlen, cutoff_in, tfin);
    const int64_t score = sp.score;
    if (first_out) *first_out = sp.b.first;
    if (last_out) *last_out = sp.b.last;
    if (block_advances) *block_advances = sp.b.advances;
    score_pass_free(&sp);
    pat_free(&pat);
    return score;
}

/* ------------------------------------------------------------------------- */
/* BandEd fill + traceback (bpm_banded.c:199-316, 967-1036)                  */
/* ------------------------------------------------------------------------- */
int64_t qo_banded_align(const char* pattern, int plen, const char* text, int tlen,
                        int64_t cutoff_in, char* ops, int64_t* score_out,
                        int64_t* block_advances, int64_t* tb_steps) {
    pat_t pat;
    pat_compile(&pat, pattern, plen);
    geom_t g;
    band_geometry(plen, tlen, cutoff_in, &g);
    const int64_t ebb = g.ebb;
    band_t b;
    band_init(&b, &g, ebb, pat.nw);
    /* column-major matrix, stride ebb words (BPM_PATTERN_BDP_IDX, bpm_commons.h:42) */
    uint64_t* Pm = (uint64_t*)calloc((size_t)ebb * (size_t)(tlen + 1), sizeof(uint64_t));
    uint64_t* Mm = (uint64_t*)calloc((size_t)ebb * (size_t)(tlen + 1), sizeof(uint64_t));
    /* stored slot range of every column (defines out-of-band reads, see header comment) */
    const int64_t nchunks = tlen / W64 + 2;
    int64_t* cfirst = (int64_t*)malloc((size_t)nchunks * sizeof(int64_t));
    int64_t* clast = (int64_t*)malloc((size_t)nchunks * sizeof(int64_t));
    for (int64_t i = 0; i < ebb; ++i) { Pm[i] = ONES; Mm[i] = 0; }
    cfirst[0] = b.first;
    clast[0] = b.last;
    for (int64_t col = 0; col < tlen; ++col) {
        uint64_t* Pc = Pm + col * ebb;
        uint64_t* Mc = Mm + col * ebb;
        band_column(&b, &pat, enc(text[col]), Pc, Mc, Pc + ebb, Mc + ebb);
        if ((col + 1) % W64 == 0) {
            band_chunk_end(&b, &g, Pc + ebb, Mc + ebb, pat.nw - 1);
            cfirst[b.pos_h] = b.first;
            clast[b.pos_h] = b.last;
        }
    }
    if (score_out) *score_out = band_final_score(&b, plen);
    if (block_advances) *block_advances = b.advances;

    /* traceback, priority D -> I -> M/X on raw bytes (bpm_banded.c:994-1024) */
    char* rev = (char*)malloc((size_t)plen + (size_t)tlen + 1);
    int64_t n = 0, steps = 0;
    int64_t h = tlen - 1, v = plen - 1;
    while (v >= 0 && h >= 0) {
        const int64_t bh = h / W64, bhr = (h + 1) / W64;
        const int64_t ev = v - W64 * (bh - g.prolog);       /* band-relative row at column h   */
        const int64_t evr = v - W64 * (bhr - g.prolog);     /* ... at column h+1               */
        /* one test decides both reads: block row v/64 must have been computed at column h, i.e. lie
         * in the stored range of column h+1; otherwise P = 0, M = 0 (reference: uninitialised words) */
        uint64_t pbit = 0, mbit = 0;
        {
            const int64_t col = h + 1;
            const int64_t lo = cfirst[col / W64];
            const int64_t hi = (col % W64 == 0) ? clast[col / W64 - 1] : clast[col / W64];
            if (evr >= 0 && ev >= 0) {
                const int64_t slot = evr / W64;
                if (slot >= lo && slot <= hi) {
                    pbit = (Pm[col * ebb + slot] >> (evr % W64)) & 1;       /* Pv at column h+1 */
                    mbit = (Mm[h * ebb + ev / W64] >> (ev % W64)) & 1;      /* Mv at column h   */
                }
            }
        }
        if (pbit) { rev[n++] = 'D'; --v; }
        else if (mbit) { rev[n++] = 'I'; --h; }
        else if (text[h] == pattern[v]) { rev[n++] = 'M'; --h; --v; }
        else { rev[n++] = 'X'; --h; --v; }
        ++steps;
    }
    while (h >= 0) { rev[n++] = 'I'; --h; }
    while (v >= 0) { rev[n++] = 'D'; --v; }
    for (int64_t i = 0; i < n; ++i) ops[i] = rev[n - 1 - i];
    if (tb_steps) *tb_steps = steps;
    free(rev); free(cfirst); free(clast); free(Pm); free(Mm); free(b.scores);
    pat_free(&pat);
    return n;
}

/* ------------------------------------------------------------------------- */
/* WindowEd (bpm_windowed.c:202-280 scalar window, 283-445 SSE window,       */
/*           448-561 in-window tracebacks, 563-628 chain)                    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int W;
    uint64_t *Pv, *Mv;     /* [(64W+2)][W] */
    uint64_t* peqw;        /* [W][5] */
    int64_t pos_v, pos_h, hew, score;
    int64_t block_steps;
} win_t;

static void window_fill(win_t* w, const pat_t* pat, const char* text, int64_t tlen, int sse) {
    const int W = w->W;
    const int64_t v_fi = w->pos_v, h_fi = w->pos_h;
    const int64_t v0 = (v_fi - W64 * W + 1 >= 0) ? v_fi - W64 * W + 1 : 0;
    const int64_t h0 = (h_fi - W64 * W + 1 >= 0) ? h_fi - W64 * W + 1 : 0;
    /* left boundary: real column 0 has D[i][0]=i, otherwise free start (bpm_windowed.c:226-230) */
    for (int i = 0; i < W; ++i) { w->Pv[i] = (h0 == 0) ? ONES : 0; w->Mv[i] = 0; }
    const int64_t steps_v = (v_fi - v0) / W64 + 1;
    const int64_t steps_h = h_fi - h0;
    const int64_t shift = v0 % W64, vb = v0 / W64;
    /* bit-unaligned window rows: funnel-shift two PEQ blocks (bpm_windowed.c:237-244) */
    for (int64_t i = 0; i < steps_v; ++i)
        for (int c = 0; c < ALPHA; ++c) {
            const uint64_t a = pat->peq[(i + vb) * ALPHA + c] >> shift;
            const uint64_t bq = shift ? pat->peq[(i + vb + 1) * ALPHA + c] << (W64 - shift) : 0;
            w->peqw[i * ALPHA + c] = a | bq;
        }
    const uint64_t ph_first = (v0 == 0) ? 1 : 0;
    if (!sse) {
        for (int64_t t = 0; t <= steps_h; ++t) {
            const int code = enc(text[t + h0]);
            uint64_t PHin = ph_first, MHin = 0, PHout, MHout;
            for (int64_t i = 0; i < steps_v; ++i) {
                uint64_t P = w->Pv[t * W + i], M = w->Mv[t * W + i];
                block_step(w->peqw[i * ALPHA + code], (uint64_t)1 << 63, &P, &M, PHin, MHin, &PHout, &MHout);
                w->Pv[(t + 1) * W + i] = P;
                w->Mv[(t + 1) * W + i] = M;
                PHin = PHout;
                MHin = MHout;
                w->block_steps++;
            }
        }
        return;
    }
    /* x86 SSE kernel semantics, W == 2: block 0's top carry is
     * ph_first, 1, then 1/0 alternating on even/odd columns; when steps_h is
     * odd, block 0 runs one extra column and block 1's last column is redone
     * with the carries of that extra column. */
    const int64_t ncol0 = (steps_h % 2 == 1) ? steps_h + 2 : steps_h + 1;   /* columns of block 0 */
    uint64_t* ph0 = (uint64_t*)malloc((size_t)ncol0 * sizeof(uint64_t));
    uint64_t* mh0 = (uint64_t*)malloc((size_t)ncol0 * sizeof(uint64_t));
    for (int64_t t = 0; t < ncol0; ++t) {
        const int64_t tp = t + h0;
        const int code = (tp < tlen) ? enc(text[tp]) : 4;                    /* A.7(4) */
        const uint64_t PHin = (t == 0) ? ph_first : (t == 1) ? 1 : (uint64_t)((t % 2) == 0);
        uint64_t P = w->Pv[t * W + 0], M = w->Mv[t * W + 0];
        block_step(w->peqw[0 * ALPHA + code], (uint64_t)1 << 63, &P, &M, PHin, 0, &ph0[t], &mh0[t]);
        w->Pv[(t + 1) * W + 0] = P;
        w->Mv[(t + 1) * W + 0] = M;
        if (t <= steps_h) w->block_steps++;         /* work unit = window cells; the extra column is not one */
    }
    if (steps_v > 1) {
        for (int64_t t = 0; t <= steps_h; ++t) {
            const int code = enc(text[t + h0]);
            uint64_t PHin = ph0[t], MHin = mh0[t], PHout, MHout;
            if (t == steps_h && (steps_h % 2 == 1)) { PHin = ph0[t + 1]; MHin = mh0[t + 1]; }
            uint64_t P = w->Pv[t * W + 1], M = w->Mv[t * W + 1];
            block_step(w->peqw[1 * ALPHA + code], (uint64_t)1 << 63, &P, &M, PHin, MHin, &PHout, &MHout);
            w->Pv[(t + 1) * W + 1] = P;
            w->Mv[(t + 1) * W + 1] = M;
            w->block_steps++;
        }
    }
    free(ph0);
    free(mh0);
}

/* in-window traceback; score-only priority D -> I -> match -> X (bpm_windowed.c:527-549),
 * CIGAR priority match -> D -> I -> X (476-495).  ops written back-to-front via *n. */
static void window_traceback(win_t* w, const pat_t* pat, const char* text, int O, int hew_threshold,
                             int score_only, char* rev_ops, int64_t* n) {
    const int W = w->W;
    int64_t h = w->pos_h, v = w->pos_v;
    const int64_t h_min = (w->pos_h - W64 * W + 1 > 0) ? w->pos_h - W64 * W + 1 : 0;
    const int64_t h_ov = (w->pos_h - W64 * (W - O) + 1 > 0) ? w->pos_h - W64 * (W - O) + 1 : 0;
    const int64_t v_min = (w->pos_v - W64 * W + 1 > 0) ? w->pos_v - W64 * W + 1 : 0;
    const int64_t v_ov = (w->pos_v - W64 * (W - O) + 1 > 0) ? w->pos_v - W64 * (W - O) + 1 : 0;
    const char* pattern = pat->pattern;
    int64_t score = 0;
    while (v >= v_ov && h >= h_ov) {
        const int64_t block = ((v - v_min) / W64) & 0xff;                /* uint8_t block */
        const int64_t idx = (h - h_min + 1) * W + block;
        const uint64_t mask = (uint64_t)1 << ((v - v_min) & 63);         /* A.7(1) */
        const int p = (w->Pv[idx] & mask) != 0;
        const int m = (w->Mv[idx - W] & mask) != 0;
        const int eq = text[h] == pattern[v];
        if (score_only) {
            if (p) { score++; --v; }
            else if (m) { score++; --h; }
            else if (eq) { --h; --v; }
            else { score++; --h; --v; }
        } else {
            if (eq) { rev_ops[(*n)++] = 'M'; --h; --v; }
            else if (p) { rev_ops[(*n)++] = 'D'; --v; }
            else if (m) { rev_ops[(*n)++] = 'I'; --h; }
            else { rev_ops[(*n)++] = 'X'; --h; --v; }
        }
    }
    if (score_only) {
        if (score > (int64_t)((W - O) * W64 * hew_threshold / 100)) w->hew++;
        w->score += score;
    }
    w->pos_h = h;
    w->pos_v = v;
}

int qo_windowed(const char* pattern, int plen, const char* text, int tlen,
                int W, int O, int hew_threshold, int score_only, int sse_compat,
                int64_t* score_out, int64_t* hew_out, char* ops, int64_t* n_ops,
                int64_t* block_steps) {
    pat_t pat;
    pat_compile(&pat, pattern, plen);
    win_t w;
    w.W = W;
    w.Pv = (uint64_t*)calloc((size_t)W * (W64 * W + 3), sizeof(uint64_t));
    w.Mv = (uint64_t*)calloc((size_t)W * (W64 * W + 3), sizeof(uint64_t));
    w.peqw = (uint64_t*)calloc((size_t)W * ALPHA, sizeof(uint64_t));
    w.pos_v = plen - 1;
    w.pos_h = tlen - 1;
    w.hew = 0;
    w.score = 0;
    w.block_steps = 0;
    char* rev = score_only ? NULL : (char*)malloc((size_t)plen + (size_t)tlen + 1);
    int64_t n = 0;
    const int sse = sse_compat && W == 2;                 /* bpm_windowed.c:577 */
    while (w.pos_v >= 0 && w.pos_h >= 0) {
        window_fill(&w, &pat, text, tlen, sse);
        window_traceback(&w, &pat, text, O, hew_threshold, score_only, rev, &n);
    }
    if (score_only) {
        if (w.pos_h >= 0) w.score += w.pos_h + 1;
        if (w.pos_v >= 0) w.score += w.pos_v + 1;
    } else {
        for (int64_t h = w.pos_h; h >= 0; --h) rev[n++] = 'I';
        for (int64_t v = w.pos_v; v >= 0; --v) rev[n++] = 'D';
        for (int64_t i = 0; i < n; ++i) ops[i] = rev[n - 1 - i];
        if (n_ops) *n_ops = n;
        free(rev);
    }
    if (score_out) *score_out = w.score;
    if (hew_out) *hew_out = w.hew;
    if (block_steps) *block_steps += w.block_steps;           /* accumulates: QuickEd counts stage 1 and both stage-2 passes */
    free(w.Pv); free(w.Mv); free(w.peqw);
    pat_free(&pat);
    return QO_WIP;
}

/* ------------------------------------------------------------------------- */
/* Hirschberg (bpm_hirschberg.c:33-270), clean join                          */
/* ------------------------------------------------------------------------- */
static void reverse_copy(const char* in, char* out, int64_t n) {   /* commons.c:82-88 */
    for (int64_t i = 0; i < n; ++i) out[n - 1 - i] = in[i];
}

/* D[i][column] for the pattern prefixes i the stopped band covers; dist[i] = -1 elsewhere.
 * scores[r] is D at the bottom row of block r (A.8); rows inside a block follow
 * from the P/M vertical deltas.  Row 64*r_first (the band's assumed +1 top
 * boundary) is only exact -- and only reported -- when it is matrix row 0. */
static void band_column_distances(const score_pass_t* sp, const pat_t* pat, int64_t column, int64_t* dist) {
    const int64_t m = pat->plen;
    for (int64_t i = 0; i <= m; ++i) dist[i] = -1;
    const band_t* b = &sp->b;
    for (int64_t s = b->first; s <= b->last; ++s) {
        const int64_t r = s + b->pos_v;
        if (r < 0 || r >= pat->nw || r > b->max_row_init) continue;
        const int64_t bottom = imin(W64 * (r + 1), m);          /* pattern prefix length at block bottom */
        int64_t d = (bottom == W64 * (r + 1)) ? b->scores[r] : b->scores[r] - (W64 * (r + 1) - m);
        for (int64_t i = bottom; i > W64 * r; --i) {
            dist[i] = d;
            const int bit = (int)((i - 1) % W64);
            d -= (int64_t)((sp->P[s] >> bit) & 1) - (int64_t)((sp->M[s] >> bit) & 1);
        }
        if (r == 0) dist[0] = column;
    }
}

static int hirschberg_rec(const char* pattern, int64_t plen, const char* text, int64_t tlen,
                          int64_t cutoff_in, uint64_t split_bytes, char* ops, int64_t* n, qo_trace_t* tr);

static int hirschberg_leaf(const char* pattern, int64_t plen, const char* text, int64_t tlen,
                           int64_t cutoff_in, char* ops, int64_t* n, qo_trace_t* tr) {
    if (plen == 0) { for (int64_t i = 0; i < tlen; ++i) ops[(*n)++] = 'I'; return QO_OK; }
    if (tlen == 0) { for (int64_t i = 0; i < plen; ++i) ops[(*n)++] = 'D'; return QO_OK; }
    int64_t adv = 0, steps = 0;
    *n += qo_banded_align(pattern, (int)plen, text, (int)tlen, cutoff_in, ops + *n, NULL, &adv, &steps);
    if (tr) { tr->leaves++; tr->fill_block_advances += adv; tr->traceback_steps += steps; }
    return QO_OK;
}

static int hirschberg_rec(const char* pattern, int64_t plen, const char* text, int64_t tlen,
                          int64_t cutoff_in, uint64_t split_bytes, char* ops, int64_t* n, qo_trace_t* tr) {
    if (plen == 0 || tlen == 0) return hirschberg_leaf(pattern, plen, text, tlen, cutoff_in, ops, n, tr);
    geom_t g;
    band_geometry(plen, tlen, cutoff_in, &g);
    const uint64_t footprint = (uint64_t)g.ebb * (uint64_t)tlen * 8u * 2u;   /* bpm_hirschberg.c:63 */
    if (footprint <= split_bytes) return hirschberg_leaf(pattern, plen, text, tlen, cutoff_in, ops, n, tr);

    const int64_t n1 = (tlen + 1) / 2, n2 = tlen - n1;                         /* bpm_hirschberg.c:68-69 */
    char* pattern_r = (char*)malloc((size_t)plen);
    char* text_r = (char*)malloc((size_t)tlen);
    reverse_copy(pattern, pattern_r, plen);
    reverse_copy(text, text_r, tlen);
    pat_t pf, pr;
    pat_compile(&pf, pattern, plen);
    pat_compile(&pr, pattern_r, plen);
    score_pass_t f, r;
    /* both half passes use the FULL (plen, tlen, cutoff) geometry and stop at their half (85-100) */
    score_pass_run(&f, &pf, text, tlen, cutoff_in, n1);
    score_pass_run(&r, &pr, text_r, tlen, cutoff_in, n2);
    if (tr) { tr->hirschberg_splits++; tr->score_block_advances += f.b.advances + r.b.advances; }
    int64_t* df = (int64_t*)malloc((size_t)(plen + 1) * sizeof(int64_t));
    int64_t* dr = (int64_t*)malloc((size_t)(plen + 1) * sizeof(int64_t));
    band_column_distances(&f, &pf, n1, df);
    band_column_distances(&r, &pr, n2, dr);
    int64_t best = -1, best_i = -1;
    for (int64_t i = 0; i <= plen; ++i) {
        if (df[i] < 0 || dr[plen - i] < 0) continue;
        const int64_t s = df[i] + dr[plen - i];
        if (best < 0 || s < best) { best = s; best_i = i; }
    }
    int64_t score_l = 0, score_r = 0;
    if (best_i >= 0) { score_l = df[best_i]; score_r = dr[plen - best_i]; }
    free(df); free(dr);
    score_pass_free(&f); score_pass_free(&r);
    pat_free(&pf); pat_free(&pr);
    free(pattern_r); free(text_r);
    if (best_i < 0) return QO_FAIL_NON_CONVERGENCE;                           /* bpm_hirschberg.c:116-122 */
    /* ops are produced front-to-back here, so left child first (the reference
     * prepends into a back-to-front buffer, right child first: 212-243) */
    int st = hirschberg_rec(pattern, best_i, text, n1, score_l, split_bytes, ops, n, tr);
    if (st < 0) return st;
    return hirschberg_rec(pattern + best_i, plen - best_i, text + n1, n2, score_r, split_bytes, ops, n, tr);
}

int qo_hirschberg(const char* pattern, int plen, const char* text, int tlen,
                  int64_t cutoff_in, uint64_t split_bytes,
                  char* ops, int64_t* n_ops, qo_trace_t* trace) {
    int64_t n = 0;
    const int st = hirschberg_rec(pattern, plen, text, tlen, cutoff_in, split_bytes, ops, &n, trace);
    if (n_ops) *n_ops = n;
    return st;
}

/* ------------------------------------------------------------------------- */
/* CIGAR helpers (quicked_utils/src/cigar.c)                                 */
/* ------------------------------------------------------------------------- */
int64_t qo_cigar_rle(const char* ops, int64_t n, char* buf) {          /* cigar.c:453-488 */
    int64_t cur = 0;
    if (n <= 0) { buf[0] = '\0'; return 0; }
    char last = ops[0];
    int64_t len = 1;
    for (int64_t i = 1; i < n; ++i) {
        if (ops[i] == last) { ++len; continue; }
        cur += sprintf(buf + cur, "%d%c", (int)len, last);
        last = ops[i];
        len = 1;
    }
    cur += sprintf(buf + cur, "%d%c", (int)len, last);
    buf[cur] = '\0';
    return cur;
}

int64_t qo_cigar_score(const char* ops, int64_t n) {                   /* cigar.c:274-289 */
    int64_t s = 0;
    for (int64_t i = 0; i < n; ++i) s += (ops[i] != 'M');
    return s;
}

int qo_cigar_check(const char* pattern, int plen, const char* text, int tlen,
                   const char* ops, int64_t n) {                       /* cigar.c:363-434 */
    int64_t v = 0, h = 0;
    for (int64_t i = 0; i < n; ++i) {
        switch (ops[i]) {
            case 'M': if (v >= plen || h >= tlen || pattern[v] != text[h]) return 0; ++v; ++h; break;
            case 'X': if (v >= plen || h >= tlen || pattern[v] == text[h]) return 0; ++v; ++h; break;
            case 'I': if (h >= tlen) return 0; ++h; break;
            case 'D': if (v >= plen) return 0; ++v; break;
            default: return 0;
        }
    }
    return v == plen && h == tlen;
}

/* SAM CIGAR of an operations string: cigar_compute_CIGAR (cigar.c:194-240) + cigar_sprint_SAM_CIGAR
 * (cigar.c:504-529: "%d%c" over "MIDN---=X").  With show_mismatches M prints as '='; without it every X
 * becomes M before equal neighbours are merged -- except the very first operation, which the reference
 * reads before its mapping step (cigar.c:211 vs 217): a leading X is emitted as "1X" and only the
 * operations after it fold.  Kept as is: the compiled reference pins it.  buf >= 2*n+10 bytes. */
int64_t qo_cigar_sam(const char* ops, int64_t n, int show_mismatches, char* buf) {
    int64_t cur = 0;
    buf[0] = '\0';
    if (n <= 0) return 0;
    char last = ops[0];
    int64_t len = 1;
    for (int64_t i = 1; i <= n; ++i) {
        char op = 0;
        if (i < n) { op = ops[i]; if (!show_mismatches && op == 'X') op = 'M'; }
        if (i < n && op == last) { ++len; continue; }
        cur += sprintf(buf + cur, "%lld%c", (long long)len, (show_mismatches && last == 'M') ? '=' : last);
        last = op; len = 1;
    }
    return cur;
}

int64_t qo_rle_to_ops(const char* rle, char* ops, int64_t max_ops) {   /* cigar.c:252-270 */
    int64_t n = 0, num = 0;
    for (const char* p = rle; *p; ++p) {
        if (*p >= '0' && *p <= '9') { num = num * 10 + (*p - '0'); continue; }
        for (int64_t j = 0; j < num; ++j) { if (n >= max_ops) return -1; ops[n++] = *p; }
        num = 0;
    }
    return n;
}

/* plain full-height Myers, no band, no cuts: independent of the band geometry */
int64_t qo_exact_distance(const char* pattern, int plen, const char* text, int tlen) {
    if (plen == 0) return tlen;
    if (tlen == 0) return plen;
    pat_t pat;
    pat_compile(&pat, pattern, plen);
    const int64_t nw = pat.nw;
    uint64_t* P = (uint64_t*)malloc((size_t)nw * sizeof(uint64_t));
    uint64_t* M = (uint64_t*)calloc((size_t)nw, sizeof(uint64_t));
    for (int64_t i = 0; i < nw; ++i) P[i] = ONES;
    int64_t score = plen;
    for (int64_t col = 0; col < tlen; ++col) {
        const int code = enc(text[col]);
        uint64_t PHin = 1, MHin = 0, PHout = 0, MHout = 0;
        for (int64_t r = 0; r < nw; ++r) {
            block_step(pat.peq[r * ALPHA + code], pat.level_mask[r], &P[r], &M[r], PHin, MHin, &PHout, &MHout);
            PHin = PHout;
            MHin = MHout;
        }
        score += (int64_t)PHout - (int64_t)MHout;
    }
    free(P); free(M);
    pat_free(&pat);
    return score;
}

/* ------------------------------------------------------------------------- */
/* Drivers (quicked.c)                                                       */
/* ------------------------------------------------------------------------- */
void qo_default_params(qo_params_t* p) {                               /* quicked.c:308-321 */
    memset(p, 0, sizeof(*p));
    p->algo = QO_QUICKED;
    p->bandwidth = 15;
    p->window_size = 9;
    p->overlap_size = 1;
    p->hew_threshold[0] = p->hew_threshold[1] = 40;
    p->hew_percentage[0] = p->hew_percentage[1] = 15;
}

const char* qo_status_msg(int status) {                                /* quicked.c:382-403 */
    switch (status) {
        case QO_ERROR: return "ERROR: QuickEd has finished with unspecific error\n";
        case QO_FAIL_NON_CONVERGENCE: return "ERROR: Hirschberg algorithm can not find a middle point of subsequence division!\n";
        case QO_UNIMPLEMENTED: return "ERROR: The algorithm or parameter combination selected is not implemented\n";
        case QO_UNKNOWN_ALGO: return "ERROR: Unknown algorithm selection\n";
        case QO_EMPTY_SEQUENCE: return "ERROR: Tried to align an empty sequence\n";
        default: return "QuickEd finished without errors.\n";
    }
}

void qo_free(void* p) { free(p); }

/* extract_results (quicked.c:34-56) for a CIGAR-producing run */
static void results_from_ops(const qo_params_t* p, const char* ops, int64_t n, int* score_out, char** cigar_out) {
    *score_out = (int)qo_cigar_score(ops, n);
    if (!p->only_score && n > 0 && cigar_out) {
        *cigar_out = (char*)malloc((size_t)(2 * n + 10));
        qo_cigar_rle(ops, n, *cigar_out);
    }
}

int qo_align(const qo_params_t* p, const char* pattern, int plen, const char* text, int tlen,
             int* score_out, char** cigar_out, qo_trace_t* trace) {
    qo_trace_t local;
    qo_trace_t* tr = trace ? trace : &local;
    memset(tr, 0, sizeof(*tr));
    if (cigar_out) *cigar_out = NULL;
    if (plen == 0 || tlen == 0) return QO_EMPTY_SEQUENCE;               /* quicked.c:411-414 */
    const int64_t max_len = imax(tlen, plen);
    const int sse = !p->force_scalar;
    switch (p->algo) {
    case QO_BANDED: {                                                   /* run_banded, quicked.c:58-89 */
        const int64_t cutoff = (int64_t)(((uint32_t)max_len * p->bandwidth) / 100u);
        if (p->only_score) {
            int64_t adv = 0;
            *score_out = (int)qo_banded_score(pattern, plen, text, tlen, cutoff, tlen, NULL, NULL, &adv);
            tr->score_block_advances = adv;
        } else {
            char* ops = (char*)malloc((size_t)plen + (size_t)tlen + 1);
            const int64_t n = qo_banded_align(pattern, plen, text, tlen, cutoff, ops, NULL,
                                              &tr->fill_block_advances, &tr->traceback_steps);
            results_from_ops(p, ops, n, score_out, cigar_out);
            free(ops);
        }
        return QO_WIP;
    }
    case QO_WINDOWED: {                                                 /* run_windowed, quicked.c:91-123 */
        int64_t score = 0, hew = 0, n = 0;
        if (p->only_score) {
            qo_windowed(pattern, plen, text, tlen, (int)p->window_size, (int)p->overlap_size, 0, 1, sse,
                        &score, &hew, NULL, NULL, &tr->window_block_steps);
            *score_out = (int)score;
        } else {
            char* ops = (char*)malloc((size_t)plen + (size_t)tlen + 1);
            qo_windowed(pattern, plen, text, tlen, (int)p->window_size, (int)p->overlap_size, 0, 0, sse,
                        &score, &hew, ops, &n, &tr->window_block_steps);
            results_from_ops(p, ops, n, score_out, cigar_out);
            free(ops);
        }
        return QO_WIP;
    }
    case QO_HIRSCHBERG: {                                               /* run_hirschberg, quicked.c:125-161 */
        const int64_t cutoff = (int64_t)(((uint32_t)max_len * p->bandwidth) / 100u);
        char* ops = (char*)malloc((size_t)plen + (size_t)tlen + 1);
        int64_t n = 0;
        const int st = qo_hirschberg(pattern, plen, text, tlen, cutoff, (uint64_t)1 << 24, ops, &n, tr);
        results_from_ops(p, ops, n, score_out, cigar_out);
        free(ops);
        return st;
    }
    case QO_QUICKED: {                                                  /* run_quicked, quicked.c:163-306 */
        int64_t score = 0, hew = 0;
        /* stage 1: WindowEd(2,1) score-only (178-199) */
        qo_windowed(pattern, plen, text, tlen, 2, 1, (int)p->hew_threshold[0], 1, sse,
                    &score, &hew, NULL, NULL, &tr->window_block_steps);
        tr->ws_score = score; tr->ws_hew = hew; tr->stage = 1;
        if ((uint64_t)hew * 64u > (uint64_t)((uint32_t)max_len * p->hew_percentage[0] / 100u)) {
            /* stage 2: WindowEd(W,O) forward and on the reversed strings (204-235) */
            char* pattern_r = (char*)malloc((size_t)plen);
            char* text_r = (char*)malloc((size_t)tlen);
            reverse_copy(pattern, pattern_r, plen);
            reverse_copy(text, text_r, tlen);
            int64_t s_f = 0, h_f = 0, s_r = 0, h_r = 0;
            qo_windowed(pattern, plen, text, tlen, (int)p->window_size, (int)p->overlap_size,
                        (int)p->hew_threshold[1], 1, sse, &s_f, &h_f, NULL, NULL, &tr->window_block_steps);
            qo_windowed(pattern_r, plen, text_r, tlen, (int)p->window_size, (int)p->overlap_size,
                        (int)p->hew_threshold[1], 1, sse, &s_r, &h_r, NULL, NULL, &tr->window_block_steps);
            free(pattern_r); free(text_r);
            score = imin(s_f, s_r);
            hew = (score >= s_r) ? h_r : h_f;                           /* 229-230 */
            tr->wl_fwd_score = s_f; tr->wl_rev_score = s_r;
            tr->wl_score = score; tr->wl_hew = hew; tr->stage = 2;
            if ((uint64_t)hew * 64u * (uint64_t)(p->window_size - p->overlap_size) >
                (uint64_t)((uint32_t)max_len * p->hew_percentage[1] / 100u)) {
                /* stage 3: score-only BandEd with band doubling (240-280) */
                tr->stage = 3;
                score = imin((int64_t)((uint32_t)max_len * p->bandwidth / 100u), score);
                int64_t adv = 0;
                int64_t ns = qo_banded_score(pattern, plen, text, tlen, score, tlen, NULL, NULL, &adv);
                tr->score_block_advances += adv; tr->banded_calls++;
                while ((ns > max_len / 4 && score * 3 / 2 < ns) || ns < 0) {
                    /* the reference doubles a cutoff of 0 (bandwidth % of a short read rounds to 0) to 0 forever;
                     * defined behaviour here and in the kernels' driver: doubling starts from 1 */
                    score = score * 2 > 1 ? score * 2 : 1;
                    ns = qo_banded_score(pattern, plen, text, tlen, score, tlen, NULL, NULL, &adv);
                    tr->score_block_advances += adv; tr->banded_calls++;
                }
                score = ns;
            }
        }
        tr->bound = score;
        /* align: Hirschberg with the bound as cutoff; status ignored (283-294, A.7(8)) */
        char* ops = (char*)malloc((size_t)plen + (size_t)tlen + 1);
        int64_t n = 0;
        qo_hirschberg(pattern, plen, text, tlen, score, (uint64_t)1 << 24, ops, &n, tr);
        results_from_ops(p, ops, n, score_out, cigar_out);
        free(ops);
        return QO_WIP;
    }
    default:
        return QO_UNKNOWN_ALGO;
    }
}
