// qe_types.h -- argument blocks shared by the host driver and the gfx950 kernels.
//
// Data model (DESIGN.md "Data layout in HBM"):
//   pair    one (pattern, text) of the batch; owns ASCII bytes and bit-planes.
//   planes  per pair and per sequence, 3 x u64 per 64 bases, interleaved
//           [row][plane]: plane 0/1 = the two bits of the base code, plane 2 =
//           "not ACGT" (the reference's code 4, dna_text.c:41-46).  Two zero
//           rows of padding follow every sequence (funnel-shift over-read).
//   task    one unit of kernel work = a sub-rectangle of a pair:
//           pattern[p0, p0+m) x text[t0, t0+n) with a cutoff.  Whole-pair
//           alignments are tasks with p0 = t0 = 0; Hirschberg children are not.
//   group   64 consecutive tasks = one wavefront; lane l of group g owns task
//           64 g + l for the whole kernel (one lane per alignment, no cross-lane
//           traffic; all per-task state is addressed [..][lane] so that every
//           global access of a wave is one contiguous 256/512/1024-byte row).
#pragma once
#include <stdint.h>

namespace qe {

typedef uint64_t u64;
typedef uint32_t u32;

enum : u32 { FLAG_HAS_N = 1u, FLAG_NONCANON = 2u };
// The BandEd fill leaves a checkpoint {Pv, Mv} every QE_CP_COLS columns of every band slot, QE_CPC per 64-column chunk: the
// width of the tile the traceback rebuilds in one round (k_traceback's TW).  Round 2 stored one every 8 columns and the
// traceback read every other one.
enum : int { QE_CP_COLS = 16, QE_CPC = 64 / QE_CP_COLS };
enum : u32 { OP_M = 0, OP_X = 1, OP_I = 2, OP_D = 3 };

struct PairView {
    const uint8_t* asc_p;  const int64_t* asc_p_off;  const int32_t* p_len;
    const uint8_t* asc_t;  const int64_t* asc_t_off;  const int32_t* t_len;
    const u64* pl_p;  const int64_t* pl_p_off;     // forward planes, word offsets
    const u64* pl_t;  const int64_t* pl_t_off;
    const u32* flags;
};

struct TaskView {
    int32_t ntasks;
    const int32_t* pair;     // -1 = empty slot
    const int32_t* p0;  const int32_t* m;
    const int32_t* t0;  const int32_t* n;
    const int32_t* cutoff;   // cutoff_in of banded_matrix_allocate
    const int32_t* tfin;     // text_finish_pos (score-only passes)
};

// pack: ASCII -> planes (+ flags).  reverse != 0 packs the reversed string.
struct PackArgs {
    int32_t nseq;
    const uint8_t* asc;  const int64_t* asc_off;  const int32_t* len;
    u64* planes;  const int64_t* pl_off;
    u32* flags;          // OR-ed into (may be null)
    int32_t reverse;
};

// packed wire format -> planes (quicked_batch_create_packed).  wire 2: 2 bits per base, base i of a sequence in bits
// 2(i%32).. of word i/32, codes A0 C1 G2 T3 (dna_text.c:41-46), no N; wire 3: the plane layout itself, [row][3] =
// {code bit 0, code bit 1, not-ACGT} per 64 bases, without the padding rows
struct WireArgs {
    int32_t nseq, wire;
    const u64* words;  const int64_t* w_off;  const int32_t* len;
    u64* planes;  const int64_t* pl_off;
    u32* flags;          // FLAG_HAS_N is OR-ed in (may be null)
};
// planes of the reversed sequences from the forward planes (what k_pack(reverse) makes from ASCII)
struct RevArgs {
    int32_t nseq;
    const u64* fwd;  u64* rev;  const int64_t* pl_off;  const int32_t* len;
};

// BandEd, score-only or fill (bpm_banded.c:791-964 / 199-316)
struct BandedArgs {
    PairView P;
    TaskView T;
    // per-group workspace: Pv[(ns+1)][64] u64 | Mv[(ns+1)][64] u64 | S[nrows][64] i32 | cf[nch][64] i16 | cl[nch][64] i16
    uint8_t* ws;  const int64_t* g_ws_off;  const int32_t* g_nslots;  const int32_t* g_nrows;  const int32_t* g_nch;
    // fill only: checkpoints {Pv,Mv} every QE_CP_COLS columns [(QE_CPC chunk + j) * g_nslots + slot][64] x 16 B, then the carry-in
    // words of every (chunk, slot) [(chunk * g_nslots + slot)][64] x 16 B
    uint4* mat;  const int64_t* g_mat_off;
    // outputs per task
    int32_t* o_score;  int32_t* o_first;  int32_t* o_last;  int32_t* o_posv;  u32* o_adv;  int32_t* o_maxrow;
    const int32_t* only_if;  // run only tasks whose flag is non-zero (fallback pass after k_banded_coop); may be null
    int32_t fill_multi = 1;  // fill: K-slot skewed passes where no lane needs the general form (0: single-slot passes only; tests)
    int32_t lane_rel = 1;    // every lane walks ITS band (slot first + j at step j of a chunk) instead of the wave walking the union of its lanes' bands (0: tests)
    int32_t* o_abort = nullptr;   // k_banded_sys: 1 where a task is left to k_banded<true> (N in the pair, a band of more than 15 slots)
    int32_t doubling = 0;         // k_banded_sys<.., false>: QuickEd's stage-3 band doubling in the launch (quicked.c:248-278)
    int32_t* o_cutoff = nullptr;  // ... the cutoff of every task's last pass (or, flagged, of the pass it was handed back before)
    int32_t prio = 0;             // the cooperative forms: s_setprio 3 (few waves on a serial chain, next to chip-filling launches)
    int32_t fill_geom = 0;        // k_banded<false>: the FILL's band geometry (a6's ebb, stop rule nw - 1: bpm_banded.c:121-135, 295) instead of
                                  // the score-only kernel's narrower one (801-803, 917): the cells, hence the end value, of the fill -- QuickEd with
                                  // only_score takes its score from such a pass instead of filling, tracing back and counting edits
};

// BandEd score-only, G lanes per alignment (cooperative form of k_banded<false>)
struct CoopArgs {
    PairView P;
    TaskView T;
    int32_t G;                               // lanes per alignment: 2, 4, ..., 64; a wave owns 64 / G tasks
    // per-WAVE workspace (A = 64 / G): Pv[(ns+1)][A] u64 | Mv[(ns+1)][A] u64 | S[nrows][A] i32 |
    //                                  CF[nch][A] i16 | CL[nch][A] i16 | KF[A] i32 | KL[A] i32
    uint8_t* ws;  const int64_t* w_ws_off;  const int32_t* w_nslots;  const int32_t* w_nrows;  const int32_t* w_nch;
    int32_t* o_score;  int32_t* o_first;  int32_t* o_last;  int32_t* o_posv;  u32* o_adv;  int32_t* o_maxrow;
    int32_t* o_abort;                        // 1: a band-edge decision could not be resolved in time; rerun with k_banded<false>
    int32_t fill_geom = 0;                   // k_banded_coop_lds<false>: the fill's band geometry (BandedArgs::fill_geom)
};

// the same with the band state of a wave's tasks in LDS (k_banded_coop_lds): what the whole launch is sized for
struct CoopLdsArgs {
    CoopArgs A;
    int32_t lgG;             // log2(A.G)
    int32_t ns;              // band slots per task the LDS holds (>= every task's band height)
    int32_t rr;              // scores[] ring: ns + G + 4 block rows per task and parity
    int32_t cr;              // band-edge rings: a power of two >= G + 4 chunks
    int32_t lds_per_wave;    // bytes
    // FILL form only (k_banded_coop_lds<true>): where the traceback expects the fill's checkpoints, carry words and band
    // edges -- BandedArgs' per-GROUP layout (64 tasks per group, column = task & 63); a wave's 64 / G tasks lie in one group
    uint4* mat;  const int64_t* g_mat_off;
    uint8_t* gws;  const int64_t* g_ws_off;  const int32_t* g_nslots;  const int32_t* g_nrows;  const int32_t* g_nch;
};

// BandEd traceback over a filled matrix (bpm_banded.c:967-1036) -> RLE runs, back to front
struct TraceArgs {
    PairView P;
    TaskView T;
    const uint8_t* ws;  const int64_t* g_ws_off;  const int32_t* g_nslots;  const int32_t* g_nrows;  const int32_t* g_nch;
    const uint4* mat;  const int64_t* g_mat_off;
    u32* runs;  const int64_t* g_runs_off;  const int32_t* g_runs_cap;   // [idx][64] u32 = len << 2 | op
    int32_t* o_nruns;  int32_t* o_nops;  int32_t* o_edits;  u32* o_steps;
    int32_t runs_by_task;    // != 0: task (g, lane) has the stretch [lane * cap, (lane + 1) * cap) of its group's buffer to itself
                             // (the wave-per-alignment formatter reads it sequentially; [idx][lane] rows cost it a 256-byte row per run)
    const int32_t* only_if = nullptr;   // k_traceback: walk only tasks whose flag is non-zero (what k_traceback_sys left)
    int32_t* o_abort = nullptr;         // k_traceback_sys: 1 where a task is left to k_traceback (N / non-canonical symbols)
    int32_t prio = 0;                   // k_traceback_sys: s_setprio 3
};

// WindowEd chain (bpm_windowed.c:563-628)
struct WindowArgs {
    PairView P;
    TaskView T;
    int32_t W, O, hew_threshold, score_only, sse, reversed;
    int32_t cp_path;       // k_windowed_cp: checkpoints + carry words instead of the full history (0: the history path, for tests)
    // per-group workspace: Pv[W][64] u64 | Mv[W][64] u64 ; history {Pv,Mv}[(64W+3)*W][64] x 16 B (k_windowed_cp uses a prefix:
    // checkpoints [8W][W][64] + carry words [W][W][64], 16 B each)
    uint8_t* ws;  const int64_t* g_ws_off;
    u32* runs;  const int64_t* g_runs_off;  const int32_t* g_runs_cap;
    int32_t* o_score;  int32_t* o_hew;  int32_t* o_nruns;  int32_t* o_nops;  int32_t* o_edits;  u32* o_steps;
    // chain state [5][ntasks] = {pos_v, pos_h, score, hew, steps} per task: k_windowed_quad (four lanes per alignment, full
    // (2, 1) windows only) leaves every task's chain where its first clamped window begins, k_windowed takes it up from
    // there; null: k_windowed runs the whole chain
    int32_t* state = nullptr;
    // k_windowed_sys (16 lanes per alignment, any window shape of up to 15 blocks, score only) flags what it leaves to
    // k_windowed_cp -- N / non-canonical symbols -- in o_abort; k_windowed_cp then runs only the tasks whose flag is set
    int32_t* o_abort = nullptr;  const int32_t* only_if = nullptr;
    int32_t prio = 0;      // k_windowed_quad / k_windowed_sys: s_setprio 3
};
// k_windowed_quad: LDS per wave = {Pv after, Mv before} of the traceback's 64 columns, [slot][lane] x 8 B, 65 slots
// (the lanes of a quad run one column apart)
enum : int { QE_WQ_SLOTS = 65, QE_WQ_LDS_PER_WAVE = QE_WQ_SLOTS * 64 * 8 };

// QuickEd's stage-1 rule on the device (quicked.c:201-202): the WindowEd(2,1) score of a task is its bound unless too many
// of its windows were high-error ones; est = the cutoff the host sized the align step for
struct Stage1Args {
    int32_t nt;
    const int32_t* pair;  const int32_t* m;  const int32_t* n;
    const int32_t* score;  const int32_t* hew;  const u32* steps;   // k_windowed's outputs
    const int32_t* est;
    u32 hew_percentage;
    int32_t* o_cut;      // the bound = the align step's cutoff
    int32_t* o_skip;     // bit 0: the pair goes on to stage 2; bit 1: its bound exceeds the estimate the buffers were sized for; bit 2: see flags
    u32* o_steps;        // copy of steps (the stage's buffers are recycled before the run is fetched)
    const u32* flags = nullptr;   // the pack flags by pair, where a score pass follows instead of the align step (only_score): a pair with
                                  // FLAG_NONCANON leaves the list too (bit 2 of o_skip) -- its edit count is not the matrix's end value
};

// Where a stopped score-only BandEd launch left its band (what the Hirschberg join reads)
struct BandState {
    int32_t G;     // 1: k_banded<false> layout (per 64-task group, column = lane); >= 2: k_banded_coop layout (per wave of 64/G tasks)
    const uint8_t* ws;  const int64_t* g_ws_off;  const int32_t* g_nslots;  const int32_t* g_nrows;  const int32_t* g_nch;
    const int32_t* first;  const int32_t* last;  const int32_t* posv;  const int32_t* maxrow;
    const int32_t* abort;  // coop only: tasks recomputed by the fallback pass live in fb's layout
};

// Hirschberg midpoint join (bpm_hirschberg.c:102-200, re-derived: SURVEY A.5 / A.7(12)); node j of
// the split list is task j of both the forward and the reverse half-pass launch
struct JoinArgs {
    int32_t nnodes;
    const int32_t* m;  const int32_t* n1;  const int32_t* n2;
    BandState F, R;       // cooperative launches (or the only ones when G == 1)
    BandState Ffb, Rfb;   // their k_banded<false> fallback passes (G == 1 layout); used where abort[j] != 0
    int32_t* o_best;  int32_t* o_score_l;  int32_t* o_score_r;  int32_t* o_ok;
};

// CIGAR of a pair = its segments in order, each either the RLE runs of a leaf alignment
// (kind 0: a = leaf task index) or a literal run (kind 1: a = op, b = count), merged across
// segment borders exactly like cigar_sprint over one operations buffer (cigar.c:453-488)
struct SegFormatArgs {
    int32_t npairs;
    const int64_t* seg_off;      // [npairs + 1]
    const int32_t* seg_kind;  const int32_t* seg_a;  const int32_t* seg_b;
    const u32* runs;  const int64_t* g_runs_off;  const int32_t* nruns;   // per leaf task
    const int32_t* g_runs_cap;  int32_t runs_by_task;                      // layout of `runs` (TraceArgs::runs_by_task)
    int32_t* o_len;  int32_t* o_edits;  int32_t* o_nops;                  // per pair-list entry
    const int64_t* str_off;  char* pool;
    // 0: the reference's RLE "MXID" (cigar_sprint, cigar.c:453-488); 1: SAM with mismatches "=XID";
    // 2: SAM "MID", X folded into M before runs are merged (cigar_compute_CIGAR, cigar.c:194-240, 504-529)
    int32_t style;
};

// cigar_check_alignment (cigar.c:363-434) of every entry's alignment against the raw bytes of its pair
struct SegCheckArgs {
    SegFormatArgs F;
    PairView P;
    const int32_t* root_pair;
    int32_t* o_ok;               // 1 valid, 0 not
};

}  // namespace qe
