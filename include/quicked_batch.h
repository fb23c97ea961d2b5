/*
 * quicked_batch.h -- additive batch surface of libquicked_hip.so.
 *
 * Not in the reference (its only batch mode is an OpenMP loop over
 * quicked_align calls, tools/align_benchmark/align_benchmark.c:269-284, fed
 * from sequence_buffer_t, quicked_utils/include/sequence_buffer.h:30-50).
 * Per-pair semantics are those of quicked_align() with the aligner's params;
 * the six reference signatures and both struct layouts are untouched.
 * Plain pointers and sizes only.
 */
#ifndef QUICKED_BATCH_H
#define QUICKED_BATCH_H

#include <stddef.h>

#include "quicked.h"

#ifdef __cplusplus
extern "C" {
#endif

/* selects the HIP device used by subsequent calls of this thread (default 0).  One process may drive every GPU of a node:
 * a host thread per device, each with its own aligner / batch objects (tools/align_benchmark.cpp -t N: the reference's one
 * aligner per thread, align_benchmark.c:246-249, mapped to devices); pairs are sharded, there is no data-path collective. */
quicked_status_t quicked_set_device(int device);
/* HIP devices this process sees (0 if there is no usable runtime) */
int quicked_device_count(void);

/* Convenience form: n pairs given as host pointers.  scores_out[n];
 * cigars_out may be NULL (scores only); otherwise cigars_out[i] is a
 * NUL-terminated RLE string owned by the aligner until the next batch call or
 * quicked_free() (NULL for pairs whose status is an error).  status_out may be
 * NULL.  Returns the first error status, else the common success status. */
quicked_status_t quicked_align_batch(quicked_aligner_t* aligner, int n,
                                     const char* const* patterns, const int* pattern_lens,
                                     const char* const* texts, const int* text_lens,
                                     int* scores_out, char** cigars_out,
                                     quicked_status_t* status_out);

/* Pinned host memory for sequence pools: quicked_batch_create() DMAs straight from it (pageable pools
 * are pipelined through pinned staging instead). */
void* quicked_host_alloc(size_t bytes);
void quicked_host_free(void* p);

/* ---- resident batches: upload once, run many times (what bench.py times) --- */
typedef struct quicked_batch quicked_batch_t;

/* Pairs stored back to back in two host byte pools (the batch wire format):
 * pattern i = pattern_pool[pattern_off[i] .. +pattern_len[i]).  Copies the
 * pools to HBM (H2D) and sizes the device pool; no alignment work. */
quicked_batch_t* quicked_batch_create(int64_t n,
                                      const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                      const char* text_pool, const int64_t* text_off, const int32_t* text_len);
void quicked_batch_destroy(quicked_batch_t* batch);
/* Loads n new pairs into an existing batch object: same arguments as quicked_batch_create, but the object's device
 * arena is kept when it is large enough (no hipMalloc / hipFree, which synchronise the device).  Waits for the runs of
 * the batch that are still on the device; may be called from another thread than the one that runs the batch -- a
 * client streams by reloading batch k+1 on an uploader thread while batch k runs (bench.py's end-to-end leg). */
quicked_status_t quicked_batch_reload(quicked_batch_t* batch, int64_t n,
                                      const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                      const char* text_pool, const int64_t* text_off, const int32_t* text_len);

/* ---- packed wire format (supersedes sequence_buffer_t, tools/align_benchmark/utils/sequence_buffer.h:30-50)
 * For sequences over upper-case A, C, G, T (and N in PLANES3) -- the symbols whose raw-byte and encoded
 * comparisons agree (dna_text.c:41-46, bpm_banded.c:1012); everything else must use the ASCII form.
 *   QUICKED_WIRE_2BIT     2 bits per base: base i of a sequence in bits 2(i%32).. of its word i/32,
 *                         codes A 0, C 1, G 2, T 3; ceil(len/32) words
 *   QUICKED_WIRE_PLANES3  per 64 bases three words {code bit 0, code bit 1, not-ACGT}; 3 ceil(len/64) words
 * A packed batch uploads 4x / 2.7x fewer bytes than ASCII, keeps no bytes in HBM and skips the pack stage of
 * every run; scores and CIGARs are identical to the ASCII batch of the same sequences.  The raw-byte
 * validator is not available for it (QUICKED_UNIMPLEMENTED). */
typedef enum { QUICKED_WIRE_2BIT = 2, QUICKED_WIRE_PLANES3 = 3 } quicked_wire_t;
int64_t quicked_wire_words(int32_t len, int wire);
/* host-side serializer of one sequence; QUICKED_ERROR if a symbol is not representable in `wire` */
quicked_status_t quicked_wire_pack(const char* seq, int32_t len, int wire, uint64_t* out);
/* The same serializer over a whole pool in one call: sequence i = pool[off[i] .. +len[i]) goes to out_words + out_off[i]
 * (quicked_wire_words(len[i], wire) words; quicked_wire_offsets lays the sequences out back to back and returns the
 * total).  SIMD (AVX-512BW or AVX2, with BMI2; picked at run time, scalar otherwise) on `threads` host threads (0: the
 * CPUs the process may use, at most 32).  This is how a caller that holds ASCII -- what the reference's API consumes,
 * quicked.c:405-437 -- gets under the PCIe bound: 2 GB of ASCII per 100 k pairs of 10 kb become 0.5 GB on the wire.
 * Words identical to quicked_wire_pack's.  QUICKED_ERROR if a sequence holds a symbol `wire` cannot represent;
 * *bad_seq (may be NULL) = the first such sequence, else -1. */
quicked_status_t quicked_wire_pack_pool(int64_t n, const char* pool, const int64_t* off, const int32_t* len, int wire,
                                        uint64_t* out_words, const int64_t* out_off, int threads, int64_t* bad_seq);
int64_t quicked_wire_offsets(int64_t n, const int32_t* len, int wire, int64_t* out_off);
/* which kernel quicked_wire_pack_pool uses: 0 scalar, 1 AVX2, 2 AVX-512BW; force >= 0 pins a kernel the CPU has (tests),
 * force < 0 restores the run-time choice; returns the kernel in use, or -1 if the CPU lacks the one asked for */
int quicked_wire_pack_isa(int force);
quicked_batch_t* quicked_batch_create_packed(int64_t n, int wire,
                                             const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                             const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len);

quicked_status_t quicked_batch_reload_packed(quicked_batch_t* batch, int64_t n, int wire,
                                             const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                             const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len);

/* Runs the hot path for every pair with `params` (algo, only_score, ...), from
 * the ASCII bytes resident in HBM to scores (and CIGAR runs) resident in HBM.
 * sync != 0: waits for the run and copies scores / statuses / CIGARs / counters
 * to the host, where the getters below read them.
 * sync == 0: returns once the run is queued; nothing is copied to the host until quicked_batch_fetch().  HIRSCHBERG, and
 * QUICKED where a pair may split or the batch runs for the first time, return once their host-driven stages are done;
 * otherwise QUICKED queues stage 1 and the align step together (the stage-1 rule of quicked.c:201-202 runs on the
 * device) and aligns the pairs that go on to stages 2 / 3 when the run is fetched.  Consecutive runs of a thread rotate over several sets of stream, device
 * pool and bit-planes -- three for batches that fill the chip, up to twelve for small ones (quicked_pool_stats()[2]) --
 * so the kernels of the next runs overlap those of run k. */
quicked_status_t quicked_batch_run(quicked_batch_t* batch, const quicked_params_t* params, int sync);
quicked_status_t quicked_batch_sync(quicked_batch_t* batch);
/* Brings the results of the batch's last sync == 0 run to the host: waits for that run (only that one: later runs of
 * this or other batches keep executing) and copies scores / statuses / CIGARs / counters to where the getters read
 * them.  A sync == 0 run itself leaves the getters' data untouched.  At its end such a run moves its results from the
 * queueing thread's rotating pools into memory of the batch object (device to device), so the thread may queue any number
 * of runs of OTHER batch objects before this one is fetched; the batch's own next run, reload or destroy discards them.
 * Any thread may fetch (bench.py's end-to-end leg fetches on a thread of its own) as long as no other call on this batch
 * object runs at the same time; what the fetch itself has to compute runs on the calling thread's streams and pools.
 * QUICKED: the pairs a queued run left for the host-driven stages (those past stage 1, those above the bound estimate)
 * are aligned by library threads as soon as the run is over ("early finish": up to QE_FINISHERS = 3 threads with streams
 * and pools of their own, started when first needed); the fetch then only waits for that.  Calls on one batch object are
 * serialised against such a thread by the library, and such a thread writes into a second set of host-side result buffers
 * that the fetch makes visible: the getters and the zero-copy views keep showing the previous results until the fetch. */
quicked_status_t quicked_batch_fetch(quicked_batch_t* batch);

/* results of the last sync != 0 run or of the last quicked_batch_fetch (host copies) */
quicked_status_t quicked_batch_scores(quicked_batch_t* batch, int32_t* scores_out, int32_t* status_out);
/* total bytes of all CIGAR strings incl. terminators, then the strings themselves
 * (cigar_off[i] = offset of string i in cigar_pool, -1 if none) */
int64_t quicked_batch_cigar_bytes(quicked_batch_t* batch);
quicked_status_t quicked_batch_cigars(quicked_batch_t* batch, char* cigar_pool, int64_t* cigar_off);
/* the same without the copy: pointers into the batch's own (pinned) result buffers, valid until the next sync != 0 run,
 * fetch, reload or destroy of the batch */
quicked_status_t quicked_batch_cigar_view(quicked_batch_t* batch, const char** cigar_pool, const int64_t** cigar_off);

/* Output options of the runs to come.
 * cigar_style: 0 = the reference's RLE "MXID" (cigar_sprint, quicked_utils/src/cigar.c:453-488; default),
 *              1 = SAM CIGAR with mismatches, "=XID", 2 = SAM CIGAR "MID" with X folded into M before
 *              runs are merged (cigar_compute_CIGAR + cigar_sprint_SAM_CIGAR, cigar.c:194-240, 504-529),
 *              byte-identical to that printer including its one quirk: an alignment that STARTS with a
 *              mismatch keeps it as "1X" (the first operation is read before the mapping step).
 * check != 0:  every CIGAR is validated on the device against the raw bytes of its pair, the same walk
 *              as cigar_check_alignment (cigar.c:363-434); verdicts via quicked_batch_check_results
 *              (1 valid, 0 not, -1 the pair has no alignment) after a sync != 0 run. */
quicked_status_t quicked_batch_configure(quicked_batch_t* batch, int cigar_style, int check);
quicked_status_t quicked_batch_check_results(quicked_batch_t* batch, int32_t* ok_out);
/* The same validator for CIGAR strings from anywhere ("<len><op>", op in MXID, '=' read as M): string i is
 * cigar_pool + cigar_off[i], NUL-terminated, against the batch's resident pair i; cigar_off[i] < 0 -> -1.
 * What `align_benchmark -c correct` does per pair on the host (benchmark_check.c), at batch scale. */
quicked_status_t quicked_batch_validate(quicked_batch_t* batch, const char* cigar_pool, int64_t pool_bytes,
                                        const int64_t* cigar_off, int32_t* ok_out);

/* counters of the last run, for the measurement harness:
 *   [0] block-advances of score-only BandEd passes   [1] of fills
 *   [2] WindowEd block steps   [3] traceback steps   [4] CIGAR ops
 *   [5] last kernel-only time in ns (HIP events on the batch's stream)
 *   [6] pairs that went past stage 1   [7] pairs that went past stage 2 */
quicked_status_t quicked_batch_counters(quicked_batch_t* batch, int64_t counters_out[8]);
/* QUICKED runs that queue stage 1 and the align step together (see quicked_batch_run) align the pairs that leave stage 1,
 * or whose bound exceeds the planned buffers, after the run (early-finish threads, or the fetch): how many pairs of the last
 * sync != 0 run / fetch that were (0 on data like BASELINE configs 2-3; a harness that times sync == 0 runs it never
 * fetches should check this, bench.py does) */
int64_t quicked_batch_deferred_pairs(quicked_batch_t* batch);

/* Early-finish threads, process-wide since load: [0] host-driven flows they ran, [1] batch objects those finished, [2] flows
 * that served the pairs of SEVERAL batch objects at once (the runs that were over when the thread got to work: the flow's
 * duration is launch latency, not pairs), [3] batch objects in such flows.  QE_FINISH_MERGE (default 4) caps the batch
 * objects per flow; 1 = never merge. */
quicked_status_t quicked_early_finish_stats(int64_t stats_out[4]);

/* The library reads its QE_* environment switches (test hooks that force kernel forms, traces; production
 * needs none) ONCE, at their first use, into a table; this parses the environment again.  For in-process test suites that
 * change a switch between runs -- not to be called while other threads are inside the library. */
quicked_status_t quicked_debug_reload_env(void);

/* The device-pool planner's view of the calling thread (replaces mm_allocator, quicked_utils/src/mm_allocator.c:141-426):
 *   [0] bytes its pools hold   [1] allocations that had to take memory from this thread's other pools or from other threads
 *   (process-wide; the planner is there to keep this 0)   [2] pool sets in rotation in the last run   [3] fill sub-batches of
 *   the last run   [4] bytes one pool may hold   [5] bytes all pools of the thread's device hold (every thread's)
 *   [6] contexts {streams, pools} in existence   [7] contexts on lease to a live thread */
quicked_status_t quicked_pool_stats(int64_t stats_out[8]);

/* Gives the calling thread's device pools back to the device (waits for its runs first), and those that ended threads left
 * behind.  The pools belong to a per-thread context and stay allocated between runs -- that is what makes a steady stream of
 * batches allocation-free -- so a thread that is done with large batches while others go on should call this.  A thread that
 * simply ends leaves its context, pools and all, to the next thread that needs one; an allocation that finds the device
 * full takes the pools of threads that have no call in progress (their next run allocates again). */
quicked_status_t quicked_pool_trim(void);

/* Sum of the HIP-event durations (ms) of the dominant kernel (BandEd score /
 * fill) over the runs of this thread since the previous call, and how many
 * launches that was; synchronises the batch's stream. */
quicked_status_t quicked_batch_kernel_time(quicked_batch_t* batch, double* ms_sum, int64_t* launches);
/* The same by kind of launch: [0] score-only BandEd passes (a BANDED run, QuickEd's stage 3), [1] fills, [2] the half passes of
 * Hirschberg's split levels (bpm_hirschberg.c:85-100; the dominant launches of long reads), [3] unused. */
quicked_status_t quicked_batch_kernel_times(quicked_batch_t* batch, double ms_sum[4], int64_t launches[4]);

#ifdef __cplusplus
}
#endif
#endif /* QUICKED_BATCH_H */
