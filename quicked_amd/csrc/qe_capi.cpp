// qe_capi.cpp -- the C-ABI of libquicked_hip.so: the six reference entry points (include/quicked.h == quicked/quicked.h:36-96)
// and the additive batch surface (include/quicked_batch.h).  Host code only: argument checks, locking (a batch's fin_mu, the
// calling thread's context: ApiScope), error translation (the ABI has no exception channel), the aligner's timers and
// strings.  Everything that touches the device is in qe_driver.hip; device memory and contexts in qe_pool.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "qe_batch.h"

#define QE_API extern "C" __attribute__((visibility("default")))

using namespace qe;

QE_API void* quicked_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
QE_API void quicked_host_free(void* p) { if (p) (void)hipHostFree(p); }

QE_API int quicked_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return std::min(count, (int)QE_MAX_DEVICES);
}

QE_API quicked_status_t quicked_set_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= QE_MAX_DEVICES) return QUICKED_ERROR;
    tl_device = device;
    return QUICKED_OK;
}

// every call that works on a batch object holds its fin_mu: an early finish of the batch's last run (qe::finisher_work) is
// waited for, and cannot start in the middle of the call
static quicked_status_t guard(quicked_batch* B, quicked_status_t (*fn)(quicked_batch*, void*), void* arg) {
    std::unique_lock<std::mutex> lk;
    if (B) lk = std::unique_lock<std::mutex>(B->fin_mu);
    ApiScope scope;
    try { return fn(B, arg); }
    catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
        return QUICKED_ERROR;
    }
    catch (const std::bad_alloc&) { fprintf(stderr, "[quicked_hip] out of host memory\n"); return QUICKED_ERROR; }
}


static quicked_batch* guarded_new(const std::function<void(quicked_batch*)>& load) {
    quicked_batch* B = nullptr;
    ApiScope scope;
    try {
        B = new quicked_batch();
        load(B);
        return B;
    } catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
    } catch (const std::bad_alloc&) {
        fprintf(stderr, "[quicked_hip] out of host memory\n");
    }
    delete B;
    return nullptr;
}

QE_API quicked_batch_t* quicked_batch_create(int64_t n,
                                             const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                             const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    if (n < 0) return nullptr;
    return guarded_new([&](quicked_batch* B) {
        Context& C = ctx();
        batch_load(B, C, n, pattern_pool, pattern_off, pattern_len, text_pool, text_off, text_len);
    });
}

QE_API quicked_status_t quicked_batch_reload(quicked_batch_t* batch, int64_t n,
                                             const char* pattern_pool, const int64_t* pattern_off, const int32_t* pattern_len,
                                             const char* text_pool, const int64_t* text_off, const int32_t* text_len) {
    if (!batch || n < 0) return QUICKED_ERROR;
    struct Arg { int64_t n; const char* pp; const int64_t* po; const int32_t* pl; const char* tp; const int64_t* to; const int32_t* tl; }
        arg{n, pattern_pool, pattern_off, pattern_len, text_pool, text_off, text_len};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        batch_quiesce(B);
        batch_load(B, C, x->n, x->pp, x->po, x->pl, x->tp, x->to, x->tl);
        return QUICKED_OK;
    }, &arg);
}

// ---- packed wire format (supersedes sequence_buffer_t, sequence_buffer.h:30-50) ----------------
QE_API int64_t quicked_wire_words(int32_t len, int wire) {
    if (len < 0) return -1;
    if (wire == QUICKED_WIRE_2BIT) return ((int64_t)len + 31) / 32;
    if (wire == QUICKED_WIRE_PLANES3) return 3 * (((int64_t)len + 63) / 64);
    return -1;
}

// host-side serializer of one sequence (upper-case A, C, G, T; N only in PLANES3): the reference's code table
// (dna_text.c:41-46) restricted to the symbols whose raw-byte and encoded comparisons agree
QE_API quicked_status_t quicked_wire_pack(const char* seq, int32_t len, int wire, uint64_t* out) {
    const int64_t nwords = quicked_wire_words(len, wire);
    if (nwords < 0 || (len > 0 && (!seq || !out))) return QUICKED_ERROR;
    for (int64_t i = 0; i < nwords; ++i) out[i] = 0;
    for (int32_t i = 0; i < len; ++i) {
        int code;
        switch (seq[i]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                          case 'N': code = 4; break; default: return QUICKED_ERROR; }
        if (wire == QUICKED_WIRE_2BIT) {
            if (code == 4) return QUICKED_ERROR;
            out[i >> 5] |= (uint64_t)code << (2 * (i & 31));
        } else {
            uint64_t* row = out + 3 * (int64_t)(i >> 6);
            const uint64_t bit = (uint64_t)1 << (i & 63);
            if (code == 4) row[2] |= bit;
            else { if (code & 1) row[0] |= bit; if (code & 2) row[1] |= bit; }
        }
    }
    return QUICKED_OK;
}


QE_API quicked_batch_t* quicked_batch_create_packed(int64_t n, int wire,
                                                    const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                                    const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    if (n < 0 || (wire != QUICKED_WIRE_2BIT && wire != QUICKED_WIRE_PLANES3)) return nullptr;
    return guarded_new([&](quicked_batch* B) {
        Context& C = ctx();
        batch_load_packed(B, C, n, wire, pattern_words, pattern_word_off, pattern_len, text_words, text_word_off, text_len);
    });
}

QE_API quicked_status_t quicked_batch_reload_packed(quicked_batch_t* batch, int64_t n, int wire,
                                                    const uint64_t* pattern_words, const int64_t* pattern_word_off, const int32_t* pattern_len,
                                                    const uint64_t* text_words, const int64_t* text_word_off, const int32_t* text_len) {
    if (!batch || n < 0 || (wire != QUICKED_WIRE_2BIT && wire != QUICKED_WIRE_PLANES3)) return QUICKED_ERROR;
    struct Arg { int64_t n; int wire; const uint64_t* pw; const int64_t* po; const int32_t* pl; const uint64_t* tw; const int64_t* to; const int32_t* tl; }
        arg{n, wire, pattern_words, pattern_word_off, pattern_len, text_words, text_word_off, text_len};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        batch_quiesce(B);
        batch_load_packed(B, C, x->n, x->wire, x->pw, x->po, x->pl, x->tw, x->to, x->tl);
        return QUICKED_OK;
    }, &arg);
}

QE_API quicked_status_t quicked_batch_fetch(quicked_batch_t* batch) {
    if (!batch) return QUICKED_ERROR;
    return guard(batch, [](quicked_batch* B, void*) {
        if (!B->pending_fetch && B->shadow_ready) {
            // an early-finish thread has brought the run's results to the host already, into the set the getters do not
            // read: it becomes the visible one
            B->shadow_ready = false;
            B->vis ^= 1;
            B->wr = &B->res[B->vis];
            return B->fin_status < 0 ? B->fin_status : QUICKED_OK;
        }
        B->wr = &B->res[B->vis];
        return fetch_results(*B);
    }, nullptr);
}

QE_API void quicked_batch_destroy(quicked_batch_t* batch) {
    if (!batch) return;
    {   // early-finish jobs still queued for this batch find nothing to do and retire.  The batch's fin_mu BEFORE this thread
        // takes its context, the order every other call takes them in (guard()); it is released before the context is taken
        std::unique_lock<std::mutex> lk(batch->fin_mu);
        batch->pending_fetch.reset();
        batch->fin_cv.wait(lk, [&] { return batch->fin_jobs == 0; });
    }
    ApiScope scope;
    try {
        tl_device = batch->device;
        (void)ctx();                           // binds the batch's device to this thread
        batch_quiesce(batch);                  // runs queued by any thread; hipFree then synchronises the device itself
    } catch (const HipError&) { (void)hipGetLastError(); }
    delete batch;
}

QE_API quicked_status_t quicked_batch_run(quicked_batch_t* batch, const quicked_params_t* params, int sync) {
    struct Arg { const quicked_params_t* p; int sync; } arg{params, sync};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        return run_batch(*B, *x->p, x->sync != 0);
    }, &arg);
}

QE_API quicked_status_t quicked_batch_sync(quicked_batch_t* batch) {
    return guard(batch, [](quicked_batch* B, void*) {
        tl_device = B->device;
        Context& C = ctx();
        C.sync_all();
        batch_quiesce(B);                      // runs of this batch queued by OTHER threads are over too
        B->pending = false;
        return QUICKED_OK;
    }, nullptr);
}

QE_API quicked_status_t quicked_batch_kernel_times(quicked_batch_t* batch, double ms_sum[4], int64_t launches[4]) {
    struct Arg { double* ms; int64_t* n; } arg{ms_sum, launches};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        tl_device = B->device;
        Context& C = ctx();
        C.sync_all();
        for (int k = 0; k < 4; ++k) { x->ms[k] = 0; x->n[k] = 0; }
        for (size_t i = 0; i < C.kev_used; ++i) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, C.kev[i].first, C.kev[i].second));
            const int k = C.kev_kind[i] & 3;
            x->ms[k] += ms; ++x->n[k];
        }
        C.kev_used = 0;
        return QUICKED_OK;
    }, &arg);
}

QE_API quicked_status_t quicked_batch_kernel_time(quicked_batch_t* batch, double* ms_sum, int64_t* launches) {
    double ms[4]; int64_t n[4];
    const quicked_status_t st = quicked_batch_kernel_times(batch, ms, n);
    if (st < 0) return st;
    *ms_sum = ms[0] + ms[1]; *launches = n[0] + n[1];
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_scores(quicked_batch_t* batch, int32_t* scores_out, int32_t* status_out) {
    if (batch->res[batch->vis].score.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (scores_out) memcpy(scores_out, batch->res[batch->vis].score.data(), (size_t)batch->n * sizeof(int32_t));
    if (status_out) memcpy(status_out, batch->res[batch->vis].status.data(), (size_t)batch->n * sizeof(int32_t));
    return QUICKED_OK;
}

QE_API int64_t quicked_batch_cigar_bytes(quicked_batch_t* batch) { return (int64_t)batch->res[batch->vis].cigar_pool.size; }

QE_API quicked_status_t quicked_batch_cigar_view(quicked_batch_t* batch, const char** cigar_pool, const int64_t** cigar_off) {
    if (!batch || batch->res[batch->vis].cigar_off.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (cigar_pool) *cigar_pool = batch->res[batch->vis].cigar_pool.p;
    if (cigar_off) *cigar_off = batch->res[batch->vis].cigar_off.data();
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_cigars(quicked_batch_t* batch, char* cigar_pool, int64_t* cigar_off) {
    if (batch->res[batch->vis].cigar_off.size() != (size_t)batch->n) return QUICKED_ERROR;
    if (cigar_pool && batch->res[batch->vis].cigar_pool.size) memcpy(cigar_pool, batch->res[batch->vis].cigar_pool.p, batch->res[batch->vis].cigar_pool.size);
    if (cigar_off) memcpy(cigar_off, batch->res[batch->vis].cigar_off.data(), (size_t)batch->n * sizeof(int64_t));
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_configure(quicked_batch_t* batch, int cigar_style, int check) {
    if (!batch || cigar_style < 0 || cigar_style > 2) return QUICKED_ERROR;
    if (check && batch->packed) return QUICKED_UNIMPLEMENTED;
    batch->cigar_style = cigar_style;
    batch->check = check != 0;
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_check_results(quicked_batch_t* batch, int32_t* ok_out) {
    if (!batch || batch->res[batch->vis].check_ok.size() != (size_t)batch->n) return QUICKED_ERROR;
    memcpy(ok_out, batch->res[batch->vis].check_ok.data(), (size_t)batch->n * sizeof(int32_t));
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_batch_validate(quicked_batch_t* batch, const char* cigar_pool, int64_t pool_bytes,
                                               const int64_t* cigar_off, int32_t* ok_out) {
    struct Arg { const char* pool; int64_t bytes; const int64_t* off; int32_t* ok; } arg{cigar_pool, pool_bytes, cigar_off, ok_out};
    return guard(batch, [](quicked_batch* B, void* a) {
        Arg* x = (Arg*)a;
        if (!x->off || !x->ok || (x->bytes > 0 && !x->pool)) return QUICKED_ERROR;
        if (B->packed) return QUICKED_UNIMPLEMENTED;          // the validator compares raw bytes; a packed batch has none
        tl_device = B->device;
        Context& C = ctx();
        return batch_validate(B, C, x->pool, x->bytes, x->off, x->ok);
    }, &arg);
}

QE_API quicked_status_t quicked_pool_stats(int64_t stats_out[8]) {
    for (int q = 0; q < 8; ++q) stats_out[q] = 0;
    for (const auto& bk : g_book) stats_out[1] += bk.oom_events.load();
    Context* C = tl_ctx;
    const int dev = C ? C->device : tl_device;
    if (dev >= 0 && dev < QE_MAX_DEVICES) stats_out[5] = (int64_t)g_book[dev].held.load();
    { std::lock_guard<std::mutex> lk(g_ctx_mu); for (const Context* c : g_ctx_all) { ++stats_out[6]; if (c->leased.load()) ++stats_out[7]; } }
    if (!C) return QUICKED_OK;
    stats_out[0] = (int64_t)C->held.load();
    stats_out[2] = C->last_na; stats_out[3] = C->last_sub_batches; stats_out[4] = (int64_t)C->pool_budget;
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_pool_trim(void) {
    finisher_retire();                                      // the library's own idle threads (their contexts lose their leases)
    ApiScope scope;
    try {
        Context& C = ctx();
        (void)C.release_pools(nullptr, true);
        C.release_small_pinned();                           // the pinned block of small loads and read-backs (1 MB; comes back on demand)
        { std::lock_guard<std::mutex> lk(g_ctx_mu); C.planned = 0; C.wanted = 0; }
        (void)release_unleased(C.device);                  // what threads that have ended left behind: pools ...
        retire_idle_streams(C.device);                      // ... and streams nobody is using
        return QUICKED_OK;
    } catch (const HipError& e) {
        fprintf(stderr, "[quicked_hip] HIP error %d (%s) at %s, qe_driver.hip:%d\n", (int)e.e, hipGetErrorString(e.e), e.what, e.line);
        return QUICKED_ERROR;
    }
}

QE_API quicked_status_t quicked_debug_reload_env(void) {
    qe::switches_reload();
    return QUICKED_OK;
}

QE_API quicked_status_t quicked_early_finish_stats(int64_t stats_out[4]) {
    qe::early_finish_stats(stats_out);
    return QUICKED_OK;
}

QE_API int64_t quicked_batch_deferred_pairs(quicked_batch_t* batch) { return batch ? batch->res[batch->vis].deferred_pairs : -1; }

QE_API quicked_status_t quicked_batch_counters(quicked_batch_t* batch, int64_t counters_out[8]) {
    memcpy(counters_out, batch->res[batch->vis].counters, sizeof(batch->counters));
    return QUICKED_OK;
}

// ---- the six reference entry points ---------------------------------------
QE_API bool quicked_check_error(quicked_status_t status) { return status < 0; }    // quicked.c:380

QE_API const char* quicked_status_msg(quicked_status_t status) {                    // quicked.c:382-403
    switch (status) {
        case QUICKED_ERROR: return "ERROR: QuickEd has finished with unspecific error\n";
        case QUICKED_FAIL_NON_CONVERGENCE: return "ERROR: Hirschberg algorithm can not find a middle point of subsequence division!\n";
        case QUICKED_UNIMPLEMENTED: return "ERROR: The algorithm or parameter combination selected is not implemented\n";
        case QUICKED_UNKNOWN_ALGO: return "ERROR: Unknown algorithm selection\n";
        case QUICKED_EMPTY_SEQUENCE: return "ERROR: Tried to align an empty sequence\n";
        default: return "QuickEd finished without errors.\n";
    }
}

QE_API quicked_params_t quicked_default_params(void) {                             // quicked.c:308-321
    quicked_params_t p;
    memset(&p, 0, sizeof(p));
    p.algo = QUICKED;
    p.bandwidth = 15;
    p.window_size = 9;
    p.overlap_size = 1;
    p.hew_threshold[0] = p.hew_threshold[1] = 40;
    p.hew_percentage[0] = p.hew_percentage[1] = 15;
    return p;
}

// host timers the ABI exposes (profiler_timer.c:53-73, profiler_counter.c:46-66)
static void qe_timer_reset(profiler_timer_t* t) { memset(t, 0, sizeof(*t)); t->time_ns.min = UINT64_MAX; }

// what the library hangs off aligner->mm_allocator when it owns it: the
// reference keeps its arena there (quicked.c:330-334); here it is the host
// block that owns the five timers and the last CIGAR strings.
struct AlignerState {
    mm_allocator_t shim;                  // first member: a valid mm_allocator_t* for callers that only pass it around
    profiler_timer_t timers[5];
    std::vector<char*> batch_cigars;
    std::vector<char> batch_pool;
    std::vector<char*> strings;           // every CIGAR quicked_align handed out: valid until quicked_free (quicked.c:48-50, 357-361)
    uint32_t magic;
};
static const uint32_t QE_MAGIC = 0x51CEDA11u;
// the same list for aligners that were given an external allocator (no AlignerState to hang it on)
static std::mutex g_strings_mu;
static std::unordered_map<const quicked_aligner_t*, std::vector<char*>> g_strings;
static AlignerState* own_state(const quicked_aligner_t* aligner) {
    if (aligner->mm_allocator == nullptr || aligner->params->external_allocator != nullptr) return nullptr;
    AlignerState* st = (AlignerState*)aligner->mm_allocator;
    return st->magic == QE_MAGIC ? st : nullptr;
}
static void keep_string(quicked_aligner_t* aligner, char* str) {
    if (AlignerState* st = own_state(aligner)) { st->strings.push_back(str); return; }
    std::lock_guard<std::mutex> lk(g_strings_mu);
    g_strings[aligner].push_back(str);
}
static void drop_strings(quicked_aligner_t* aligner) {
    bool listed = false;
    auto drop = [&](std::vector<char*>& v) { for (char* q : v) { listed |= q == aligner->cigar; free(q); } v.clear(); };
    if (AlignerState* st = own_state(aligner)) drop(st->strings);
    {
        std::lock_guard<std::mutex> lk(g_strings_mu);
        auto it = g_strings.find(aligner);
        if (it != g_strings.end()) { drop(it->second); g_strings.erase(it); }
    }
    if (aligner->cigar != nullptr && !listed) free(aligner->cigar);
    aligner->cigar = nullptr;
}

QE_API quicked_status_t quicked_new(quicked_aligner_t* aligner, quicked_params_t* params) {    // quicked.c:323-352
    aligner->params = params;
    aligner->score = -1;
    aligner->cigar = nullptr;
    AlignerState* st = nullptr;
    if (params->external_allocator == nullptr) {
        st = new AlignerState();
        memset(&st->shim, 0, sizeof(st->shim));
        st->magic = QE_MAGIC;
        aligner->mm_allocator = &st->shim;
    } else {
        aligner->mm_allocator = params->external_allocator;
    }
    if (params->external_timer) {
        // the caller patches the five pointers after quicked_new (benchmark_edit.c:61-65); NULL until then
        aligner->timer = aligner->timer_windowed_s = aligner->timer_windowed_l = aligner->timer_banded = aligner->timer_align = nullptr;
    } else {
        profiler_timer_t* tm = st ? st->timers : (profiler_timer_t*)calloc(5, sizeof(profiler_timer_t));
        for (int i = 0; i < 5; ++i) qe_timer_reset(&tm[i]);
        aligner->timer = &tm[0]; aligner->timer_windowed_s = &tm[1]; aligner->timer_windowed_l = &tm[2];
        aligner->timer_banded = &tm[3]; aligner->timer_align = &tm[4];
    }
    return QUICKED_WIP;
}

QE_API quicked_status_t quicked_free(quicked_aligner_t* aligner) {                             // quicked.c:354-378
    drop_strings(aligner);                 // every string quicked_align returned stays valid until here, as in the reference
    const bool own = aligner->mm_allocator != nullptr && aligner->params->external_allocator == nullptr;
    if (!aligner->params->external_timer && !own) free(aligner->timer);       // calloc'ed block of five
    if (own) {
        AlignerState* st = (AlignerState*)aligner->mm_allocator;
        if (st->magic == QE_MAGIC) delete st;
        aligner->mm_allocator = nullptr;
    }
    return QUICKED_WIP;
}

static quicked_status_t align_pairs(quicked_aligner_t* aligner, int n, const char* const* patterns, const int* plens,
                                    const char* const* texts, const int* tlens, int* scores_out, char** cigars_out,
                                    quicked_status_t* status_out, std::vector<char>* pool_keep) {
    std::vector<int64_t> po((size_t)n), to((size_t)n);
    std::vector<int32_t> pl((size_t)n), tl((size_t)n);
    size_t pb = 0, tb = 0;
    for (int i = 0; i < n; ++i) { po[i] = (int64_t)pb; pb += (size_t)plens[i]; to[i] = (int64_t)tb; tb += (size_t)tlens[i]; pl[i] = plens[i]; tl[i] = tlens[i]; }
    std::vector<char> pp(pb + 1), tp(tb + 1);
    for (int i = 0; i < n; ++i) {
        if (plens[i]) memcpy(pp.data() + po[i], patterns[i], (size_t)plens[i]);
        if (tlens[i]) memcpy(tp.data() + to[i], texts[i], (size_t)tlens[i]);
    }
    double tr_last = now_ms();
    ApiScope scope;
    // small calls (quicked_align, small quicked_align_batch) reuse one batch object per thread and device: no hipMalloc /
    // hipFree (a device-wide synchronisation) per call
    // (the object belongs to the thread's context: the next thread that takes the context over inherits it)
    const bool small = pb + tb <= ((size_t)8 << 20);
    quicked_batch_t* B = nullptr;
    if (small) {
        quicked_batch*& slot = *reinterpret_cast<quicked_batch**>(&ctx().small_batch);
        if (!slot) slot = quicked_batch_create(n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data());
        else if (quicked_batch_reload(slot, n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data()) < 0) {
            quicked_batch_destroy(slot); slot = nullptr;
        }
        B = slot;
    } else B = quicked_batch_create(n, pp.data(), po.data(), pl.data(), tp.data(), to.data(), tl.data());
    if (!B) return QUICKED_ERROR;
    QE_TRACE_POINT("align_pairs: create");
    const quicked_params_t* p = aligner->params;
    // the five host timers are ticked around the stages they bracket in the reference
    // (quicked.c:76-78,184-193,204-235,240-275,283-294); a batch is one lap of each.
    tl_timers.windowed_s = aligner->timer_windowed_s; tl_timers.windowed_l = aligner->timer_windowed_l;
    tl_timers.banded = aligner->timer_banded; tl_timers.align = aligner->timer_align;
    qe_timer_start(aligner->timer);
    quicked_status_t st = quicked_batch_run(B, p, 1);
    qe_timer_stop(aligner->timer);
    QE_TRACE_POINT("align_pairs: run");
    tl_timers = HostTimers();
    quicked_status_t first_err = QUICKED_OK;
    bool any_err = false;
    for (int i = 0; i < n; ++i) {
        const quicked_status_t s = (st < 0 && st != QUICKED_EMPTY_SEQUENCE) ? st : (quicked_status_t)B->res[B->vis].status[(size_t)i];
        if (status_out) status_out[i] = s;
        if (s < 0 && !any_err) { any_err = true; first_err = s; }
        // a split that did not converge still has a score and a CIGAR in the reference (run_hirschberg extracts them from the
        // partial operations buffer before it returns the status, quicked.c:149-160): what the converged leaves gave
        if (scores_out && (s >= 0 || s == QUICKED_FAIL_NON_CONVERGENCE)) scores_out[i] = B->res[B->vis].score[(size_t)i];
    }
    if (cigars_out) {
        const quicked_batch::HostResults& R = B->res[B->vis];
        pool_keep->assign(R.cigar_pool.p, R.cigar_pool.p + R.cigar_pool.size);
        for (int i = 0; i < n; ++i)
            cigars_out[i] = (R.cigar_off[(size_t)i] >= 0 && !p->only_score) ? pool_keep->data() + R.cigar_off[(size_t)i] : nullptr;
    }
    if (!small) quicked_batch_destroy(B);
    QE_TRACE_POINT("align_pairs: destroy");
    if (any_err) return first_err;
    return st;
}

QE_API quicked_status_t quicked_align(quicked_aligner_t* aligner, const char* pattern, const int pattern_len,
                                      const char* text, const int text_len) {                  // quicked.c:405-437
    if (pattern_len == 0 || text_len == 0) return QUICKED_EMPTY_SEQUENCE;
    if ((unsigned)aligner->params->algo > (unsigned)HIRSCHBERG) return QUICKED_UNKNOWN_ALGO;
    int score = -1;
    char* cg = nullptr;
    std::vector<char> keep;
    quicked_status_t one = QUICKED_OK;
    const quicked_status_t st = align_pairs(aligner, 1, &pattern, &pattern_len, &text, &text_len, &score,
                                            aligner->params->only_score ? nullptr : &cg, &one, &keep);
    if (st < 0 && st != QUICKED_FAIL_NON_CONVERGENCE) return st;
    aligner->score = score;                      // also on QUICKED_FAIL_NON_CONVERGENCE, like extract_results (quicked.c:149-160)
    if (!aligner->params->only_score && cg) {
        // a previous align's string stays valid until quicked_free, as in the reference (arena allocation that the
        // next align does not release, quicked.c:48-50, 357-361)
        char* dup = strdup(cg);
        if (!dup) return QUICKED_ERROR;
        keep_string(aligner, dup);
        aligner->cigar = dup;
    }
    return st;
}

QE_API quicked_status_t quicked_align_batch(quicked_aligner_t* aligner, int n,
                                            const char* const* patterns, const int* pattern_lens,
                                            const char* const* texts, const int* text_lens,
                                            int* scores_out, char** cigars_out, quicked_status_t* status_out) {
    if (n <= 0) return QUICKED_OK;
    if ((unsigned)aligner->params->algo > (unsigned)HIRSCHBERG) {
        if (status_out) for (int i = 0; i < n; ++i) status_out[i] = QUICKED_UNKNOWN_ALGO;
        return QUICKED_UNKNOWN_ALGO;
    }
    const bool own = aligner->mm_allocator != nullptr && aligner->params->external_allocator == nullptr &&
                     ((AlignerState*)aligner->mm_allocator)->magic == QE_MAGIC;
    static thread_local std::vector<char> tl_keep;     // strings of the last batch when the aligner cannot own them
    std::vector<char>* keep = own ? &((AlignerState*)aligner->mm_allocator)->batch_pool : &tl_keep;
    return align_pairs(aligner, n, patterns, pattern_lens, texts, text_lens, scores_out, cigars_out, status_out, keep);
}
