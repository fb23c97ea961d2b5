// align_benchmark.cpp -- command-line harness over libquicked_hip.so with the interface of the
// reference's tools/align_benchmark:
//   input   two lines per pair, ">PATTERN" then "<TEXT" (align_benchmark.c:73-99: the first
//           character of each line is dropped)
//   -a      quicked | edit-banded | edit-banded-hirschberg | edit-windowed   (align_benchmark.c:146-184;
//           like the reference's CLI, edit-banded runs with only_score = false)
//   params  --bandwidth --window-size --overlap-size --hew-threshold --hew-percentage --force-scalar
//           (align_benchmark_params.c:108-131; quicked defaults bandwidth to 15: 299-306)
//   -o      "score\tCIGAR" per pair; --output-full: plen, tlen, score, pattern, text, CIGAR
//           (benchmark_utils.c:151-170)
//   -c      score | alignment | correct: CIGAR validity + edit count -- the validity walk of every pair's CIGAR over its
//           bases (benchmark_check.c:117-176 -> cigar_check_alignment, cigar.c:363-434) runs on the device, a batch at a
//           time (quicked_batch_validate); the host only adds up the run lengths -- and for `score` an independent exact
//           distance (full-height bit-parallel DP in this tool; the reference uses edlib)
//   -t N    the reference's parallel mode (align_benchmark.c:246-284: N aligners, one per OpenMP thread, over disjoint pairs
//           of every block read) mapped to GPUs: N worker threads, worker g on device g % (devices in use) with its own
//           aligner; every --batch-size pairs are cut into contiguous jobs that go to the workers in turn; reading the
//           next jobs, aligning and writing the finished ones (in input order: the output file is byte-identical to
//           -t 1's) overlap.  --devices D limits the devices in use (default: all the node has, at most N).
//           The totals the tool prints (reads, score sum, correct alignments) are reduced over the devices with
//           ncclAllReduce (RCCL over xGMI; librccl is loaded when more than one device is in use; with one device only
//           when --force-rccl / QE_FORCE_RCCL=1 asks for it: ncclCommInitAll over that one device, the same grouped
//           all-reduce, checked against the host-summed totals -- how the path is exercised on a one-GPU box).
// Instead of the reference's loop over quicked_align calls every job goes through one quicked_align_batch call.
//
//   g++ -O2 -std=c++17 -pthread tools/align_benchmark.cpp -Iinclude -Lquicked_amd -lquicked_hip -ldl -Wl,-rpath,$PWD/quicked_amd
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <deque>
#include <dlfcn.h>
#include <getopt.h>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "quicked_batch.h"

static void usage() {
    fprintf(stderr,
            "USE: ./align_benchmark -a ALGORITHM -i PATH\n"
            "  --algorithm|a quicked | edit-banded | edit-banded-hirschberg | edit-windowed\n"
            "  --input|i PATH   --output|o PATH   --output-full PATH\n"
            "  --bandwidth INT  --window-size INT  --overlap-size INT  --hew-threshold INT  --hew-percentage INT\n"
            "  --force-scalar   --only-score\n"
            "  --check|c score|alignment|correct   --batch-size INT   --progress|P INT   --verbose INT | -v   --quiet|q   --help|h\n"
            "  --num-threads|t INT (worker threads, one aligner each)   --device INT (first device)   --devices INT (devices in use)\n"
            "  --force-rccl (reduce the totals through ncclAllReduce with ONE device in use too)\n");
}

static int encode(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 4; }
}

// independent exact edit distance: full-height Myers bit-parallel DP, no band
static int exact_distance(const std::string& p, const std::string& t) {
    const size_t m = p.size(), nw = (m + 63) / 64;
    std::vector<uint64_t> peq(nw * 5, 0), P(nw, ~0ull), M(nw, 0);
    for (size_t i = 0; i < m; ++i) peq[(i / 64) * 5 + encode(p[i])] |= 1ull << (i % 64);
    int score = (int)m;
    const uint64_t last_bit = 1ull << ((m - 1) % 64);
    for (char ch : t) {
        const int c = encode(ch);
        uint64_t hp = 1, hm = 0;
        for (size_t r = 0; r < nw; ++r) {
            const uint64_t Eq = peq[r * 5 + c], Pv = P[r], Mv = M[r];
            const uint64_t Xv = Eq | Mv, Eqc = Eq | hm;
            const uint64_t Xh = (((Eqc & Pv) + Pv) ^ Pv) | Eqc;
            uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
            const uint64_t out_bit = (r + 1 == nw) ? last_bit : (1ull << 63);
            const uint64_t ohp = (Ph & out_bit) != 0, ohm = (Mh & out_bit) != 0;
            Ph = (Ph << 1) | hp; Mh = (Mh << 1) | hm;
            P[r] = Mh | ~(Xv | Ph); M[r] = Ph & Xv;
            hp = ohp; hm = ohm;
        }
        score += (int)hp - (int)hm;
    }
    return score;
}

// edit count of an RLE CIGAR (cigar_score_edit, cigar.c:274-289): the sum of its X / I / D run lengths -- O(runs); the
// walk over the bases that decides validity is the device's (quicked_batch_validate)
static long cigar_edits(const char* rle) {
    long num = 0, e = 0;
    for (const char* q = rle; *q; ++q) {
        if (*q >= '0' && *q <= '9') { num = num * 10 + (*q - '0'); continue; }
        if (*q != 'M') e += num;
        num = 0;
    }
    return e;
}

// ---- totals over the devices in use: ncclAllReduce(SUM) from librccl, one rank per device, driven from one thread inside a
// group (the single-process multi-GPU form of the API).  librccl (half a GB) is loaded only when it is needed.
struct DeviceTotals { long long v[5]; };          // reads, score sum, correct alignments, correct scores, alignments checked
static bool reduce_over_devices(std::vector<DeviceTotals>& per_dev, const std::vector<int>& devs, DeviceTotals* out) {
    typedef int (*init_all_t)(void**, int, const int*);
    typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, void*);
    typedef int (*group_t)(void);
    typedef int (*destroy_t)(void*);
    typedef int (*hip_malloc_t)(void**, size_t);
    typedef int (*hip_free_t)(void*);
    typedef int (*hip_set_t)(int);
    typedef int (*hip_memcpy_t)(void*, const void*, size_t, int);
    typedef int (*hip_sync_t)(void);
    void* rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!rccl) rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);        // already in the process (libquicked_hip.so depends on it)
    if (!rccl || !hip) { fprintf(stderr, "RCCL reduction: %s\n", dlerror()); return false; }
    init_all_t comm_init_all = (init_all_t)dlsym(rccl, "ncclCommInitAll");
    all_reduce_t all_reduce = (all_reduce_t)dlsym(rccl, "ncclAllReduce");
    group_t group_start = (group_t)dlsym(rccl, "ncclGroupStart"), group_end = (group_t)dlsym(rccl, "ncclGroupEnd");
    destroy_t comm_destroy = (destroy_t)dlsym(rccl, "ncclCommDestroy");
    hip_malloc_t d_malloc = (hip_malloc_t)dlsym(hip, "hipMalloc");
    hip_free_t d_free = (hip_free_t)dlsym(hip, "hipFree");
    hip_set_t set_device = (hip_set_t)dlsym(hip, "hipSetDevice");
    hip_memcpy_t d_memcpy = (hip_memcpy_t)dlsym(hip, "hipMemcpy");
    hip_sync_t d_sync = (hip_sync_t)dlsym(hip, "hipDeviceSynchronize");
    if (!comm_init_all || !all_reduce || !group_start || !group_end || !comm_destroy || !d_malloc || !d_free || !set_device || !d_memcpy || !d_sync) return false;
    const int nd = (int)devs.size();
    std::vector<void*> comms((size_t)nd, nullptr), bufs((size_t)nd, nullptr);
    if (comm_init_all(comms.data(), nd, devs.data()) != 0) { fprintf(stderr, "ncclCommInitAll failed\n"); return false; }
    bool ok = true;
    for (int r = 0; r < nd && ok; ++r) {
        ok = set_device(devs[(size_t)r]) == 0 && d_malloc(&bufs[(size_t)r], sizeof(DeviceTotals)) == 0 &&
             d_memcpy(bufs[(size_t)r], &per_dev[(size_t)r], sizeof(DeviceTotals), 1 /* hipMemcpyHostToDevice */) == 0;
    }
    if (ok) {
        group_start();
        for (int r = 0; r < nd; ++r) {
            set_device(devs[(size_t)r]);
            ok &= all_reduce(bufs[(size_t)r], bufs[(size_t)r], 5, 4 /* ncclInt64 */, 0 /* ncclSum */, comms[(size_t)r], nullptr /* the device's null stream */) == 0;
        }
        ok &= group_end() == 0;
        for (int r = 0; r < nd && ok; ++r) { set_device(devs[(size_t)r]); ok &= d_sync() == 0; }
    }
    // every rank holds the sum: rank 0's copy is the one reported, the others are compared with it
    for (int r = 0; r < nd && ok; ++r) {
        DeviceTotals got;
        set_device(devs[(size_t)r]);
        ok = d_memcpy(&got, bufs[(size_t)r], sizeof(got), 2 /* hipMemcpyDeviceToHost */) == 0;
        if (r == 0) *out = got;
        else if (ok && memcmp(&got, out, sizeof(got)) != 0) { fprintf(stderr, "RCCL reduction: ranks disagree\n"); ok = false; }
    }
    for (int r = 0; r < nd; ++r) { set_device(devs[(size_t)r]); if (bufs[(size_t)r]) d_free(bufs[(size_t)r]); if (comms[(size_t)r]) comm_destroy(comms[(size_t)r]); }
    return ok;
}

// one job: a contiguous stretch of the input, aligned by one worker, written in sequence order
struct Job {
    long seq = 0, first = 0;                      // sequence number, index of its first pair in the input
    std::vector<std::string> pats, txts;
    std::string text;                             // what goes to the output file
    std::string err;                              // what goes to stderr (in order too)
    bool done = false;
};

int main(int argc, char** argv) {
    // the HIP runtime reads this when it initialises (the library's first HIP call): a thread's runs rotate over up to 12
    // stream sets, and streams that share a hardware queue serialise
    setenv("GPU_MAX_HW_QUEUES", "20", 0);
    std::string algo_name, input, output, output_full, check;
    quicked_params_t params = quicked_default_params();
    bool bandwidth_set = false, force_rccl = getenv("QE_FORCE_RCCL") != nullptr && atoi(getenv("QE_FORCE_RCCL")) != 0;
    int verbose = 0;                              // align_benchmark_params.c:60, 241-252: -v is level 1, --verbose takes the level
    long batch_size = 65536, progress = 100000;
    int device = 0, num_threads = 1, devices_wanted = 0;
    static struct option opts[] = {
        {"algorithm", required_argument, 0, 'a'}, {"input", required_argument, 0, 'i'}, {"output", required_argument, 0, 'o'},
        {"output-full", required_argument, 0, 800}, {"bandwidth", required_argument, 0, 2000},
        {"window-size", required_argument, 0, 2001}, {"overlap-size", required_argument, 0, 2002},
        {"hew-threshold", required_argument, 0, 2003}, {"hew-percentage", required_argument, 0, 2004},
        {"force-scalar", no_argument, 0, 2005}, {"only-score", no_argument, 0, 2006}, {"check", required_argument, 0, 'c'},
        {"num-threads", required_argument, 0, 't'}, {"batch-size", required_argument, 0, 4000}, {"device", required_argument, 0, 4002},
        {"devices", required_argument, 0, 4003},
        {"progress", required_argument, 0, 'P'}, {"verbose", required_argument, 0, 4001}, {"verbose1", no_argument, 0, 'v'},
        {"quiet", no_argument, 0, 'q'}, {"force-rccl", no_argument, 0, 4004},
        {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    if (argc <= 1) { usage(); return 0; }
    int c, idx;
    while ((c = getopt_long(argc, argv, "a:i:o:c:t:P:vqh", opts, &idx)) != -1) {
        switch (c) {
            case 'a': algo_name = optarg; break;
            case 'i': input = optarg; break;
            case 'o': output = optarg; break;
            case 800: output_full = optarg; break;
            case 2000: params.bandwidth = (unsigned)atoi(optarg); bandwidth_set = true; break;
            case 2001: params.window_size = (unsigned)atoi(optarg); break;
            case 2002: params.overlap_size = (unsigned)atoi(optarg); break;
            case 2003: params.hew_threshold[0] = params.hew_threshold[1] = (unsigned)atoi(optarg); break;
            case 2004: params.hew_percentage[0] = params.hew_percentage[1] = (unsigned)atoi(optarg); break;
            case 2005: params.force_scalar = true; break;
            case 2006: params.only_score = true; break;
            case 'c': check = optarg; break;
            case 't': num_threads = atoi(optarg); break;
            case 4000: batch_size = atol(optarg); break;
            case 4002: device = atoi(optarg); break;
            case 4003: devices_wanted = atoi(optarg); break;
            case 'P': progress = atol(optarg); break;
            case 'v': verbose = 1; break;
            case 4001:                                                  // align_benchmark_params.c:244-250
                verbose = atoi(optarg);
                if (verbose < 0 || verbose > 4) { fprintf(stderr, "Option '--verbose' must be in {0,1,2,3,4}\n"); return 1; }
                break;
            case 4004: force_rccl = true; break;
            case 'q': progress = 0; verbose = -1; break;                // align_benchmark_params.c:251-253
            case 'h': usage(); return 0;
            default: fprintf(stderr, "Option not recognized\n"); return 1;
        }
    }
    if (algo_name == "quicked") params.algo = QUICKED;
    else if (algo_name == "edit-banded") params.algo = BANDED;
    else if (algo_name == "edit-banded-hirschberg") params.algo = HIRSCHBERG;
    else if (algo_name == "edit-windowed") params.algo = WINDOWED;
    else { fprintf(stderr, "Algorithm '%s' not recognized\n", algo_name.c_str()); return 1; }
    if (input.empty()) { fprintf(stderr, "Option --input is required \n"); return 1; }
    if (!check.empty() && check != "score" && check != "alignment" && check != "correct") {
        fprintf(stderr, "Option '--check' must be in {'correct','score','alignment'}\n"); return 1;
    }
    if (num_threads < 1 || num_threads > 256) { fprintf(stderr, "Option '--num-threads' must be in [1, 256]\n"); return 1; }
    if (batch_size < 1) batch_size = 1;
    if (!bandwidth_set) params.bandwidth = 15;    // align_benchmark_params.c:299-306
    std::ifstream in(input);
    if (!in) { fprintf(stderr, "Input file '%s' couldn't be opened\n", input.c_str()); return 1; }
    FILE* out = nullptr; bool full = false;
    if (!output_full.empty()) { out = fopen(output_full.c_str(), "w"); full = true; }
    else if (!output.empty()) out = fopen(output.c_str(), "w");

    const int ndev_node = quicked_device_count();
    if (ndev_node <= 0 || device < 0 || device >= ndev_node) { fprintf(stderr, "no usable HIP device %d\n", device); return 1; }
    int ndev = std::min(num_threads, ndev_node - device);
    if (devices_wanted > 0) ndev = std::min(ndev, devices_wanted);
    std::vector<int> devs;
    for (int d = 0; d < ndev; ++d) devs.push_back(device + d);

    // ---- the pipeline: reader (this thread) -> workers -> writer
    const int W = num_threads;
    const long job_pairs = std::max<long>(1, (batch_size + W - 1) / W);     // a block of --batch-size pairs = W contiguous jobs (align_benchmark.c:269-284)
    const size_t max_jobs_in_flight = (size_t)3 * (size_t)W;                // reading runs at most three jobs per worker ahead of writing
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_room;
    std::vector<std::deque<std::shared_ptr<Job>>> todo((size_t)W);
    std::deque<std::shared_ptr<Job>> in_order;                              // every job not yet written, by sequence number
    bool reading_done = false, failed = false;
    std::vector<DeviceTotals> per_dev((size_t)ndev, DeviceTotals{{0, 0, 0, 0, 0}});
    std::vector<double> align_s((size_t)W, 0.0);
    std::vector<quicked_aligner_t> aligners((size_t)W);
    std::vector<quicked_params_t> wparams((size_t)W, params);

    auto work = [&](int g) {
        const int dev_slot = g % ndev;
        if (quicked_set_device(devs[(size_t)dev_slot]) != QUICKED_OK || quicked_check_error(quicked_new(&aligners[(size_t)g], &wparams[(size_t)g]))) {
            std::lock_guard<std::mutex> lk(mu); failed = true; cv_done.notify_all(); cv_room.notify_all(); return;
        }
        quicked_aligner_t& aligner = aligners[(size_t)g];
        DeviceTotals mine{{0, 0, 0, 0, 0}};
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return !todo[(size_t)g].empty() || reading_done || failed; });
                if (failed || todo[(size_t)g].empty()) break;
                job = todo[(size_t)g].front(); todo[(size_t)g].pop_front();
            }
            Job& J = *job;
            const int n = (int)J.pats.size();
            std::vector<const char*> pp(n), tp(n);
            std::vector<int> pl(n), tl(n), scores(n, -1);
            std::vector<char*> cigs(n, nullptr);
            std::vector<quicked_status_t> status(n, QUICKED_OK);
            for (int i = 0; i < n; ++i) { pp[i] = J.pats[i].data(); pl[i] = (int)J.pats[i].size(); tp[i] = J.txts[i].data(); tl[i] = (int)J.txts[i].size(); }
            const auto t0 = std::chrono::steady_clock::now();
            quicked_align_batch(&aligner, n, pp.data(), pl.data(), tp.data(), tl.data(), scores.data(),
                                params.only_score ? nullptr : cigs.data(), status.data());
            align_s[(size_t)g] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            // --check: every CIGAR of the job validated against its pair on the device in one call
            std::vector<int32_t> valid(n, 1);
            if (!check.empty() && !params.only_score) {
                std::string ppool, tpool, cpool;
                std::vector<int64_t> po(n), to(n), co(n, -1);
                std::vector<int32_t> pl32(n), tl32(n);
                for (int i = 0; i < n; ++i) {
                    po[i] = (int64_t)ppool.size(); ppool += J.pats[i]; to[i] = (int64_t)tpool.size(); tpool += J.txts[i];
                    pl32[i] = pl[i]; tl32[i] = tl[i];
                    if (!quicked_check_error(status[i]) && cigs[i]) { co[i] = (int64_t)cpool.size(); cpool.append(cigs[i]); cpool.push_back('\0'); }
                }
                quicked_batch_t* vb = quicked_batch_create(n, ppool.data(), po.data(), pl32.data(), tpool.data(), to.data(), tl32.data());
                if (!vb || quicked_check_error(quicked_batch_validate(vb, cpool.data(), (int64_t)cpool.size(), co.data(), valid.data()))) {
                    fprintf(stderr, "--check: the device validator failed\n");
                    exit(1);
                }
                quicked_batch_destroy(vb);
            }
            char num[160];
            for (int i = 0; i < n; ++i) {
                if (quicked_check_error(status[i])) {
                    J.err += quicked_status_msg(status[i]);
                    if (out) {
                        if (full) { snprintf(num, sizeof(num), "%d\t%d\t-\t", pl[i], tl[i]); J.text += num; J.text += J.pats[i]; J.text += '\t'; J.text += J.txts[i]; J.text += "\t-\n"; }
                        else J.text += "-\t-\n";
                    }
                    continue;
                }
                const char* cg = (!params.only_score && cigs[i]) ? cigs[i] : "-";
                if (out) {
                    if (full) {
                        snprintf(num, sizeof(num), "%d\t%d\t%d\t", pl[i], tl[i], scores[i]);
                        J.text += num; J.text += J.pats[i]; J.text += '\t'; J.text += J.txts[i]; J.text += '\t'; J.text += cg; J.text += '\n';
                    } else { snprintf(num, sizeof(num), "%d\t", scores[i]); J.text += num; J.text += cg; J.text += '\n'; }
                }
                mine.v[1] += scores[i];
                if (!check.empty()) {
                    ++mine.v[4];
                    if (params.only_score || (valid[i] == 1 && cigar_edits(cg) == scores[i])) ++mine.v[2];
                    else { snprintf(num, sizeof(num), "INCORRECT ALIGNMENT (pair %ld)\n", J.first + i); J.err += num; }
                    if (check == "score" || check == "alignment") {
                        const int exact = exact_distance(J.pats[i], J.txts[i]);
                        if (exact == scores[i]) ++mine.v[3];
                        else { snprintf(num, sizeof(num), "INACCURATE SCORE (pair %ld: %d, exact %d)\n", J.first + i, scores[i], exact); J.err += num; }
                    }
                }
            }
            mine.v[0] += n;
            J.pats.clear(); J.txts.clear(); J.pats.shrink_to_fit(); J.txts.shrink_to_fit();
            { std::lock_guard<std::mutex> lk(mu); J.done = true; }
            cv_done.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        for (int k = 0; k < 5; ++k) per_dev[(size_t)dev_slot].v[k] += mine.v[k];
    };

    auto write = [&]() {
        long next_report = progress;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_done.wait(lk, [&] { return failed || (!in_order.empty() && in_order.front()->done) || (in_order.empty() && reading_done); });
                if (failed || in_order.empty()) return;
                job = in_order.front(); in_order.pop_front();
            }
            cv_room.notify_all();
            if (!job->err.empty()) fputs(job->err.c_str(), stderr);
            if (out && !job->text.empty()) fwrite(job->text.data(), 1, job->text.size(), out);
            const long upto = job->first + job_pairs;
            if (progress > 0 && upto >= next_report) { fprintf(stderr, "...processed %ld reads\n", (upto / progress) * progress); next_report = (upto / progress + 1) * progress; }
        }
    };

    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int g = 0; g < W; ++g) workers.emplace_back(work, g);
    std::thread writer(write);
    {   // the reader
        long seq = 0, total_read = 0;
        std::shared_ptr<Job> cur;
        auto submit = [&]() {
            if (!cur || cur->pats.empty()) return;
            std::unique_lock<std::mutex> lk(mu);
            cv_room.wait(lk, [&] { return failed || in_order.size() < max_jobs_in_flight; });
            if (failed) return;
            in_order.push_back(cur);
            todo[(size_t)(cur->seq % W)].push_back(cur);
            lk.unlock();
            cv_work.notify_all();
            cur.reset();
        };
        std::string l1, l2;
        while (std::getline(in, l1) && std::getline(in, l2)) {
            if (!l1.empty() && l1.back() == '\r') l1.pop_back();
            if (!l2.empty() && l2.back() == '\r') l2.pop_back();
            // the tag character tells which line is which (generate_dataset.c:398-408 writes either order)
            std::string& pat = (!l1.empty() && l1[0] == '<') ? l2 : l1;
            std::string& txt = (!l1.empty() && l1[0] == '<') ? l1 : l2;
            if (!cur) { cur = std::make_shared<Job>(); cur->seq = seq++; cur->first = total_read; }
            cur->pats.emplace_back(pat.empty() ? "" : pat.substr(1));
            cur->txts.emplace_back(txt.empty() ? "" : txt.substr(1));
            ++total_read;
            if ((long)cur->pats.size() >= job_pairs) submit();
        }
        submit();
        { std::lock_guard<std::mutex> lk(mu); reading_done = true; }
        cv_work.notify_all(); cv_done.notify_all();
    }
    for (auto& th : workers) th.join();
    { std::lock_guard<std::mutex> lk(mu); reading_done = true; }
    cv_done.notify_all();
    writer.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (out) fclose(out);
    if (failed) { fprintf(stderr, "a worker thread could not set up its device / aligner\n"); return 1; }

    // ---- totals: over the devices in use through RCCL, over one device on the host
    DeviceTotals tot{{0, 0, 0, 0, 0}};
    const char* how = "host";
    if (ndev > 1 || force_rccl) {
        if (!reduce_over_devices(per_dev, devs, &tot)) { fprintf(stderr, "the RCCL reduction over %d devices failed\n", ndev); return 1; }
        how = "ncclAllReduce";
        // the host's own sum of the same figures: what the collective must return
        DeviceTotals host{{0, 0, 0, 0, 0}};
        for (const DeviceTotals& d : per_dev) for (int k = 0; k < 5; ++k) host.v[k] += d.v[k];
        if (memcmp(&host, &tot, sizeof(tot)) != 0) { fprintf(stderr, "the RCCL reduction disagrees with the host-summed totals\n"); return 1; }
    } else tot = per_dev[0];
    const long total = (long)tot.v[0], ok_cigar = (long)tot.v[2], ok_score = (long)tot.v[3];
    const long checked = (long)tot.v[4];
    double busiest = 0;
    for (double a : align_s) busiest = std::max(busiest, a);
    fprintf(stderr, "[Benchmark]\n=> Total.reads              %ld\n=> Time.Benchmark           %.3f s\n  => Time.Alignment         %.3f s (%.1f seq/s)\n",
            total, wall, busiest, busiest > 0 ? total / busiest : 0.0);
    fprintf(stderr, "=> Threads %d  Devices %d  Totals.by %s  Score.sum %lld\n", W, ndev, how, tot.v[1]);
    if (params.algo == QUICKED && verbose > 0) {
        // the stage timers of the aligners, as align_benchmark.c:116-128 prints them with --verbose (a job is one lap of
        // each stage timer: calls = jobs that went through the stage), summed over the workers
        auto line = [&](const char* name, int which) {
            double s = 0; unsigned long long calls = 0;
            for (const quicked_aligner_t& a : aligners) {
                const profiler_timer_t* t = which == 0 ? a.timer_windowed_s : which == 1 ? a.timer_windowed_l : which == 2 ? a.timer_banded : a.timer_align;
                if (t) { s += (double)t->time_ns.total * 1e-9; calls += t->time_ns.samples; }
            }
            fprintf(stderr, "  => Time.%-15s %9.3f s  (%6.2f %%) (%llu calls)\n", name, s, wall > 0 ? 100.0 * s / wall : 0.0, calls);
        };
        line("Windowed Small", 0);
        line("Windowed Large", 1);
        line("Banded", 2);
        line("Align", 3);
    }
    for (quicked_aligner_t& a : aligners) quicked_free(&a);
    if (!check.empty()) {
        fprintf(stderr, "[Accuracy]\n => Alignments.Correct     %ld/%ld (%.2f %%)\n", ok_cigar, checked, checked ? 100.0 * ok_cigar / checked : 0.0);
        if (check != "correct")
            fprintf(stderr, " => Score.Correct          %ld/%ld (%.2f %%)\n", ok_score, checked, checked ? 100.0 * ok_score / checked : 0.0);
    }
    return 0;
}
