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
// Instead of the reference's OpenMP loop over quicked_align calls (align_benchmark.c:269-284) every
// --batch-size pairs go through one quicked_align_batch call.
//
//   g++ -O2 -std=c++17 tools/align_benchmark.cpp -Iinclude -Lquicked_amd -lquicked_hip -Wl,-rpath,$PWD/quicked_amd
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <getopt.h>
#include <string>
#include <vector>

#include "quicked_batch.h"

static void usage() {
    fprintf(stderr,
            "USE: ./align_benchmark -a ALGORITHM -i PATH\n"
            "  --algorithm|a quicked | edit-banded | edit-banded-hirschberg | edit-windowed\n"
            "  --input|i PATH   --output|o PATH   --output-full PATH\n"
            "  --bandwidth INT  --window-size INT  --overlap-size INT  --hew-threshold INT  --hew-percentage INT\n"
            "  --force-scalar   --only-score\n"
            "  --check|c score|alignment|correct   --batch-size INT   --device INT   --progress|P INT   --verbose|v   --help|h\n");
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

int main(int argc, char** argv) {
    // the HIP runtime reads this when it initialises (the library's first HIP call): a thread's runs rotate over up to 12
    // stream sets, and streams that share a hardware queue serialise
    setenv("GPU_MAX_HW_QUEUES", "24", 0);
    std::string algo_name, input, output, output_full, check;
    quicked_params_t params = quicked_default_params();
    bool bandwidth_set = false, verbose = false;
    long batch_size = 65536, progress = 100000;
    int device = 0;
    static struct option opts[] = {
        {"algorithm", required_argument, 0, 'a'}, {"input", required_argument, 0, 'i'}, {"output", required_argument, 0, 'o'},
        {"output-full", required_argument, 0, 800}, {"bandwidth", required_argument, 0, 2000},
        {"window-size", required_argument, 0, 2001}, {"overlap-size", required_argument, 0, 2002},
        {"hew-threshold", required_argument, 0, 2003}, {"hew-percentage", required_argument, 0, 2004},
        {"force-scalar", no_argument, 0, 2005}, {"only-score", no_argument, 0, 2006}, {"check", required_argument, 0, 'c'},
        {"num-threads", required_argument, 0, 't'}, {"batch-size", required_argument, 0, 4000}, {"device", required_argument, 0, 4002},
        {"progress", required_argument, 0, 'P'}, {"verbose", no_argument, 0, 'v'}, {"quiet", no_argument, 0, 'q'},
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
            case 't': break;                       // accepted for compatibility: the batch call replaces the thread pool
            case 4000: batch_size = atol(optarg); break;
            case 4002: device = atoi(optarg); break;
            case 'P': progress = atol(optarg); break;
            case 'v': verbose = true; break;
            case 'q': progress = 0; break;
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
    if (!bandwidth_set) params.bandwidth = 15;    // align_benchmark_params.c:299-306
    std::ifstream in(input);
    if (!in) { fprintf(stderr, "Input file '%s' couldn't be opened\n", input.c_str()); return 1; }
    FILE* out = nullptr; bool full = false;
    if (!output_full.empty()) { out = fopen(output_full.c_str(), "w"); full = true; }
    else if (!output.empty()) out = fopen(output.c_str(), "w");

    if (quicked_set_device(device) != QUICKED_OK) { fprintf(stderr, "no usable HIP device %d\n", device); return 1; }
    quicked_aligner_t aligner;
    if (quicked_check_error(quicked_new(&aligner, &params))) return 1;

    std::vector<std::string> pats, txts;
    long total = 0, ok_score = 0, ok_cigar = 0, checked = 0;
    double align_s = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto flush = [&]() {
        const int n = (int)pats.size();
        if (n == 0) return 0;
        std::vector<const char*> pp(n), tp(n);
        std::vector<int> pl(n), tl(n), scores(n, -1);
        std::vector<char*> cigs(n, nullptr);
        std::vector<quicked_status_t> status(n, QUICKED_OK);
        for (int i = 0; i < n; ++i) { pp[i] = pats[i].data(); pl[i] = (int)pats[i].size(); tp[i] = txts[i].data(); tl[i] = (int)txts[i].size(); }
        const auto t0 = std::chrono::steady_clock::now();
        quicked_align_batch(&aligner, n, pp.data(), pl.data(), tp.data(), tl.data(), scores.data(),
                            params.only_score ? nullptr : cigs.data(), status.data());
        align_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // --check: every CIGAR of the batch validated against its pair on the device in one call
        std::vector<int32_t> valid(n, 1);
        if (!check.empty() && !params.only_score) {
            std::string ppool, tpool, cpool;
            std::vector<int64_t> po(n), to(n), co(n, -1);
            std::vector<int32_t> pl32(n), tl32(n);
            for (int i = 0; i < n; ++i) {
                po[i] = (int64_t)ppool.size(); ppool += pats[i]; to[i] = (int64_t)tpool.size(); tpool += txts[i];
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
        for (int i = 0; i < n; ++i) {
            if (quicked_check_error(status[i])) {
                fprintf(stderr, "%s", quicked_status_msg(status[i]));
                if (out) fprintf(out, full ? "%d\t%d\t-\t%s\t%s\t-\n" : "-\t-\n", pl[i], tl[i], pats[i].c_str(), txts[i].c_str());
                continue;
            }
            const char* cg = (!params.only_score && cigs[i]) ? cigs[i] : "-";
            if (out) {
                if (full) fprintf(out, "%d\t%d\t%d\t%s\t%s\t%s\n", pl[i], tl[i], scores[i], pats[i].c_str(), txts[i].c_str(), cg);
                else fprintf(out, "%d\t%s\n", scores[i], cg);
            }
            if (!check.empty()) {
                ++checked;
                if (params.only_score || (valid[i] == 1 && cigar_edits(cg) == scores[i])) ++ok_cigar;
                else fprintf(stderr, "INCORRECT ALIGNMENT (pair %ld)\n", total + i);
                if (check == "score" || check == "alignment") {
                    const int exact = exact_distance(pats[i], txts[i]);
                    if (exact == scores[i]) ++ok_score;
                    else fprintf(stderr, "INACCURATE SCORE (pair %ld: %d, exact %d)\n", total + i, scores[i], exact);
                }
            }
        }
        total += n;
        pats.clear(); txts.clear();
        return n;
    };
    std::string l1, l2;
    while (std::getline(in, l1) && std::getline(in, l2)) {
        if (!l1.empty() && l1.back() == '\r') l1.pop_back();
        if (!l2.empty() && l2.back() == '\r') l2.pop_back();
        // the tag character tells which line is which (generate_dataset.c:398-408 writes either order)
        std::string& pat = (!l1.empty() && l1[0] == '<') ? l2 : l1;
        std::string& txt = (!l1.empty() && l1[0] == '<') ? l1 : l2;
        pats.emplace_back(pat.empty() ? "" : pat.substr(1));
        txts.emplace_back(txt.empty() ? "" : txt.substr(1));
        if ((long)pats.size() >= batch_size) {
            flush();
            if (progress > 0 && total % progress < batch_size)
                fprintf(stderr, "...processed %ld reads\n", total);
        }
    }
    flush();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (out) fclose(out);
    fprintf(stderr, "[Benchmark]\n=> Total.reads              %ld\n=> Time.Benchmark           %.3f s\n  => Time.Alignment         %.3f s (%.1f seq/s)\n",
            total, wall, align_s, align_s > 0 ? total / align_s : 0.0);
    if (params.algo == QUICKED && verbose) {
        // the stage timers of the aligner, as align_benchmark.c:116-128 prints them with --verbose (a batch is one lap
        // of each stage timer: calls = batches that went through the stage)
        auto line = [&](const char* name, const profiler_timer_t* t) {
            const double s = t ? (double)t->time_ns.total * 1e-9 : 0.0;
            fprintf(stderr, "  => Time.%-15s %9.3f s  (%6.2f %%) (%llu calls)\n", name, s, wall > 0 ? 100.0 * s / wall : 0.0,
                    (unsigned long long)(t ? t->time_ns.samples : 0));
        };
        line("Windowed Small", aligner.timer_windowed_s);
        line("Windowed Large", aligner.timer_windowed_l);
        line("Banded", aligner.timer_banded);
        line("Align", aligner.timer_align);
    }
    quicked_free(&aligner);
    if (!check.empty()) {
        fprintf(stderr, "[Accuracy]\n => Alignments.Correct     %ld/%ld (%.2f %%)\n", ok_cigar, checked, checked ? 100.0 * ok_cigar / checked : 0.0);
        if (check != "correct")
            fprintf(stderr, " => Score.Correct          %ld/%ld (%.2f %%)\n", ok_score, checked, checked ? 100.0 * ok_score / checked : 0.0);
    }
    return 0;
}
