"""README.md's "Measured on one MI355X" paragraph, generated from the committed bench lines (profiles/<tag>_bench_line.json,
profiles/<tag>_bench_cfg4.json, profiles/<tag>_single_call_latency.txt) instead of edited by hand.

    python tools/readme_numbers.py            # prints the paragraph for the newest tag under profiles/
    python tools/readme_numbers.py --write    # rewrites it in README.md between the bench:begin / bench:end markers"""
import glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- bench:begin -->", "<!-- bench:end -->"


def newest_tag():
    tags = sorted(re.match(r"(r\d+_[a-z]+)_bench_line\.json", os.path.basename(f)).group(1)
                  for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    return tags[-1]


def last_json_line(path):
    with open(path) as f:
        return json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])


def m(x, digits=1):
    return f"{x / 1e6:.{digits}f} M"


def paragraph(tag=None):
    tag = tag or newest_tag()
    d = last_json_line(os.path.join(ROOT, "profiles", f"{tag}_bench_line.json"))
    c4 = last_json_line(os.path.join(ROOT, "profiles", f"{tag}_bench_cfg4.json"))
    q = d["workloads"]["quicked"]
    s = d["strong_share"]
    e, eq = d["e2e"], q["e2e"]
    lat = {}
    p = os.path.join(ROOT, "profiles", f"{tag}_single_call_latency.txt")
    if os.path.exists(p):
        for line in open(p):
            mm = re.match(r"len\s+(\d+) (BandEd score-only|QuickEd \+ CIGAR)\s*: median\s+([\d.]+) ms", line)
            if mm:
                lat[(int(mm.group(1)), mm.group(2))] = float(mm.group(3))
    cfg = d["config"]
    text = (
        f"Measured on one MI355X (`profiles/{tag}_bench_line.json` -- the one line `python bench.py` prints -- and "
        f"`{tag}_bench_cfg4.json`; this paragraph is generated from them by `tools/readme_numbers.py`), "
        f"{cfg['pairs_per_gpu'] // 1000} k pairs of {cfg['length'] // 1000} kb at {cfg['error'] * 100:.0f} % error per batch: "
        f"BandEd score-only **{m(d['value'], 2)} alignments/s** device-resident ({d['gcups'] / 1e3:.0f} k GCUPS); "
        f"QuickEd + CIGAR **{m(q['value'], 2)} alignments/s**"
        + (f" (scores only: {m(d['workloads']['quicked_score']['value'], 1)}, one batch alone in "
           f"{d['workloads']['quicked_score']['single_batch_latency_ms']:.1f} ms)" if "value" in d["workloads"].get("quicked_score", {}) else "")
        + f"; end to end with the results on the host "
        f"{m(e['2bit_pinned']['value'])} / {m(eq['2bit_pinned']['value'])} from pinned 2-bit words, "
        f"**{m(e['ascii_hostpacked']['value'])} / {m(eq['ascii_hostpacked']['value'])} from ASCII packed on the host inside the clock**, "
        f"{m(e['ascii_pinned']['value'])} / {m(eq['ascii_pinned']['value'])} from ASCII over the link (PCIe-bound); "
        f"{s['pairs_per_gpu'] / 1e3:.1f} k pairs per batch (the share of one GPU when 100 k pairs go to 8): "
        f"{m(s['banded_score']['value'])} / {m(s['quicked']['value'])} alignments/s as a stream, "
        f"{s['banded_score']['single_batch_latency_ms']:.1f} / {s['quicked']['single_batch_latency_ms']:.1f} ms for one batch alone; "
        f"QuickEd + Hirschberg CIGAR on {c4['config']['pairs_per_gpu'] // 1000} k pairs of {c4['config']['length'] // 1000} kb / "
        f"{c4['config']['error'] * 100:.0f} % reads **{c4['value'] / 1e3:.1f} k alignments/s**; "
        f"QuickEd on pairs with 4 x 800-base indels (stages 2 / 3, band doubling) "
        f"{d['workloads']['quicked_indels']['value'] / 1e6:.2f} M alignments/s"
        + (f", on ordinary batches with {d['workloads']['quicked_mixed']['hard_pairs'] * 100 // d['workloads']['quicked_mixed']['pairs_per_gpu']} % of such pairs "
           f"among them {d['workloads']['quicked_mixed']['value'] / 1e6:.1f} M" if "value" in d["workloads"].get("quicked_mixed", {}) else "")
        + (f" (the per-GPU share, 12.5 k pairs with 1 % of them: {s['quicked_mixed']['value'] / 1e6:.1f} M, the hard pairs of several runs aligned by one merged flow)"
           if "value" in s.get("quicked_mixed", {}) else "")
        + ". "
        + (f"Config 4 inside the one line ({d['workloads']['cfg4']['steps']} timed steps): {d['workloads']['cfg4']['value'] / 1e3:.1f} k alignments/s. "
           if "value" in d["workloads"].get("cfg4", {}) else "")
        + (f"The compiled reference on the same {d['cpu_baseline']['cores']} CPUs does {d['workloads']['cfg4']['cpu_baseline']['value'] / 1e3:.2f} k (config 4), "
           f"{d['workloads']['quicked_indels']['cpu_baseline']['value'] / 1e3:.0f} k (indel pairs) and {d['workloads']['quicked_mixed']['cpu_baseline']['value'] / 1e3:.0f} k (1 % mix) alignments/s. "
           if all("cpu_baseline" in d["workloads"].get(k, {}) and "value" in d["workloads"][k]["cpu_baseline"] for k in ("cfg4", "quicked_indels", "quicked_mixed")) else "")
        + f"The compiled reference on the same box, on the {d['cpu_baseline']['cores']} CPUs the process is allowed (one aligner per thread): "
        f"{d['cpu_baseline']['value'] / 1e3:.0f} k and {q['cpu_baseline']['value'] / 1e3:.0f} k alignments/s for the two 10 kb workloads, "
        f"identical scores on all pairs of the sample.")
    if lat:
        text += (f" One pair at a time the GPU is slower than one CPU core ({lat.get((1000, 'BandEd score-only'), 0):.2f} ms for a 1 kb score, "
                 f"{lat.get((10000, 'BandEd score-only'), 0):.1f} ms for a 10 kb score, {lat.get((10000, 'QuickEd + CIGAR'), 0):.1f} ms for a "
                 f"10 kb QuickEd alignment: DESIGN.md §4.3); use the batch entry points.")
    # wrapped at 120 columns like the rest of the file
    out, line = [], ""
    for w in text.split(" "):
        if line and len(line) + 1 + len(w) > 120:
            out.append(line); line = w
        else:
            line = (line + " " + w) if line else w
    out.append(line)
    return "\n".join(out)


def main():
    tag = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    para = paragraph(tag)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "README.md")
        s = open(path).read()
        a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(path, "w").write(s[:a] + "\n" + para + "\n" + s[b:])
    else:
        print(para)


if __name__ == "__main__":
    main()
