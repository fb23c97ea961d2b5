"""Folds the rocprofv3 counter CSVs of tools/collect_profiles.sh into one JSON per milestone:
per workload and kernel the average per-launch FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) and the HBM
byte estimate; SQ / GRBM counters of the BandEd score kernel when collected.
FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide streaming reads (MI355X_MICROARCH.md, HBM): the guide's x2
correction applies to 16 B/lane loads; `fetch_factor` says which factor each kernel's figure uses (tools/pmc_calib.hip
calibrates the 8 B/lane row pattern of the BandEd kernels)."""
import collections
import csv
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
wls = sys.argv[3:] or ["banded_score", "quicked", "cfg4"]


def rows(path):
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return list(csv.DictReader(f))


def short(k):
    return k.replace("void ", "").replace("qe::", "").split("(")[0]


res = {}
for wl in wls:
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in rows(f"{out}/{tag}_{wl}_pmc_{c}.csv"):
            k = r["Kernel_Name"]
            if "qe::" not in k:
                continue
            k = short(k)
            d[k][c] += float(r["Counter_Value"])
            n[(k, c)].add(r["Dispatch_Id"])
    res[wl] = {}
    for k, v in d.items():
        f = v["FETCH_SIZE"] / max(len(n[(k, "FETCH_SIZE")]), 1)
        w = v["WRITE_SIZE"] / max(len(n[(k, "WRITE_SIZE")]), 1)
        factor = float(os.environ.get("QE_FETCH_FACTOR", "2.0"))
        res[wl][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_factor": factor,
                      "hbm_bytes": (factor * f + w) * 1024, "launches": len(n[(k, "FETCH_SIZE")])}
# calibration: tools/pmc_calib streams 4 GiB per kernel; factor = true bytes / reported bytes
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in rows(f"{out}/{tag}_calib_pmc_{c}.csv"):
        k = r["Kernel_Name"]
        if "k_read" not in k and "k_write" not in k:
            continue
        name = k.replace("void ", "").split("(")[0]
        kib = float(r["Counter_Value"])
        if kib > 0 and (("k_read" in k) == (c == "FETCH_SIZE")):
            calib[f"{name}:{c}"] = {"reported_KiB": kib, "true_KiB": 4.0 * 1024 * 1024, "factor": 4.0 * 1024 * 1024 / kib}
if calib:
    res["calibration"] = calib
sq = {}
for part in ("sq1", "sq2", "grbm"):
    for r in rows(f"{out}/{tag}_banded_score_pmc_{part}.csv"):
        if "k_banded" not in r["Kernel_Name"]:
            continue
        sq[r["Counter_Name"]] = sq.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if sq:
    res["banded_score_sq"] = sq
qsq = collections.defaultdict(dict)
for part in ("sq1", "sq2", "tcc", "tcp"):
    for r in rows(f"{out}/{tag}_quicked_pmc_{part}.csv"):
        if "qe::" in r["Kernel_Name"]:
            k = short(r["Kernel_Name"])
            qsq[k][r["Counter_Name"]] = qsq[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in qsq.items():
    if v.get("SQ_WAVE_CYCLES"):
        v["wait_any_share"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
        v["active_valu_share"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"]
        if "SQ_INST_LEVEL_VMEM" in v:      # vector-memory instructions in flight, summed over wave-cycles (another pass than SQ_WAVE_CYCLES)
            v["vmem_in_flight_per_wave_cycle"] = v["SQ_INST_LEVEL_VMEM"] / v["SQ_WAVE_CYCLES"]
    if v.get("TCC_HIT_sum") is not None and v.get("TCC_MISS_sum") is not None and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
if qsq:
    res["quicked_sq"] = qsq
with open(f"{out}/{tag}_pmc_summary.json", "w") as f:
    json.dump(res, f, indent=1)
# what bench.py prints as roofline.traffic: the dominant kernel's HBM bytes per launch, keyed by workload and size
shape = {"banded_score": [("banded_score:100000x10000", "k_banded<false>")], "quicked": [("quicked:100000x10000", "k_banded<true>")],
         "cfg4": [("quicked:10000x100000", "k_banded<true>"), ("cfg4:10000x100000", "k_banded_coop_lds<false>")],
         "share": [("banded_score:12500x10000", "k_banded<false>")]}
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "quicked_amd", "csrc", "qe_kernels.hip"), "rb") as f:
    kernels_sha = hashlib.sha256(f.read()).hexdigest()
# bench.py prints these figures only while qe_kernels.hip is the file they were measured with
latest = {"kernels_sha256": kernels_sha}
for wl, (key, kern) in [(w, kk) for w, lst in shape.items() for kk in lst]:
    ent = res.get(wl, {}).get(kern)
    if ent:
        latest[key] = dict(kernel=kern, hbm_bytes=ent["hbm_bytes"], FETCH_SIZE_KiB=ent["FETCH_SIZE_KiB"], WRITE_SIZE_KiB=ent["WRITE_SIZE_KiB"],
                           fetch_factor=ent["fetch_factor"],
                           source=f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, per launch; "
                                  "x2 on FETCH_SIZE calibrated with tools/pmc_calib.hip for 4/8/16 B per lane)")
with open(f"{out}/pmc_traffic_latest.json", "w") as f:
    json.dump(latest, f, indent=1)
print(json.dumps(res, indent=1))
