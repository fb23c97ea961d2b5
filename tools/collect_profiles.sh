#!/bin/bash
# Collects what profiles/ holds for one milestone (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag>       -> gpurun_out/<tag>/...
# kernel-trace/stats and each PMC counter in separate passes, as MI355X_MICROARCH.md prescribes.
tag=${1:-r01_x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
for wl in banded_score quicked; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > $out/stats_$wl.log 2>&1
  cp $out/stats_$wl/*/*kernel_stats.csv $out/${tag}_${wl}_100k_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 bench.py --workload $wl --no-cpu-baseline --steps 1 --warmup 0 > $out/pmc_${wl}_$c.log 2>&1
    cp $out/pmc_${wl}_$c/*/*counter_collection.csv $out/${tag}_${wl}_pmc_$c.csv
  done
  python3 bench.py --workload $wl > $out/${tag}_bench_$wl.json 2> $out/bench_$wl.err
done
./tools/bin/valu_rate > $out/${tag}_valu_rates.txt 2>&1
python3 - <<PY
import csv, glob, json, collections
out = {}
for wl in ("banded_score", "quicked"):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(f"$out/${tag}_%s_pmc_%s.csv" % (wl, c))):
            k = r["Kernel_Name"]
            if "qe::" not in k: continue
            k = k.replace("void ", "").replace("qe::", "").split("(")[0]
            d[k][c] += float(r["Counter_Value"]); n[(k, c)].add(r["Dispatch_Id"])
    out[wl] = {}
    for k, v in d.items():
        f = v["FETCH_SIZE"] / max(len(n[(k, "FETCH_SIZE")]), 1); w = v["WRITE_SIZE"] / max(len(n[(k, "WRITE_SIZE")]), 1)
        out[wl][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes": (2 * f + w) * 1024, "launches": len(n[(k, "FETCH_SIZE")])}
json.dump(out, open("$out/${tag}_pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
