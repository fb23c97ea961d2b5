"""per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_report.py FILE... [-- name substrings]"""
import csv, collections, sys
args = sys.argv[1:]
names = ("quad", "sys", "traceback", "k_windowed(")
if "--" in args:
    names = tuple(args[args.index("--") + 1:]); args = args[:args.index("--")]
for f in args:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(n in k for n in names):
            print(k, {c.replace("SQ_", ""): round(sum(x) / len(x)) for c, x in v.items()})
