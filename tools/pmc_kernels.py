"""Per-kernel means of rocprofv3 --pmc counters from a counter_collection.csv directory: python tools/pmc_kernels.py DIR"""
import csv, glob, collections, sys
d = sys.argv[1]
fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not fs:
    raise SystemExit(f"{d}: no counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].replace("void ", "").replace("mdm::(anonymous namespace)::", "")
    k = k.split("(")[0][:56]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:12]:
    print(f"{k:50s}", {c: round(v / max(1, cnt[(k, c)])) for c, v in agg[k].items()}, "launches", max(cnt[(k, c)] for c in agg[k]))
