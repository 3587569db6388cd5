"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel totals and per-shape GEMM breakdown."""
import csv, re, collections, sys, glob, os
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
def short(n):
    n = n.replace('void ', '').replace('mdm::(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n[:70]
rows = list(csv.DictReader(open(st)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU kernel time {tot/1e6:.2f} ms over {steps} forwards -> {tot/1e6/steps:.3f} ms / forward")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'%':>6s}")
for r in rows[:22]:
    print(f"{short(r['Name']):70s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.1f}")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(tr)):
    n = r['Kernel_Name']
    if 'gemm' not in n: continue
    mm = re.search(r'gemm\w*<[^>]*>', n) or re.search(r'\w*gemm\w*', n)  # (the pack kernels have no template arguments)
    k = mm.group(0)
    key = (k, int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Z']))
    agg[key][0] += 1; agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
gt = sum(v[1] for v in agg.values())
print(f"\nGEMM launches by (kernel, blocks, batch): total {gt/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {str(k):52s} calls={v[0]:5d} avg_us={v[1]/v[0]:8.1f} total_ms={v[1]/1e3:8.2f} {100*v[1]/gt:5.1f}%")
