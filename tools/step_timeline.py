"""Per-dispatch timeline of ONE captured sampling step from a rocprofv3 --kernel-trace CSV: prints every kernel of the
last complete step in launch order with its duration, so each GEMM can be attributed to its call site."""
import csv, glob, os, re, sys

d = sys.argv[1]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "").replace("mdm::(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:60]


# one step ends with the cfg_step kernel
ends = [i for i, r in enumerate(rows) if "cfg_step" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
print(f"{b - a} dispatches in the step, wall {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
busy = 0.0
for i, r in enumerate(rows[a:b]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  grid={int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{r['Grid_Size_Z']:>3s} {short(r['Kernel_Name'])}")
print(f"sum of kernel durations {busy:.1f} us")
