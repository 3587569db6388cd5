"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) into fabric bytes per sampling
step, following /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB and
count L2 <-> fabric requests (Infinity-Cache hits included); on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it
is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

One step = the dispatches between two consecutive cfg_step_kernel launches (the last complete step of the run)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "cfg_step" in r["Kernel_Name"]]
    return rows[ends[-2] + 1:ends[-1] + 1]


def short(n):
    n = n.replace("void ", "").replace("mdm::(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:48]


def provenance():
    """(commit, hash of the kernel sources): bench.py prints a committed profile only when the hash matches the sources it runs"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    import bench
    return commit or os.environ.get("MDM_COMMIT", "unknown (no git on the GPU box: set MDM_COMMIT)"), bench.source_sha16()


def main():
    fd, wd, out = sys.argv[1:4]
    mode = sys.argv[4] if len(sys.argv) > 4 else "precision=1 (bf16)"
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    tot = {}
    for d, counter, scale, slot in ((fd, "FETCH_SIZE", 2.0 * 1024, 1), (wd, "WRITE_SIZE", 1024.0, 2)):
        rows = load(d, counter)
        tot[counter] = sum(float(r["Counter_Value"]) for r in rows) * scale
        for r in rows:
            k = short(r["Kernel_Name"])
            per[k][slot] += float(r["Counter_Value"]) * scale
            if slot == 1:
                per[k][0] += 1
    commit, sha = provenance()
    res = {
        "commit": commit, "src_sha16": sha,
        "what": "L2<->fabric bytes per CFG step (one forward of 2B=64 rows + sampler update), small-E8, B=32, T=196, " + mode,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --warmup 1 --no-graph), "
                  "summed over the dispatches of the last complete step; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                  "tallies 128-B read requests at 64 B); Infinity-Cache hits are included in these fabric counters; "
                  "aggregated by tools/pmc_traffic.py",
        "fetch_bytes_per_step": tot["FETCH_SIZE"], "write_bytes_per_step": tot["WRITE_SIZE"],
        "total_bytes_per_step": tot["FETCH_SIZE"] + tot["WRITE_SIZE"],
        "per_kernel_MB_per_call": {k: {"calls": v[0], "fetch_x2": round(v[1] / max(v[0], 1) / 1e6, 1),
                                       "write": round(v[2] / max(v[0], 1) / 1e6, 1)}
                                   for k, v in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:16]},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(f"fetch {tot['FETCH_SIZE'] / 1e9:.2f} GB + write {tot['WRITE_SIZE'] / 1e9:.2f} GB per step -> {out}")


if __name__ == "__main__":
    main()
