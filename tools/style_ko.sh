# Knock-out timings of the fused stylization launch (csrc/style_gemm.hip, 16-bit form) inside real sampling steps: kernel-trace
# averages of style_gemm_kernel under the DIAGNOSTIC library's knobs 74..77 (wrong results, timing only).
# usage (on the GPU box, after `python motiondiffusion-moe_amd/build.py --diag`): bash tools/style_ko.sh > gpurun_out/r04_style_gemm_knockouts.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MDM_LIB=$GRAFT_REPO_ROOT/motiondiffusion-moe_amd/libmdm_hip_diag.so
for v in 0 74 75 76 77; do
  rm -rf gpurun_out/sk
  rocprofv3 --kernel-trace --stats -d gpurun_out/sk --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-modes --no-other-configs --variant $v > /dev/null 2>&1
  case $v in 0) n="full";; 74) n="no row phase";; 75) n="no K loop";; 76) n="no output stores";; 77) n="no weight refills";; esac
  python3 - "$n" <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/sk/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "style_gemm_kernel" in r["Kernel_Name"]:
        d[int(r["Grid_Size_X"]) // 512].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{sys.argv[1]:18s}", "  ".join(f"grid {g}: {sorted(v)[len(v) // 2]:.1f} us median of {len(v)}" for g, v in sorted(d.items())))
PY
done
rm -rf gpurun_out/sk
