# rocprofv3 kernel trace of the bench process -> per-kernel stats + per-dispatch timeline of one captured step
# usage (on the GPU box): bash tools/profile_step.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-graph --no-cpu-baseline --no-modes --no-other-configs "$@" > gpurun_out/prof_$TAG.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_$TAG 8 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python3 tools/step_timeline.py gpurun_out/prof_$TAG > gpurun_out/${TAG}_step_timeline.txt 2>&1
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
head -30 gpurun_out/${TAG}_kernel_stats.txt
