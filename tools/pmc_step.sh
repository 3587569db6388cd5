# fabric traffic of one sampling step (two rocprofv3 --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one) -> JSON
# usage (on the GPU box): bash tools/pmc_step.sh <out.json>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/r03_pmc_traffic.json}
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
ARGS="--steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-modes --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write $OUT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
