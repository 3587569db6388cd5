# fabric traffic of one sampling step (two rocprofv3 --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one) -> JSON
# usage (on the GPU box): MDM_COMMIT=<short hash> bash tools/pmc_step.sh <precision> [<out.json>]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=${1:-1}
OUT=${2:-gpurun_out/r04_pmc_traffic_p$P.json}
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
ARGS="--precision $P --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-modes --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write $OUT "precision=$P"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
