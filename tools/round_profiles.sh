# the round's evidence in one GPU call: counter passes (fabric bytes per step, per precision), kernel traces of the four workloads,
# and the default bench line.  usage (on the GPU box): MDM_COMMIT=<short hash> bash tools/round_profiles.sh <tag, e.g. r04>
TAG=${1:-r04}
bash tools/pmc_step.sh 1 gpurun_out/${TAG}_pmc_traffic_p1.json > gpurun_out/${TAG}_pmc_p1.log 2>&1 &&
bash tools/pmc_step.sh 3 gpurun_out/${TAG}_pmc_traffic_p3.json > gpurun_out/${TAG}_pmc_p3.log 2>&1 &&
bash tools/profile_step.sh ${TAG}_bf16 > /dev/null 2>&1 &&
bash tools/profile_step.sh ${TAG}_parity --precision 3 > /dev/null 2>&1 &&
bash tools/profile_step.sh ${TAG}_big --config big > /dev/null 2>&1 &&
bash tools/profile_step.sh ${TAG}_big16 --config big16 --batch 8 --precision 2 > /dev/null 2>&1 &&
( time python bench.py ) > gpurun_out/${TAG}_bench_default.log 2>&1
tail -3 gpurun_out/${TAG}_bench_default.log | cut -c1-400
