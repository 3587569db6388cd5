# per-kernel LDS bank conflicts and MFMA busy of the bench step (two rocprofv3 --pmc passes) -> text
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# usage (on the GPU box): bash tools/pmc_lds_mfma.sh <out.txt> [<precision>]
OUT=${1:-gpurun_out/r04_pmc_lds_mfma.txt}
P=${2:-1}
ARGS="--precision $P --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-modes --no-other-configs"
rm -rf gpurun_out/pmc_l gpurun_out/pmc_m
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_l --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_l.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d gpurun_out/pmc_m --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_m.log 2>&1
{ echo "# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (per-kernel means over the launches of 3 eager steps, bench.py --no-graph, precision $P)"; python3 tools/pmc_kernels.py gpurun_out/pmc_l; echo; echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (same run shape; MFMA busy is summed over the 4 SIMDs of a CU)"; python3 tools/pmc_kernels.py gpurun_out/pmc_m; } > $OUT
rm -rf gpurun_out/pmc_l gpurun_out/pmc_m
cat $OUT
