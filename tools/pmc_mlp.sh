# rocprofv3 counter passes over the fused expert MLP (tools/mlp_ko.py): bash tools/pmc_mlp.sh [knobs]   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=${1:-0}
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_a --output-format csv -- python3 tools/mlp_ko.py $K only16 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d gpurun_out/pmc_b --output-format csv -- python3 tools/mlp_ko.py $K only16 > gpurun_out/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d gpurun_out/pmc_c --output-format csv -- python3 tools/mlp_ko.py $K only16 > gpurun_out/pmc_c.log 2>&1
for p in a b c; do python3 tools/pmc_kernels.py gpurun_out/pmc_$p | grep fused_mlp; done
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
