cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc_a --output-format csv -- python3 tools/mlp_ko.py 0,48,41 only16 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d gpurun_out/pmc_b --output-format csv -- python3 tools/mlp_ko.py 0,48,41 only16 > gpurun_out/pmc_b.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_a > gpurun_out/r3_pmc_mlp_a.txt 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_b > gpurun_out/r3_pmc_mlp_b.txt 2>&1
cat gpurun_out/r3_pmc_mlp_a.txt gpurun_out/r3_pmc_mlp_b.txt
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
