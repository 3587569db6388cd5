cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sh in "12544 512 512" "12544 1536 512"; do
rm -rf gpurun_out/pa gpurun_out/pb gpurun_out/pc
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pa --output-format csv -- python3 tools/x3_one.py $sh > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d gpurun_out/pb --output-format csv -- python3 tools/x3_one.py $sh > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pc --output-format csv -- python3 tools/x3_one.py $sh > /dev/null 2>&1
echo "shape $sh"
for p in a b c; do python3 tools/pmc_kernels.py gpurun_out/p$p | grep gemm_x3; done
done
rm -rf gpurun_out/pa gpurun_out/pb gpurun_out/pc
