# counter passes over the two expert-GEMM shapes of the fp32-grade mode: streamed-weight bf16x3 kernel vs the tile kernel
# usage (on the GPU box): bash tools/pmc_x3_stream.sh > gpurun_out/r04_pmc_gemm_stream3.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sh in "50176 512 1024 0" "50176 1024 512 1"; do
rm -rf gpurun_out/pa gpurun_out/pb gpurun_out/pc gpurun_out/pd
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pa --output-format csv -- python3 tools/x3_stream_one.py $sh > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d gpurun_out/pb --output-format csv -- python3 tools/x3_stream_one.py $sh > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pc --output-format csv -- python3 tools/x3_stream_one.py $sh > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/pd --output-format csv -- python3 tools/x3_stream_one.py $sh > /dev/null 2>&1
echo "shape (M N K gelu) $sh"
for p in a b c d; do python3 tools/pmc_kernels.py gpurun_out/p$p | grep "gemm_stream3_kernel\|gemm_x3_kernel"; done
done
rm -rf gpurun_out/pa gpurun_out/pb gpurun_out/pc gpurun_out/pd
