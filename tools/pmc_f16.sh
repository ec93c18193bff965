# PMC passes of the F16 kernels (run on the GPU box through gpurun): the microbenchmark of tools/f16_bench.py under three counter sets
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-pmcf16}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_1 -- python3 $R/tools/f16_bench.py 128 > $R/gpurun_out/${TAG}_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_2 -- python3 $R/tools/f16_bench.py 128 > $R/gpurun_out/${TAG}_2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $R/gpurun_out/${TAG}_3 -- python3 $R/tools/f16_bench.py 128 > $R/gpurun_out/${TAG}_3.log 2>&1
for i in 1 2 3; do python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_$i "f16_rows_kernel<2, 1, false, 64>"; done
