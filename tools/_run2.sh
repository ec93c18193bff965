cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc3a -- python3 $R/tools/gemm_only.py nt > $R/gpurun_out/pmc3a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc3b -- python3 $R/tools/gemm_only.py nt > $R/gpurun_out/pmc3b.log 2>&1
ls $R/gpurun_out/pmc3a/*/ | head; tail -3 $R/gpurun_out/pmc3b.log
