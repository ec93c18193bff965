cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU"
for cfg in "big nt 1" "old nt 0" "ven vnt 1"; do set -- $cfg
  INDM_BIG_TILE=$3 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc4_$1 -- python3 $R/tools/gemm_only.py $2 4096 4096 4096 > $R/gpurun_out/pmc4_$1.log 2>&1
done
ls $R/gpurun_out/pmc4_big/*/
