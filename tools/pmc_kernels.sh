# Three rocprofv3 --pmc passes (8 SQ counters each; no trace domain besides --kernel-trace) over a micro-benchmark, then
# tools/pmc_json.py -> per-kernel averages + derived shares.  usage (through gpurun): pmc_kernels.sh <tag> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
S=$1; shift
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_1 -- python3 $R/$S "$@" > $R/gpurun_out/${TAG}_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_2 -- python3 $R/$S "$@" > $R/gpurun_out/${TAG}_2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/${TAG}_3 -- python3 $R/$S "$@" > $R/gpurun_out/${TAG}_3.log 2>&1
python3 $R/tools/pmc_json.py $R/gpurun_out/${TAG} > $R/gpurun_out/${TAG}.json
