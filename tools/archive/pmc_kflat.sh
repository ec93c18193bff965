# PMC passes of the thin im2col convolution (run on the GPU box through gpurun): tools/kflat_probe.py --one <tune> under three counter sets
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export INDM_THIN_TUNE
for TUNE in ${@:-0 8192}; do
INDM_THIN_TUNE=$TUNE
TAG=pmckflat_$TUNE
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_1 -- python3 $R/tools/kflat_probe.py 128 --one $TUNE > $R/gpurun_out/${TAG}_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_2 -- python3 $R/tools/kflat_probe.py 128 --one $TUNE > $R/gpurun_out/${TAG}_2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_3 -- python3 $R/tools/kflat_probe.py 128 --one $TUNE > $R/gpurun_out/${TAG}_3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_4 -- python3 $R/tools/kflat_probe.py 128 --one $TUNE > $R/gpurun_out/${TAG}_4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_5 -- python3 $R/tools/kflat_probe.py 128 --one $TUNE > $R/gpurun_out/${TAG}_5.log 2>&1
echo "=== tune $TUNE geometry ${KFLAT_GEOM:-3,32} (FETCH_SIZE / WRITE_SIZE in KB per launch; FETCH_SIZE x 2 on gfx950 for wide reads)"
for i in 1 2 3 4 5; do python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_$i "thin_conv"; done
done
