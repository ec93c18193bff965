# Effective shader clock under load (run on the GPU box through gpurun): GRBM_GUI_ACTIVE (GFX-clock cycles while the
# kernel runs) next to the kernel's duration from the same trace -> cycles / ns = GHz.  Two programs: the pure-MFMA probe
# (no memory traffic) and the chain kernel (MFMA + 800 MB of HBM traffic per launch).   usage: pmc_clock.sh [TAG]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-clock}
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_mfma -- $R/tools/probe/mfma_peak 3 > $R/gpurun_out/${TAG}_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_chain -- python3 $R/tools/gemm_only.py chain > $R/gpurun_out/${TAG}_chain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_nng -- python3 $R/tools/gemm_only.py nng > $R/gpurun_out/${TAG}_nng.log 2>&1
python3 $R/tools/pmc_clock.py $R/gpurun_out ${TAG}
