cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vendor -- python3 $R/tools/vendor_sgemm.py > $R/gpurun_out/vendor.log 2>&1
