# A/B of the chain kernel and the thin weight grads on one box: <base.so> vs the in-tree library (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
BASE=$1
for arm in base new base new; do
  if [ $arm = base ]; then L=$R/$BASE; else L=$R/indm_amd/libindm_hip.so; fi
  echo "== $arm"; INDM_HIP_LIB=$L python3 $R/tools/chain_bench.py 2>&1 | grep -v amdgpu.ids | grep -i "fused\|chain\|us" | head -8
done
for s in 0 256 512 1024; do echo "== thin wgrad stream split $s"; INDM_THIN_WGRAD_STREAM=$s python3 $R/tools/thin_bench.py 2>&1 | grep "wgrad" | head -3; done
