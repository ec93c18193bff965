#!/bin/bash
# 2 x 2 A/B of the fused GroupNorm forward under the whole suite's `pytest -n 3` load, same load, same time:
#   serial (INDM_GN_FUSED=2)         round 3's kernel: serial fold of 256 LDS partials, arguments loaded where first used
#   serialpin (INDM_GN_FUSED=3)      the same fold, every kernel argument loaded at the kernel's first instruction
#   butterfly (default)              round 4's kernel: wave butterflies, arguments loaded first
# next to three copies of the late-kernel-argument probe (tools/probe/kernarg_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
SECS=${1:-150}
(cd tools/probe && hipcc --offload-arch=gfx950 -O3 kernarg_probe.hip -o kernarg_probe > /dev/null 2>&1)
(timeout 1000 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/ab_bg.log 2>&1 &)
sleep 45
PIDS=""
INDM_GN_FUSED=2 timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" serial f32 > gpurun_out/ab_serial.log 2>&1 &
PIDS="$PIDS $!"
INDM_GN_FUSED=3 timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" serialpin f32 > gpurun_out/ab_serialpin.log 2>&1 &
PIDS="$PIDS $!"
timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" butterfly f32 > gpurun_out/ab_butterfly.log 2>&1 &
PIDS="$PIDS $!"
for t in 1 2 3; do
  timeout $((SECS + 100)) tools/probe/kernarg_probe "$SECS" $t 30 > gpurun_out/ab_kernarg$t.log 2>&1 &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p; echo "pid $p exit $?"; done
for f in serial serialpin butterfly; do tail -n 1 gpurun_out/ab_$f.log; grep "^fwd\|^bwd\|^loss\|^all" gpurun_out/ab_$f.log | cut -c1-60 | sort | uniq -c | sort -rn | head -4; done
cat gpurun_out/ab_kernarg*.log
free -g | head -2
