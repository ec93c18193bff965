#!/bin/bash
# A/B of the fused GroupNorm forward under the whole suite's `pytest -n 3` load (an OVERSUBSCRIBED GPU: seven processes),
# same load, same time, two processes per arm:
#   serial (INDM_GN_FUSED=2)     round 3's kernel: 256 per-thread partials in LDS, barrier, GS threads fold them serially
#   butterfly (default)          round 4's kernel: wave butterflies + one LDS word per (wave, group), arguments loaded first
# (INDM_GN_FUSED=3 = the serial fold with its arguments pinned: it failed like the serial arm, which rules the late
# argument loads out; tools/probe/kernarg_probe and lds_barrier_probe found nothing under the same load)
cd "$GRAFT_REPO_ROOT" || exit 1
SECS=${1:-200}
(timeout 1200 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/ab_bg.log 2>&1 &)
sleep 45
PIDS=""
for i in 1 2; do
  INDM_GN_FUSED=2 timeout $((SECS + 500)) python3 tools/flake_pattern.py "$SECS" serial$i f32 > gpurun_out/ab_serial$i.log 2>&1 &
  PIDS="$PIDS $!"
  timeout $((SECS + 500)) python3 tools/flake_pattern.py "$SECS" butterfly$i f32 > gpurun_out/ab_butterfly$i.log 2>&1 &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p; echo "pid $p exit $?"; done
for f in serial1 serial2 butterfly1 butterfly2; do tail -n 1 gpurun_out/ab_$f.log; grep "^fwd\|^bwd\|^loss\|^all" gpurun_out/ab_$f.log | cut -c1-60 | sort | uniq -c | sort -rn | head -4; done
