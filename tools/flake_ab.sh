#!/bin/bash
# A/B of the fused GroupNorm forward under the whole suite's `pytest -n 3` load: round 3's serial fold (INDM_GN_FUSED=2)
# against the butterfly form (default), two pattern processes each, same load, same time
cd "$GRAFT_REPO_ROOT" || exit 1
SECS=${1:-150}
(timeout 1000 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/ab_bg.log 2>&1 &)
sleep 45
PIDS=""
for i in 1; do
  INDM_GN_FUSED=2 timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" serial$i f32 > gpurun_out/ab_serial$i.log 2>&1 &
  PIDS="$PIDS $!"
  timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" butterfly$i f32 > gpurun_out/ab_butterfly$i.log 2>&1 &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p; echo "pid $p exit $?"; done
for f in serial1 butterfly1; do tail -n 1 gpurun_out/ab_$f.log; grep "^fwd\|^bwd\|^loss\|^all" gpurun_out/ab_$f.log | cut -c1-60 | sort | uniq -c | sort -rn | head -4; done
sleep 200
tail -n 2 gpurun_out/ab_bg.log
