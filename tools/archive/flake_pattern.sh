#!/bin/bash
# pattern hunt: the whole suite under `pytest -n 3` as background load + two copies of tools/flake_pattern.py
cd "$GRAFT_REPO_ROOT" || exit 1
SECS=${1:-420}
(timeout 1000 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/pattern_bg.log 2>&1 &)
sleep 45
timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" pf32 f32 > gpurun_out/pattern_f32.log 2>&1 &
P1=$!
timeout $((SECS + 200)) python3 tools/flake_pattern.py "$SECS" pbf16 bf16 > gpurun_out/pattern_bf16.log 2>&1 &
P2=$!
wait $P1 $P2
tail -n 30 gpurun_out/pattern_f32.log gpurun_out/pattern_bf16.log
sleep 5
tail -n 3 gpurun_out/pattern_bg.log
