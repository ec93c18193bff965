#!/bin/bash
# discriminating experiment for the reproducibility flake: under the whole suite's `pytest -n 3` load (oversubscribed GPU)
# run (a) the LDS / barrier integrity probe, (b) the pattern tool on the fused GroupNorm, (c) the pattern tool with the
# three-kernel GroupNorm (INDM_GN_FUSED=0)
cd "$GRAFT_REPO_ROOT" || exit 1
(cd tools/probe && hipcc --offload-arch=gfx950 -O3 lds_barrier_probe.hip -o lds_barrier_probe > /dev/null 2>&1)
(timeout 1000 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/probe_bg.log 2>&1 &)
sleep 45
timeout 300 python3 tools/flake_pattern.py 120 fused f32 > gpurun_out/probe_pattern_fused.log 2>&1 &
P1=$!
INDM_GN_FUSED=0 timeout 300 python3 tools/flake_pattern.py 120 three f32 > gpurun_out/probe_pattern_three.log 2>&1 &
P2=$!
timeout 200 tools/probe/lds_barrier_probe 90 2 8 > gpurun_out/probe_lds_r2.log 2>&1 &
P3=$!
timeout 200 tools/probe/lds_barrier_probe 90 256 1 > gpurun_out/probe_lds_r256.log 2>&1 &
P4=$!
wait $P1 $P2 $P3 $P4
for f in fused three; do echo "== $f"; grep -c "FIRST" gpurun_out/probe_pattern_$f.log; grep "FIRST" -A1 gpurun_out/probe_pattern_$f.log | grep "^fwd\|^bwd\|^loss\|^all" | cut -c1-80 | sort | uniq -c | sort -rn | head -8; tail -n 1 gpurun_out/probe_pattern_$f.log; done
cat gpurun_out/probe_lds_r2.log gpurun_out/probe_lds_r256.log
# alone (the load is gone by now or soon): the same probe without co-tenants
sleep 240
tail -n 2 gpurun_out/probe_bg.log
timeout 100 tools/probe/lds_barrier_probe 40 2 8
