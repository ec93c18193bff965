#!/bin/bash
# Stress of EVERY kernel with the "LDS partials -> barrier -> few threads fold" shape (tools/lds_fold_scan.py) under the load in
# which round 3's fused GroupNorm lost partial sums: the whole GPU suite under `pytest -n 3` in the background (an oversubscribed
# GPU: seven processes) next to four repetition checkers -- two fp32, two bf16 (F16 storage) -- each repeating one full-size
# forward + backward of the training loss at batch 4 and comparing every module's output and gradient bit for bit with its
# first repetition (tools/flake_pattern.py).  The fp32 arm runs the chain kernels, the fused GroupNorm forward / backward, the
# column reductions and the Lop column norms; the bf16 arm the F16 rows / Sin-level kernels.  usage: flake_audit.sh [seconds]
cd "$GRAFT_REPO_ROOT" || exit 1
SECS=${1:-330}
(timeout 1500 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/audit_bg.log 2>&1 &)
sleep 45
PIDS=""
for i in 1 2; do
  timeout $((SECS + 500)) python3 tools/flake_pattern.py "$SECS" audit_f32_$i f32 > gpurun_out/audit_f32_$i.log 2>&1 &
  PIDS="$PIDS $!"
  timeout $((SECS + 500)) python3 tools/flake_pattern.py "$SECS" audit_bf16_$i bf16 > gpurun_out/audit_bf16_$i.log 2>&1 &
  PIDS="$PIDS $!"
done
for p in $PIDS; do wait $p; echo "pid $p exit $?"; done
for f in f32_1 f32_2 bf16_1 bf16_2; do echo "== $f"; tail -n 2 gpurun_out/audit_$f.log; grep "^fwd\|^bwd\|^loss\|^all" gpurun_out/audit_$f.log | cut -c1-60 | sort | uniq -c | sort -rn | head -4; done
tail -n 3 gpurun_out/audit_bg.log
