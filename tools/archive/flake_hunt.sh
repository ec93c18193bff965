#!/bin/bash
# Flake hunting, first half (run on the GPU box through gpurun): the whole GPU suite under `pytest -n 3 -v` up to $1
# times; on the first failure of the bit-reproducibility test, replay the failing worker's exact test order serially
# (tools/flake_replay.py) -- alone (twice), then beside background load.  Logs: gpurun_out/hunt_*.log, replay_*.log.
cd "$GRAFT_REPO_ROOT" || exit 1
MAX=${1:-3}
NEEDLE=test_training_is_bit_reproducible
mkdir -p gpurun_out
for i in $(seq 1 "$MAX"); do
  timeout 1000 python3 -m pytest tests -m gpu -n 3 -v -p no:cacheprovider > gpurun_out/hunt_$i.log 2>&1
  tail -n 3 gpurun_out/hunt_$i.log
  if grep -q "^FAILED .*$NEEDLE" gpurun_out/hunt_$i.log; then
    echo "hunt $i: reproducibility failure captured"
    python3 tools/flake_replay.py gpurun_out/hunt_$i.log $NEEDLE gpurun_out/replay_serial_a
    python3 tools/flake_replay.py gpurun_out/hunt_$i.log $NEEDLE gpurun_out/replay_serial_b
    python3 tools/flake_replay.py gpurun_out/hunt_$i.log $NEEDLE gpurun_out/replay_load2 --load 2
    exit 0
  fi
done
echo "no reproducibility failure in $MAX whole-suite runs"
