#!/bin/bash
# whole GPU suite under `pytest -n 3` N times (the condition in which the reproducibility test used to fail 2 - 3 times in 10)
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-5}
FIRST=${2:-1}   # number of the first run (so that a second call does not overwrite the logs of the first)
for i in $(seq "$FIRST" $((FIRST + N - 1))); do
  timeout 1200 python3 -m pytest tests -m gpu -n 3 -q -p no:cacheprovider > gpurun_out/suite_n3_$i.log 2>&1
  echo "run $i: $(tail -n 1 gpurun_out/suite_n3_$i.log)"
  grep "^FAILED" gpurun_out/suite_n3_$i.log
done
