# A/B of ONE build under two environments on one box (run through gpurun): alternating short bench.py lines.
# usage: ab_env.sh "VAR=value" [bench args]   -- arm "base" runs with VAR=value set, arm "new" without it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
SET=$1; shift
Q="--no-cpu-baseline --no-sampling --no-extra-legs --steps 8 --warmup 3 $*"
for round in 1 2 3; do
  for arm in base new; do
    if [ $arm = base ]; then env $SET python3 $R/bench.py $Q > $O/abe_${arm}_$round.json 2> $O/abe_${arm}_$round.err
    else python3 $R/bench.py $Q > $O/abe_${arm}_$round.json 2> $O/abe_${arm}_$round.err; fi
    python3 - <<PY
import json
d = json.loads(open("$O/abe_${arm}_$round.json").read().strip().splitlines()[-1])
print("$arm ($SET)" if "$arm" == "base" else "new", "round $round:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms", flush=True)
PY
  done
done
