# A/B of two builds of the kernel library in bf16 mode on ONE box: dense micro-benchmarks with bf16 MFMA inputs and a short
# `bench.py --dtype bf16` line per library, alternating.  usage: ab_bf16.sh <base.so> <new.so>; output under gpurun_out/ab16_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
BASE=$1; NEW=$2
Q="--dtype bf16 --no-cpu-baseline --no-sampling --no-extra-legs --steps 8 --warmup 3"
for arm in base new; do
  if [ $arm = base ]; then L=$BASE; else L=$NEW; fi
  INDM_DENSE_BENCH_DTYPE=bf16 INDM_HIP_LIB=$R/$L python3 $R/tools/dense_bench.py > $O/ab16_dense_$arm.txt 2>&1
done
paste -d'|' <(cut -c1-90 $O/ab16_dense_base.txt) <(cut -c58-90 $O/ab16_dense_new.txt) | head -36
for round in 1 2; do
  for arm in base new; do
    if [ $arm = base ]; then L=$BASE; else L=$NEW; fi
    INDM_HIP_LIB=$R/$L python3 $R/bench.py $Q > $O/ab16_bench_${arm}_$round.json 2> $O/ab16_bench_${arm}_$round.err
    python3 - <<PY
import json
d = json.loads(open("$O/ab16_bench_${arm}_$round.json").read().strip().splitlines()[-1])
print("$arm round $round:", d["value"], "img/s", d["ms_per_step"], "ms", flush=True)
PY
  done
done
