# A/B of two builds of the kernel library on ONE box (run through gpurun): for each library the dense / thin micro-benchmarks
# and a short bench.py line, alternating.  usage: ab_libs.sh <base.so> <new.so> [bench args]; output under gpurun_out/ab_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
BASE=$1; NEW=$2; shift 2
Q="--no-cpu-baseline --no-sampling --no-extra-legs --steps 8 --warmup 3 $*"
for round in 1 2; do
  for arm in base new; do
    if [ $arm = base ]; then L=$BASE; else L=$NEW; fi
    INDM_HIP_LIB=$R/$L python3 $R/bench.py $Q > $O/ab_bench_${arm}_$round.json 2> $O/ab_bench_${arm}_$round.err
    python3 - <<EOF
import json
d = json.loads(open("$O/ab_bench_${arm}_$round.json").read().strip().splitlines()[-1])
print("$arm round $round:", d["value"], "img/s", d["ms_per_step"], "ms", "dominant", d["roofline"]["achieved"], flush=True)
EOF
  done
done
for arm in base new; do
  if [ $arm = base ]; then L=$BASE; else L=$NEW; fi
  INDM_HIP_LIB=$R/$L python3 $R/tools/dense_bench.py > $O/ab_dense_$arm.txt 2>&1
  INDM_HIP_LIB=$R/$L python3 $R/tools/thin_bench.py > $O/ab_thin_$arm.txt 2>&1
done
paste -d'|' <(cut -c1-90 $O/ab_dense_base.txt) <(cut -c58-90 $O/ab_dense_new.txt)
paste -d'|' <(cut -c1-100 $O/ab_thin_base.txt) <(cut -c64-100 $O/ab_thin_new.txt)
