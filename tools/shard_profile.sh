# rocprofv3 kernel statistics of the per-rank-shard training step (run through gpurun): usage shard_profile.sh [f32|bf16] [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=${1:-f32}; B=${2:-16}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shard_prof_${D}_$B -- python3 $R/tools/shard_bench.py $D $B > $R/gpurun_out/shard_prof_${D}_$B.log 2>&1
find $R/gpurun_out/shard_prof_${D}_$B -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/shard_${D}_${B}_kernel_stats.csv
tail -2 $R/gpurun_out/shard_prof_${D}_$B.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/shard_${D}_${B}_kernel_stats.csv")))
steps = 11.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
n = sum(int(r["Calls"]) for r in rows) / steps
print(f"kernel time {tot:.1f} ms/step over {n:.0f} launches/step")
rows.sort(key=lambda r: -int(r["Calls"]))
for r in rows[:45]:
    print(f"{int(r['Calls'])/steps:7.1f}/step  avg {float(r['AverageNs'])/1e3:7.1f} us  {float(r['TotalDurationNs'])/1e6/steps:6.2f} ms/step  {r['Name'][:110]}")
PY
