# rocprofv3 kernel statistics of one BASELINE configuration's per-rank-shard step (run through gpurun):
# usage: config_profile.sh "#3" | "#4" | "#5"   (tools/config_bench.py runs the fp32 and the bf16 case of that configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$(echo "$1" | tr -d '#')
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/config_prof_$TAG -- python3 $R/tools/config_bench.py "$1" > $R/gpurun_out/config_prof_$TAG.log 2>&1
find $R/gpurun_out/config_prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/config_${TAG}_kernel_stats.csv
grep "^#" $R/gpurun_out/config_prof_$TAG.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/config_${TAG}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print(f"kernel time {tot:.0f} ms over both runs (2 x 9 steps); top 40 by time:")
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.1f} ms {float(r['Percentage']):5.2f}%  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
PY
