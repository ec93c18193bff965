# rocprofv3 kernel statistics of one BASELINE configuration's per-rank training step: usage config_profile.sh '#4' [f32|bf16]
# (9 steps per process; the score network's graph warm-up adds two eager passes of the network to the totals)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-#4}; D=${2:-f32}; T=$(echo $C | tr -d '#')
export INDM_CB_DTYPE=$D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg_prof_${T}_$D -- python3 $R/tools/config_bench.py "$C" > $R/gpurun_out/cfg_prof_${T}_$D.log 2>&1
find $R/gpurun_out/cfg_prof_${T}_$D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/cfg_${T}_${D}_kernel_stats.csv
grep "ms/step" $R/gpurun_out/cfg_prof_${T}_$D.log
python3 $R/tools/kernel_classes.py $R/gpurun_out/cfg_${T}_${D}_kernel_stats.csv 9
