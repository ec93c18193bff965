# rocprofv3 kernel stats of the bf16 training step (H16 storage of the score network): gpurun_out/prof_bf16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
Q="--no-cpu-baseline --no-sampling --no-extra-legs"
rm -rf $O/prof_bf16
INDM_SCORE_GRAPH=${INDM_SCORE_GRAPH:-0} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 3 --warmup 1 $Q > $O/prof_bf16.log 2>&1
f=$(ls -t $O/prof_bf16/*/*kernel_stats.csv | head -1)
cp $f $O/prof_bf16_kernel_stats.csv
python3 $R/tools/kernel_classes.py $f 4
head -40 $f | cut -c1-230
