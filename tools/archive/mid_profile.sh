cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
Q="--no-cpu-baseline --no-sampling --no-extra-legs"
python3 $R/bench.py --steps 10 --warmup 3 $Q > $O/mid_bench.json 2> $O/mid_bench.err
cut -c1-300 $O/mid_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid_prof -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/mid_prof.log 2>&1
find $O/mid_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/mid_kernel_stats.csv
python3 $R/tools/kernel_classes.py $O/mid_kernel_stats.csv 4 2>&1 | tail -30
