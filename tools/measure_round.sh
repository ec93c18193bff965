# round-end measurement set (run on the GPU box through gpurun): bench JSON, rocprofv3 kernel stats, PMC traffic passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
INDM_BENCH_PROFILE=1 python3 $R/bench.py > $R/gpurun_out/final_bench.json 2> $R/gpurun_out/final_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sampling > $R/gpurun_out/final_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sampling > $R/gpurun_out/final_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sampling > $R/gpurun_out/final_write.log 2>&1
cat $R/gpurun_out/final_bench.json | cut -c1-200
