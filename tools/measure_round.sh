# round-end measurement set (run on the GPU box through gpurun): bench JSON, rocprofv3 kernel stats, PMC traffic passes --
# the fp32 headline and the bf16 (F16 storage) mode -- and the per-rank shards of every BASELINE configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
Q="--no-cpu-baseline --no-sampling --no-extra-legs"
INDM_BENCH_PROFILE=1 python3 $R/bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/final_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch -- python3 $R/bench.py --steps 1 --warmup 1 $Q > $O/final_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write -- python3 $R/bench.py --steps 1 --warmup 1 $Q > $O/final_write.log 2>&1
INDM_BENCH_PROFILE=1 python3 $R/bench.py --dtype bf16 --steps 10 --warmup 3 $Q > $O/final_bench_bf16.json 2> $O/final_bench_bf16.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 3 --warmup 1 $Q > $O/final_prof_bf16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch_bf16 -- python3 $R/bench.py --dtype bf16 --steps 1 --warmup 1 $Q > $O/final_fetch_bf16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write_bf16 -- python3 $R/bench.py --dtype bf16 --steps 1 --warmup 1 $Q > $O/final_write_bf16.log 2>&1
python3 $R/tools/config_bench.py > $O/final_config_bench.txt 2>&1
cut -c1-200 $O/final_bench.json; cut -c1-200 $O/final_bench_bf16.json; cat $O/final_config_bench.txt
