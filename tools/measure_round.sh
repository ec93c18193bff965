# round-end measurement set (run on the GPU box through gpurun): bench JSON, rocprofv3 kernel stats, PMC traffic passes --
# the fp32 headline and the bf16 (F16 storage) mode -- and the per-rank shards of every BASELINE configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
Q="--no-cpu-baseline --no-sampling --no-extra-legs"
INDM_BENCH_PROFILE=1 python3 $R/bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/final_prof.log 2>&1
INDM_SCORE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch -- python3 $R/bench.py --steps 1 --warmup 1 $Q > $O/final_fetch.log 2>&1
INDM_SCORE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write -- python3 $R/bench.py --steps 1 --warmup 1 $Q > $O/final_write.log 2>&1
INDM_BENCH_PROFILE=1 python3 $R/bench.py --dtype bf16 --steps 10 --warmup 3 $Q > $O/final_bench_bf16.json 2> $O/final_bench_bf16.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 3 --warmup 1 $Q > $O/final_prof_bf16.log 2>&1
INDM_SCORE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_fetch_bf16 -- python3 $R/bench.py --dtype bf16 --steps 1 --warmup 1 $Q > $O/final_fetch_bf16.log 2>&1
INDM_SCORE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_write_bf16 -- python3 $R/bench.py --dtype bf16 --steps 1 --warmup 1 $Q > $O/final_write_bf16.log 2>&1
python3 $R/tools/config_bench.py > $O/final_config_bench.txt 2>&1
cut -c1-200 $O/final_bench.json; cut -c1-200 $O/final_bench_bf16.json; cat $O/final_config_bench.txt
# round 6: launch histograms of the 16-image shard (fp32 and bf16), the bf16 step's eager kernel classes, steady-state copy trace,
# H16 micro-benchmarks
bash $R/tools/shard_profile.sh f32 16 > $O/final_shard_f32_16.txt 2>&1
bash $R/tools/shard_profile.sh bf16 16 > $O/final_shard_bf16_16.txt 2>&1
python3 $R/tools/h16_bench.py 128 > $O/final_h16_bench.txt 2>&1
python3 $R/tools/h16_gn_bench.py 128 0.0 > $O/final_h16_gn_bench.txt 2>&1
python3 $R/tools/f16_bench.py > $O/final_f16_bench.txt 2>&1
python3 $R/tools/copy_trace.py > $O/final_copy_trace_f32.txt 2>&1
INDM_TRACE_DTYPE=bf16 python3 $R/tools/copy_trace.py > $O/final_copy_trace_bf16.txt 2>&1
tail -3 $O/final_shard_f32_16.txt | head -1; grep "kernel time" $O/final_shard_f32_16.txt $O/final_shard_bf16_16.txt
