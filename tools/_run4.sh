python -m pytest tests -x -q -m gpu 2>&1 | tail -3
INDM_BENCH_PROFILE=1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/wg4.json 2> gpurun_out/wg4.err
cut -c1-130 gpurun_out/wg4.json
