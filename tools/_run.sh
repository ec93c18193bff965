python -m pytest tests -x -q -m gpu 2>&1 | tail -3
INDM_BENCH_PROFILE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b3.json 2> gpurun_out/b3.err
cut -c1-200 gpurun_out/b3.json
