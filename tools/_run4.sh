python -m pytest tests -x -q -m gpu 2>&1 | tail -3
INDM_BENCH_PROFILE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/big1.json 2> gpurun_out/big1.err
INDM_BIG_TILE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/big0.json 2> gpurun_out/big0.err
cut -c1-130 gpurun_out/big1.json gpurun_out/big0.json
