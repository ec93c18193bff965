python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/dense_bench.py 2>&1 | grep -E "3->512|3 -> 512"
INDM_BENCH_PROFILE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2> gpurun_out/b6.err | cut -c1-200
grep -E "thin_conv|thin_k" gpurun_out/b6.err
