python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/thin_bench.py 2>&1 | grep -v amdgpu > gpurun_out/thin2.log
INDM_PROFILE_SHAPES=1 INDM_BENCH_PROFILE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/shapes2.json 2> gpurun_out/shapes2.err
cut -c1-300 gpurun_out/shapes2.json
