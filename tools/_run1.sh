for t in 1024 768 512 1536; do
  INDM_SPLITK_TARGET=$t python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sk_$t.json 2> gpurun_out/sk_$t.err
done
INDM_SPLITK_TARGET=768 INDM_SPLITK_MAX=96 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sk_768_96.json 2> gpurun_out/sk_768_96.err
grep -h value gpurun_out/sk_*.json | cut -c1-120
