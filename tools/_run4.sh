python -m pytest tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5
python bench.py --steps 5 --warmup 2 > gpurun_out/nelbo1.json 2> gpurun_out/nelbo1.err
tail -3 gpurun_out/nelbo1.err; cat gpurun_out/nelbo1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['cpu_baseline'], d.get('nelbo'))"
