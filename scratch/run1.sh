python -m pytest tests -m gpu -x -q -k "conv or networks or modconv" 2>&1 | tail -2
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/s18_bench.json 2>/dev/null; cat gpurun_out/s18_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['target_kernel'])"
