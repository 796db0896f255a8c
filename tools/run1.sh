./tools/ubench_fp64 > gpurun_out/ubench.txt 2>&1
python bench.py --steps 500 --warmup 50 --cpu-seconds 2 > gpurun_out/bench_S.json 2>gpurun_out/bench_S.err
for v in 1 2 8; do python bench.py --steps 300 --warmup 20 --variant $v --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['eval_kernel'], d['roofline']['kernel_us'], d['value'], d['pipelined_selections_per_sec'])"; done > gpurun_out/variants_S.txt 2>&1
python tools/relerr.py > gpurun_out/relerr.txt 2>&1
