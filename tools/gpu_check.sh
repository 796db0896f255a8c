#!/bin/bash
# usage: tools/gpu_check.sh "<S variants>" "<M variants>"   -- runs GPU tests, then compact bench lines per variant
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.txt
fmt='import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line); r=d["roofline"]
    print("%-24s kernel_us=%9.2f GB/s=%8.1f frac=%.3f sync_sel/s=%9.1f pipelined=%s" % (d["config"]["eval_kernel"], r["kernel_us"], r["achieved"], r["frac"], d["value"], d["pipelined_selections_per_sec"]))'
: > gpurun_out/variants.txt
for v in $1; do python bench.py --steps 300 --warmup 20 --variant $v --no-cpu-baseline 2>>gpurun_out/bench.err | python -c "$fmt" >> gpurun_out/variants.txt; done
for v in $2; do python bench.py --config M --steps 20 --warmup 3 --variant $v --no-cpu-baseline 2>>gpurun_out/bench.err | python -c "$fmt" >> gpurun_out/variants.txt; done

