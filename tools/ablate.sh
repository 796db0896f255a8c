#!/bin/bash
# usage: tools/ablate.sh <config> <variant> "<abl list>"   -- builds ablated libraries on the GPU box and times the sweep
CFG=$1; VAR=$2; mkdir -p gpurun_out /tmp/abl
for a in $3; do
  (cd probqa_amd/csrc && for f in eval_kernels select_kernels prior_kernels kb_kernels; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DPQA_ABL=$a -c $f.hip -o /tmp/abl/$f.o 2>/dev/null; done
   for f in hip_engine c_abi; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -x hip -c $f.cpp -o /tmp/abl/$f.o 2>/dev/null; done
   hipcc --offload-arch=gfx950 -shared -o /tmp/abl/libPqaCore_abl$a.so /tmp/abl/*.o)
  PQACORE_LIB=/tmp/abl/libPqaCore_abl$a.so python bench.py --config $CFG --variant $VAR --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('abl=$a', d['config']['eval_kernel'], 'kernel_us=%.1f'%d['roofline']['kernel_us'])"
done
