#!/bin/bash
# kernel resource usage summary for eval_kernels.hip
cd /root/repo/probqa_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c eval_kernels.hip -o /tmp/e.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|    VGPRs:|VGPRs Spill|Occupancy|error" | sed 's/.*remark: //; s/\[-Rpass.*//; s/Function Name: _ZN3pqa//; s/EvalArgsE//; s/Occupancy \[waves\/SIMD\]/occ/' | paste - - - -
