#!/bin/bash
# kernel resource usage of one .hip file, compact: name VGPRs scratch occupancy sgprSpill vgprSpill   (usage: tools/kres.sh file.hip [grep-pattern])
cd "$(dirname "$0")/../probqa_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|Name:|VGPRs:|ScratchSize|Occupancy|Spill" | sed -E 's/.*(Name: |VGPRs: |ScratchSize \[bytes\/lane\]: |Occupancy \[waves\/SIMD\]: |SGPRs Spill: |VGPRs Spill: )//; s/ \[-Rpass.*//' | paste - - - - - - |
  awk '{n=$1; gsub(/_ZN3pqa[0-9]*/,"",n); printf "%-60s v=%s scr=%s occ=%s sS=%s vS=%s\n", substr(n,1,60),$2,$3,$4,$5,$6}' | grep -E "${2:-.}"
