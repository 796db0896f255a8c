#!/bin/bash
# Same-box A/B against an earlier tree.  Boxes differ by several per cent (fp64-heavy kernels by more), so a number from another gpurun
# call says little; two regressions of round 5 (LDS priors on 8-byte addresses: M +38 %; a max per element in the long-row sweep's
# pass 2: +30 %) went unnoticed for hours because each was read as "a slow box".
#   tools/ab_against.sh <commit>          at home: build <commit>'s tree under _prev/ (git-ignored, travels with gpurun)
#   gpurun -- 'bash tools/ab_against.sh'  on the GPU box: the same measurements on _prev/ and on the tree, alternating
set -e
cd "$(dirname "$0")/.."
if [ -n "$1" ]; then
  rm -rf _prev && mkdir _prev && git archive "$1" | tar -x -C _prev
  (cd _prev/probqa_amd/csrc && make -j8 > /dev/null) && (cd _prev/oracle && make > /dev/null 2>&1 || true)
  ls -la _prev/probqa_amd/libPqaCore.so
  exit 0
fi
[ -d _prev ] || { echo "no _prev/: run tools/ab_against.sh <commit> at home first"; exit 1; }
for i in 1 2; do
  for c in _prev .; do
    (cd $c && python tools/sweep_timing.py 500x5x500 1000x5x1000 2000x5x2000 4000x5x4000 8000x5x8000 10000x5x10000 2>/dev/null | sed "s|^|$c |")
    (cd $c && python tools/f32_single_bench.py 2000 5 100000 10 2>/dev/null | sed "s|^|$c |")
    (cd $c && python tools/f32_single_bench.py 10000 5 10000 20 2>/dev/null | sed "s|^|$c |")
    (cd $c && python bench.py --no-points 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$c', 'S: value', round(d['value']), 'resident step', round(r['resident_step_us']['mean'], 2), 'launched kernel', round(r['launched_kernel']['kernel_us'], 2),
      'launch_per_selection', round(d['launch_per_selection']['selections_per_sec']), 'quiz_loop', round((d.get('quiz_loop') or {}).get('questions_per_sec', 0)))")
  done
done
