# per-kernel averages (rocprofv3 --kernel-trace --stats) of the sweep and of the fix behind it, fresh and late quiz states, with and without the watch
cd /tmp; export TMPDIR=/tmp
for shape in ${SHAPES:-1000x5x1000 4000x5x4000 10000x5x10000}; do
 for st in ${STATES:-fresh late}; do
  for o in ${OPTS:-pole_fix=1 pole_fix=0}; do
   rocprofv3 --kernel-trace --stats -S -d /tmp/pc_$$ -- python $GRAFT_REPO_ROOT/tools/pole_cost.py $shape $st 60 $o 2>&1 | grep -E "back to back|eval_questions_f64<|pole_fixup" | sed -E 's/pqa::\(anonymous namespace\):://; s/void pqa:://; s/\(pqa::[A-Za-z]+\)//; s/\| +KERNEL_DISPATCH//; s/ +/ /g' | cut -c1-150
  done
 done
done
