set -x
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r6_S S 0 300 > gpurun_out/prof_r6_S.log 2>&1
bash tools/prof.sh r6_M M 0 30 > gpurun_out/prof_r6_M.log 2>&1
bash tools/prof.sh r6_L1 L1 0 2 > gpurun_out/prof_r6_L1.log 2>&1
bash tools/prof.sh r6_L1_b32 L1_b32 0 3 > gpurun_out/prof_r6_L1_b32.log 2>&1
# PMC pass of the long-row single-quiz sweep (eval_cluster_kernel had none)
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r6_cluster custom:eval_cluster python $GRAFT_REPO_ROOT/tools/f32_single_bench.py 2000 5 100000 10 > gpurun_out/prof_r6_cluster.log 2>&1
# the counter passes go into profiles/traffic.json HERE, on the box, before the bench lines are taken: they report whether the
# counters they quote belong to the kernel sources in the tree (the copy that is committed is made by the same script at home)
python tools/promote_profiles.py > gpurun_out/promote_on_box.log 2>&1
python bench.py > gpurun_out/r6_bench_S.json 2> gpurun_out/r6_bench_S.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_S_driver_command_steps20_warmup5.json 2> gpurun_out/r6_bench_Sd.err
python bench.py --config M > gpurun_out/r6_bench_M.json 2> gpurun_out/r6_bench_M.err
python bench.py --config L1 > gpurun_out/r6_bench_L1.json 2> gpurun_out/r6_bench_L1.err
python bench.py --config L1 --batch 32 > gpurun_out/r6_bench_L1_b32.json 2> gpurun_out/r6_bench_L1_b32.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/tools/quiz_loop_breakdown.py > $GRAFT_REPO_ROOT/gpurun_out/r6_quiz_loop.log 2>&1
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "command: rocprofv3 --kernel-trace --stats -- python tools/quiz_loop_breakdown.py   (bench.py's quiz loop, 600 quizzes)"; grep -v "^W2026\|^E2026" $GRAFT_REPO_ROOT/gpurun_out/r6_quiz_loop.log | tail -12; cat "$f"; } > $GRAFT_REPO_ROOT/gpurun_out/r6_quiz_loop_stats.txt
for spec in "10000 5 10000 30:f32_single_M" "2000 5 100000 10:long_rows_100000"; do
  a=${spec%%:*}; n=${spec##*:}
  rm -rf /tmp/pf; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o t -- python $GRAFT_REPO_ROOT/tools/f32_single_bench.py $a > /tmp/pf.log 2>&1
  f=$(find /tmp/pf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "command: rocprofv3 --kernel-trace --stats -- python tools/f32_single_bench.py $a   (a Float engine, then a Double engine)"; grep "single quiz" /tmp/pf.log; cat "$f"; } > $GRAFT_REPO_ROOT/gpurun_out/r6_${n}_stats.txt
done
# the threaded learner loop (quiz_loop_threads at 64 client threads): which kernels serve it
cd /tmp
rm -rf /tmp/pt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python $GRAFT_REPO_ROOT/tools/threads_probe.py 64 > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "command: rocprofv3 --kernel-trace --stats -- python tools/threads_probe.py 64   (native learner client, 64 threads on one engine, sampled selector, training)"; grep "^threads" /tmp/pt.log; cat "$f"; } > $GRAFT_REPO_ROOT/gpurun_out/r6_quiz_loop_threads64_stats.txt
echo refresh done
# the fix behind the sweeps (pole_kernels.hip) -- fresh and late quiz states, with and without the watch
cd $GRAFT_REPO_ROOT
bash tools/pole_cost.sh > gpurun_out/r6_pole_fixup_stats.txt 2>&1
# round 6: the gated fix (late argmax selections), ListTopTargets on the device, the re-routed 10241..16384-target rows, .kb I/O
python tools/gate_probe.py 1000x5x1000 2000x5x2000 4000x5x4000 10000x5x10000 > gpurun_out/r6_gated_fix_stats.txt 2>&1
python tools/top_targets_bench.py > gpurun_out/r6_top_targets_stats.txt 2>&1
set +x
{ for t in 10500 12000 14000 16000; do python tools/sweep_timing.py ${t}x5x${t} 2>/dev/null | tail -1; python tools/sweep_timing.py ${t}x5x${t} cluster_from=16384 2>/dev/null | tail -1; done; } > gpurun_out/r6_rows_10241_16384_stats.txt
{ python tools/kb_io_bench.py 10000x5x10000; python tools/kb_io_bench.py 3000x5x100000 f32; } > gpurun_out/r6_kb_io_stats.txt 2>&1
# long rows by answer count (two to five: eval_cluster_five_kernel; six: round 4's kernel); each twice, the engine of the first run of a process being the coldest
{ for k in 2 3 4 5 6; do python tools/f32_single_bench.py 2000 $k 100000 20 2>/dev/null; python tools/f32_single_bench.py 2000 $k 100000 20 2>/dev/null; done; } > gpurun_out/r6_long_rows_by_answers_stats.txt
echo refresh r6 done
