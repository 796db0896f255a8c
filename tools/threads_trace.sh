#!/bin/bash
# usage: tools/threads_trace.sh <threads>     -- rocprofv3 kernel trace of the threaded learner loop: how busy the device is,
# per kernel, while <threads> client threads run quizzes (gpurun_out/threads_trace/summary.txt)
NT=${1:-64}
OUT=$PWD/gpurun_out/threads_trace
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/tools/threads_probe.py $NT > $OUT/probe.txt 2> $OUT/trace.err
cd - > /dev/null
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last run of the probe: the kernels after the longest gap in the second half do not matter -- take the busiest window
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
print(open("$OUT/probe.txt").read())
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][0] += 1
    acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("trace window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
for k, (n, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-60s %7d launches  %9.1f us avg  %8.2f ms total" % (k, n, ns / n / 1e3, ns / 1e6))
# busy fraction of the last 150 ms of the trace (the multi-threaded run)
w1 = t1; w0 = t1 - 150_000_000
busy = sum(min(int(r["End_Timestamp"]), w1) - max(int(r["Start_Timestamp"]), w0) for r in rows if int(r["End_Timestamp"]) > w0)
print("device busy in the last 150 ms of the trace: %.1f%%" % (100.0 * busy / (w1 - w0)))
gaps = collections.Counter()
prev = None
for r in rows:
    if int(r["Start_Timestamp"]) < w0: prev = r; continue
    if prev is not None:
        g = int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
        gaps[min(g // 10000, 50)] += max(g, 0)
    prev = r
print("idle time by gap length (10 us buckets): " + " ".join("%d0us:%.1fms" % (k, v / 1e6) for k, v in sorted(gaps.items()) if v > 500000))
PY
