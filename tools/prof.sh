#!/bin/bash
# usage: tools/prof.sh <tag> <config S|M> <variant> <steps>
# kernel-trace/stats pass + separate PMC passes (never combined with sys/runtime tracing), outputs under gpurun_out/prof_<tag>/
TAG=$1; CFG=$2; VAR=$3; STEPS=${4:-50}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --config $CFG --variant $VAR --steps $STEPS --warmup 5 --no-cpu-baseline --batch 0 --no-server"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o p -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd - >/dev/null
python - <<PY
import csv, glob, collections, json, os
out="$OUT"; cfg="$CFG"
# kernel stats
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", os.path.basename(f))
    for i,row in enumerate(csv.reader(open(f))):
        if i<8: print(",".join(row))
# pmc: average per dispatch of eval kernel
vals={}
for d in sorted(glob.glob(out+"/pmc_*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        rd=csv.DictReader(open(f))
        for row in rd:
            if "eval_questions" not in row.get("Kernel_Name",""): continue
            k=row["Counter_Name"]; acc[k][0]+=1; acc[k][1]+=float(row["Counter_Value"])
        for k,(n,s) in acc.items():
            print("pmc %-24s per-dispatch avg %.6g (n=%d)"%(k,s/n,n)); vals[k]=s/n
if "FETCH_SIZE" in vals:
    # FETCH_SIZE is in KB; on gfx950 it reports half of the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM)
    rec={"bytes_per_launch": vals["FETCH_SIZE"]*1024*2, "fetch_size_kb_raw": vals["FETCH_SIZE"],
         "correction": "x1024 (KB) x2 (gfx950 wide-read undercount)", "command": "$CMD"}
    json.dump({cfg: rec}, open(out+"/traffic_"+cfg+".json","w"), indent=1)
    print("traffic bytes/launch (corrected):", rec["bytes_per_launch"])
PY
