#!/bin/bash
# usage: tools/prof.sh <tag> <config S|M|L1|...> <variant> <steps>
# One rocprofv3 --kernel-trace --stats pass, then separate --pmc passes (never combined with sys/runtime tracing), of
#   python bench.py --config <config> ...      (single-quiz configs: --no-server, so that the kernel in the statistics is the
#                                               launched sweep; batched configs: the batched sweep)
# Outputs under gpurun_out/prof_<tag>/: the raw csv files, summary.txt (kernel stats + per-launch counter averages) and
# traffic_<config>.json = {bytes_per_launch (FETCH_SIZE, KB -> bytes, x2 as MI355X_MICROARCH.md prescribes for 16 B/lane
# streaming reads on gfx950), valu: {...}} -- copy both into profiles/ (traffic_<config>.json merges into profiles/traffic.json).
#        tools/prof.sh <tag> custom:<kernel-name-substring> <command ...>     (any command: the same passes, counters of that kernel)
TAG=$1; CFG=$2; VAR=${3:-0}; STEPS=${4:-50}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
case $CFG in
  custom:*) KERNEL=${CFG#custom:}; shift 2; CMD="$*"; CFG=$KERNEL ;;
  L1_b*) CMD="python $PWD/bench.py --config L1 --batch ${CFG#L1_b} --steps $STEPS --warmup 1 --no-cpu-baseline"; KERNEL=eval_batch_kernel ;;
  L1|LS|SB) CMD="python $PWD/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline"; KERNEL=eval_batch_kernel ;;
  *) CMD="python $PWD/bench.py --config $CFG --variant $VAR --steps $STEPS --warmup 5 --no-cpu-baseline --batch 0 --no-server --no-quiz-loop --no-points"; KERNEL=eval_questions ;;
esac
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o p -- $CMD > /dev/null 2> $OUT/pmc_$name.err
done
cd - >/dev/null
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections, json, os
out="$OUT"; cfg="$CFG"; kern="$KERNEL"
print("command:", "$CMD")
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print("== rocprofv3 --kernel-trace --stats:", os.path.basename(f))
    for i,row in enumerate(csv.reader(open(f))):
        if i<8: print(",".join(row))
vals={}
for d in sorted(glob.glob(out+"/pmc_*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for row in csv.DictReader(open(f)):
            if kern not in row.get("Kernel_Name",""): continue
            k=row["Counter_Name"]; acc[k][0]+=1; acc[k][1]+=float(row["Counter_Value"])
        for k,(n,s) in acc.items():
            print("pmc %-24s per-launch avg of %s: %.6g (n=%d)"%(k,kern,s/n,n)); vals[k]=s/n
import hashlib
src=hashlib.sha256(b"".join(open(os.path.join("$PWD","probqa_amd","csrc",f),"rb").read() for f in sorted(os.listdir(os.path.join("$PWD","probqa_amd","csrc"))) if f.endswith(".hip") or f in ("pqa_device.h","eval_device.h","prior_device.h","pole_device.h","pqa_kernels.h"))).hexdigest()[:16]
rec={"command": "$CMD", "kernel": kern, "kernel_sources_sha16": src}
if "FETCH_SIZE" in vals:
    rec.update({"bytes_per_launch": vals["FETCH_SIZE"]*1024*2, "fetch_size_kb_raw": vals["FETCH_SIZE"],
                "correction": "x1024 (KB) x2 (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section)"})
    print("traffic bytes/launch (corrected):", rec["bytes_per_launch"])
if "SQ_INSTS_VALU" in vals:
    v={k: vals[k] for k in ("SQ_WAVES","SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY") if k in vals}
    if v.get("SQ_WAVES"):
        v["valu_insts_per_wave"]=v["SQ_INSTS_VALU"]/v["SQ_WAVES"]
    if v.get("SQ_WAVE_CYCLES"):
        # both count quad-cycles summed over waves: the share of its resident time a wave has a VALU instruction executing
        v["valu_active_share_of_wave_cycles"]=v["SQ_ACTIVE_INST_VALU"]/v["SQ_WAVE_CYCLES"]
    rec["valu"]=v
    print("valu:", json.dumps(v))
json.dump({cfg: rec}, open(out+"/traffic_"+cfg+".json","w"), indent=1)
PY
