"""Copy what tools/prof.sh and the bench runs left under gpurun_out/ into profiles/ (tracked):
   gpurun_out/prof_r6_<CFG>/{summary.txt, trace/**/kernel_stats.csv, traffic_<CFG>.json} -> profiles/r06_<CFG>_*,
   gpurun_out/r6_bench_<name>.json -> profiles/r06_bench_<name>.json, gpurun_out/r6_<name>_stats.txt -> profiles/r06_<name>_stats.txt.
   traffic.json keeps the hand-written "note" fields of its entries."""
import glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
traffic_path = os.path.join(pr, "traffic.json")
traffic = json.load(open(traffic_path))
for cfg in ("S", "M", "L1", "L1_b32", "cluster"):
    d = os.path.join(go, "prof_r6_" + cfg)
    summ = os.path.join(d, "summary.txt")
    if not os.path.exists(summ) or os.path.getsize(summ) < 200:
        print("skip", cfg, "(no summary)")
        continue
    names = {"S": "S", "M": "M", "L1": "L1", "L1_b32": "L1_b32", "cluster": "long_rows_cluster"}
    shutil.copy(summ, os.path.join(pr, "r06_%s_rocprofv3_summary.txt" % names[cfg]))
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(pr, "r06_%s_kernel_stats.csv" % names[cfg]))
    tj = os.path.join(d, "traffic_%s.json" % ("eval_cluster" if cfg == "cluster" else cfg))
    if os.path.exists(tj):
        new = json.load(open(tj))["eval_cluster" if cfg == "cluster" else cfg]
        new["command"] = new["command"].replace(root + "/", "").replace("/root/repo/", "")
        import re
        new["command"] = re.sub(r"python \S*/bench.py", "python bench.py", new["command"])
        if "note" in traffic.get(cfg, {}):
            new["note"] = traffic[cfg]["note"]
        traffic[cfg] = new
    print("promoted", cfg)
json.dump(traffic, open(traffic_path, "w"), indent=1)
for f in glob.glob(os.path.join(go, "r6_bench_*.json")):
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(pr, "r06_" + os.path.basename(f)[3:]), "w").write(lines[-1] + "\n")
        print("promoted", os.path.basename(f))
for f in glob.glob(os.path.join(go, "r6_*_stats.txt")):
    shutil.copy(f, os.path.join(pr, "r06_" + os.path.basename(f)[3:]))
    print("promoted", os.path.basename(f))
