import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line); r=d["roofline"]
    print("%-24s kernel_us=%9.2f sync_sel/s=%9.1f (%.2f us/step) pipelined=%.0f" % (d["config"]["eval_kernel"], r["kernel_us"], d["value"], 1e6/d["value"], d["pipelined_selections_per_sec"]))
