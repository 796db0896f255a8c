"""The round's table of profiles/README.md, printed from the committed files themselves (profiles/r06_*): bench lines, rocprofv3
kernel statistics, counter summaries.  Nothing is typed in by hand and nothing is chosen: the files are one run of
tools/refresh_profiles.sh on one box.
  python tools/profiles_table.py [round prefix, default r06]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PR = os.path.join(ROOT, "profiles")
R = sys.argv[1] if len(sys.argv) > 1 else "r06"


def bench(name):
    return json.loads(open(os.path.join(PR, "%s_bench_%s.json" % (R, name))).read().strip().splitlines()[-1])


def kernel_row(path, pattern):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith('"') or re.match(r"^(void )?pqa::", ln)]
    for row in csv.reader(lines):
        if re.search(pattern, row[0]):
            return {"name": row[0], "calls": int(row[1]), "avg_us": float(row[3]) / 1e3, "min_us": float(row[5]) / 1e3, "max_us": float(row[6]) / 1e3}
    return None


def pmc(cfg, counter):
    for ln in open(os.path.join(PR, "%s_%s_rocprofv3_summary.txt" % (R, cfg))):
        m = re.match(r"pmc %s\s+per-launch avg of \S+ ([0-9.e+]+)" % counter, ln)
        if m:
            return float(m.group(1))
    return None


def short(name):
    return re.sub(r"\(pqa::.*$", "", name.replace("void ", "").replace("pqa::", "").replace("(anonymous namespace)::", ""))


rows = []
S, Sd, M, L1, L1b = bench("S"), bench("S_driver_command_steps20_warmup5"), bench("M"), bench("L1"), bench("L1_b32")
ks = kernel_row(os.path.join(PR, R + "_S_kernel_stats.csv"), "eval_questions_f64")
r = S["roofline"]
rows.append(("S 1000×5×1000 fp64, 1 quiz", "`%s`" % short(ks["name"]),
             "**%.2f µs** back to back (HIP events on the engine's stream, the figure `frac` is on); %.1f µs inside a synchronous call (p50, the engine's own events); rocprofv3 average %.2f µs over %d launches (min %.2f)"
             % (r["kernel_us"], r["synchronous_launch_us"]["p50"], ks["avg_us"], ks["calls"], ks["min_us"]),
             "HBM **%.3f**; FETCH %.1f MB (48.0 algorithmic)" % (r["frac"], r["traffic"] / 1e6),
             "`value` **%.1f k selections/s** (default path: one launch per selection), %.3f ms per step; the driver's command (20 steps, 5 warm-up) %.1f k; resident sweep (opt-in) %.1f k, step %.2f µs; CPU port %.0f /s on %d threads"
             % (S["value"] / 1e3, S["ms_per_step"], Sd["value"] / 1e3, S["resident_sweep"]["selections_per_sec"] / 1e3, S["resident_sweep"]["resident_step_us"]["mean"],
                S["cpu_baseline"]["value"], S["cpu_baseline"]["cores"])))
km = kernel_row(os.path.join(PR, R + "_M_kernel_stats.csv"), "eval_questions_f64")
r = M["roofline"]
late = (S.get("hbm_point_M") or {}).get("late_state") or {}
rows.append(("M 10000×5×10000 fp64, 1 quiz", "`%s`" % short(km["name"]),
             "**%.1f µs** HIP events; rocprofv3 average %.1f µs over %d launches (min %.1f, max %.1f)" % (r["kernel_us"], km["avg_us"], km["calls"], km["min_us"], km["max_us"]),
             "HBM **%.3f** (%.2f TB/s); FETCH %.4f GB (4.8 algorithmic); `SQ_INSTS_VALU` %.3g, `SQ_WAIT_ANY` %.0f %% of the wave-cycles"
             % (r["frac"], r["achieved"] / 1e3, r["traffic"] / 1e9, pmc("M", "SQ_INSTS_VALU"), 100 * pmc("M", "SQ_WAIT_ANY") / pmc("M", "SQ_WAVE_CYCLES")),
             "%.0f selections/s; late state (12 answers): `NextQuestionArgmax` %.0f µs gated / %.0f µs with every listed question redone; CPU port %.1f /s"
             % (M["value"], late.get("synchronous_argmax_selection_us", 0), late.get("synchronous_argmax_selection_us_with_every_listed_question_redone", 0), M["cpu_baseline"]["value"])))
for label, b, cfg in (("L1 12500×5×100000 fp32, 256 quizzes", L1, "L1"), ("L1 at 32 quizzes", L1b, "L1_b32")):
    kb = kernel_row(os.path.join(PR, "%s_%s_kernel_stats.csv" % (R, cfg)), "eval_batch_kernel")
    r = b["roofline"]
    rows.append((label, "`%s`" % short(kb["name"]), "%.1f ms per step (HIP events); rocprofv3 average %.1f ms" % (r["kernel_us"] / 1e3, kb["avg_us"] / 1e3),
                 "fp32 VALU **%.3f**; FETCH %.0f GB" % (r["frac"], r["traffic"] / 1e9), "%.0f selections/s" % b["value"]))
p = os.path.join(PR, R + "_long_rows_100000_stats.txt")
k64, k32 = kernel_row(p, "eval_cluster.*<double"), kernel_row(p, "eval_cluster.*<float")
per = dict(re.findall(r"(f32|f64) single quiz: ([0-9.]+) us", open(p).read()))
rows.append(("2000×5×100000, 1 quiz", "`%s` / `<double…>`" % short(k32["name"]),
             "rocprofv3 averages **%.3f / %.3f ms** (%d launches each); per selection %.0f / %.0f µs" % (k32["avg_us"] / 1e3, k64["avg_us"] / 1e3, k64["calls"], float(per["f32"]), float(per["f64"])),
             "%.2f / %.2f TB/s of cube = **%.2f / %.2f** of the HBM peak" % (4.8e9 / k32["avg_us"] / 1e6, 9.6e9 / k64["avg_us"] / 1e6, 4.8e9 / k32["avg_us"] / 1e6 / 8, 9.6e9 / k64["avg_us"] / 1e6 / 8),
             "FETCH %.2f GB per launch (both averaged; 7.2 algorithmic)" % (json.load(open(os.path.join(PR, "traffic.json")))["cluster"].get("bytes_per_launch", 0) / 1e9)))
p = os.path.join(PR, R + "_f32_single_M_stats.txt")
kf = kernel_row(p, "eval_questions_f32")
per = dict(re.findall(r"(f32|f64) single quiz: ([0-9.]+) us", open(p).read()))
rows.append(("10000×5×10000, 1 quiz, Float engine", "`%s`" % short(kf["name"]), "rocprofv3 average **%.0f µs** (%d launches); %.0f µs per selection" % (kf["avg_us"], kf["calls"], float(per["f32"])),
             "%.2f TB/s of cube = **%.2f**" % (2.4e9 / kf["avg_us"] / 1e6, 2.4e9 / kf["avg_us"] / 1e6 / 8), "(Double engine beside it: %.0f µs per selection)" % float(per["f64"])))
print("| config | kernel | kernel time | roofline | line |\n|---|---|---|---|---|")
for row in rows:
    print("| " + " | ".join(row) + " |")
q = S.get("quiz_loop") or {}
t = S.get("quiz_loop_threads") or {}
print("\nLearner loop at S (`%s_bench_S.json`): Python wrapper, one quiz at a time %.1f k questions/s; native client threads on ONE engine — %s."
      % (R, q.get("questions_per_sec", 0) / 1e3, ", ".join("%s: %.1f k" % (n, v["questions_per_sec"] / 1e3) for n, v in t.items() if isinstance(v, dict) and "questions_per_sec" in v)))
