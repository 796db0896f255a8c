"""Instruction mix of one kernel from the compiler's assembly, per basic block (what the M sweep's issue slots go to).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only eval_kernels.hip -o /tmp/eval.s
  python tools/isa_mix.py /tmp/eval.s _ZN3pqa18eval_questions_f64ILi8ELi10ELb1ELb1ELb0EEEvNS_8EvalArgsE [min-block-size]
Classes: fp64 = v_*_f64 except rcp/rsq/cvt; trans = v_rcp/v_rsq/v_sqrt/v_log/v_exp; xlane = DPP modifiers, v_permlane*, v_readlane,
v_readfirstlane, v_writelane, ds_bpermute/swizzle; int = every other VALU (moves, shifts, logic, 32-bit integer, cndmask, cvt);
lds = ds_*; vmem = global_/buffer_/flat_/scratch_; smem = s_load*/s_buffer_load*; wait = s_waitcnt/s_nop; salu = the other s_*."""
import collections, re, sys

def classify(op, text):
    if op.startswith("v_"):
        if "dpp" in text or "row_" in text or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "xlane"
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp")):
            return "trans"
        if op.endswith("_f64") or "_f64_" in op:
            return "int" if op.startswith("v_cvt") else "fp64"
        return "int"
    if op.startswith(("ds_bpermute", "ds_swizzle", "ds_permute")):
        return "xlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"

def main():
    path, sym = sys.argv[1], sys.argv[2]
    least = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    blocks, cur, name = [], collections.Counter(), "entry"
    ops = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op, s)
        cur[c] += 1
        ops[name][op] += 1
    blocks.append((name, cur))
    order = ["fp64", "int", "xlane", "trans", "lds", "vmem", "smem", "wait", "barrier", "salu", "branch", "other"]
    print("%-14s %6s | " % ("block", "insts") + " ".join("%6s" % o for o in order))
    total = collections.Counter()
    for name, c in blocks:
        n = sum(c.values()); total.update(c)
        if n >= least:
            print("%-14s %6d | " % (name, n) + " ".join("%6d" % c[o] for o in order))
    print("%-14s %6d | " % ("whole kernel", sum(total.values())) + " ".join("%6d" % total[o] for o in order))
    for name, c in blocks:
        if sum(c.values()) >= max(least, 200):
            top = ops[name].most_common(24)
            print("\n%s: " % name + ", ".join("%s x%d" % t for t in top))

main()
