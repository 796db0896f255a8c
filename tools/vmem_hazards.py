"""Hand-waited loads, checked in the BUILT code (cluster_kernels.hip: get_record_async / load_unit_async): between such a load and
the s_waitcnt that covers it the compiler believes the destination registers already hold the value -- it may copy, spill or reuse
them there, and the landing load then overwrites whatever lives in them.  This walks the disassembly of a kernel of the built
library along every control-flow path and reports each instruction that names a VGPR while a vector-memory load into it may still
be outstanding.
  python tools/vmem_hazards.py <kernel name pattern> [...]
Model (the compiler's own for gfx9, SIInsertWaitcnts): vector-memory operations -- loads, stores, atomics -- complete in order and
s_waitcnt vmcnt(N) leaves at most the N youngest of them outstanding.  The state at
an instruction -- per VGPR, the least number of operations issued behind the outstanding load into it, over all paths that reach the
instruction -- is propagated through the branches to a fixed point.
Used by tests/test_build_lint.py."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
READELF = kr.READELF
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
TARGET = re.compile(r"<[^>]*\+0x([0-9a-f]+)>\s*$|<[^>+]*>\s*$")
LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load")
OTHER_VMEM = ("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic", "buffer_wbl2", "buffer_inv")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(pattern):
    """{demangled name: [(address, mnemonic, operands, branch target or None)]} for the kernels of the library whose name matches"""
    found = {}
    for blob in kr.code_objects():
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            syms = subprocess.run([READELF, "-s", "--wide", f.name], capture_output=True, text=True, check=True).stdout
            names = sorted({ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln})
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
            for sym, d in zip(names, dem):
                if not re.search(pattern, d) or d in found:
                    continue
                txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, f.name], capture_output=True, text=True, check=True).stdout
                ins, base = [], None
                for ln in txt.splitlines():
                    m = LINE.match(ln)
                    if not m:
                        continue
                    addr = int(m.group(3), 16)
                    if base is None:
                        base = addr
                    target = None
                    if m.group(1).startswith(("s_cbranch", "s_branch")):
                        t = TARGET.search(ln)
                        assert t, ln
                        target = base + (int(t.group(1), 16) if t.group(1) else 0)
                    ins.append((addr, m.group(1), m.group(2), target))
                found[d] = ins
    return found


def hazards(ins):
    """[(address, instruction text, registers named while a load into them may be outstanding)]
    State at an instruction: {VGPR: the least number of loads issued behind the outstanding load into it, over all paths here}."""
    at = {a: i for i, (a, _, _, _) in enumerate(ins)}
    CAP = 64                                                    # (vmcnt counts to 63)
    state = [None] * len(ins)
    state[0] = {}
    out, work = {}, [0]

    def flow(j, st):
        cur = state[j]
        if cur is None:
            state[j] = dict(st)
            work.append(j)
            return
        changed = False
        for r, d in st.items():
            if cur.get(r, CAP + 1) > d:
                cur[r] = d
                changed = True
        if changed:
            work.append(j)

    def issue(st, dst):
        st = {r: min(d + 1, CAP) for r, d in st.items()}
        for r in dst:
            st[r] = 0
        return st

    while work:
        i = work.pop()
        addr, op, args, target = ins[i]
        st = state[i]
        if op.startswith(LOADS):
            dst = "" if re.search(r"\blds\b", args) else args.split(",")[0]   # (a load into LDS has no destination registers)
            named = vregs(args[len(dst):]) & st.keys()         # (its address registers; a load OVER an outstanding destination lands behind it: in order)
            if named:
                out[addr] = (op + " " + args, sorted(named))
            st = issue(st, vregs(dst))
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", args)
            if m:
                st = {r: d for r, d in st.items() if d < int(m.group(1))}
        elif op == "s_endpgm":
            continue
        else:
            named = vregs(args) & st.keys()
            if named:
                out[addr] = (op + " " + args, sorted(named))
            if op.startswith(OTHER_VMEM):
                returns = "atomic" in op and re.search(r"\b(glc|sc0)\b", args)
                st = issue(st, vregs(args.split(",")[0]) if returns else ())   # (an atomic with return is a load; the others only count)
        if op == "s_branch":
            flow(at[target], st)
            continue
        if op.startswith("s_cbranch"):
            flow(at[target], st)
        if i + 1 < len(ins):
            flow(i + 1, st)
    return [(a,) + out[a] for a in sorted(out)]


if __name__ == "__main__":
    bad = 0
    for pat in sys.argv[1:] or ["eval_cluster_five_kernel"]:
        for name, ins in sorted(disassemble(pat).items()):
            hz = hazards(ins)
            n_async = sum(1 for _, op, args, _ in ins if op.startswith("global_load_dwordx4") and (" nt" in args or " sc1" in args))
            print("%s: %d instructions, %d 16-byte loads with nt / sc1, %d hazards" % (name[:110], len(ins), n_async, len(hz)))
            for a, text, regs in hz:
                print("   %x: %s   <- outstanding: v%s" % (a, text, regs))
            bad += len(hz)
    sys.exit(1 if bad else 0)
