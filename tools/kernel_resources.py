"""Register and scratch use of every kernel in the BUILT library (probqa_amd/libPqaCore.so): the gfx950 code objects are taken out
of its offload bundles and their metadata notes read with llvm-readelf -- no recompilation (tools/kres.sh recompiles one file).
  python tools/kernel_resources.py [pattern]        name, VGPRs, AGPRs, scratch bytes per lane, spilled VGPRs / SGPRs
Used by tests/test_build_lint.py: no kernel the engine dispatches by default may touch scratch memory."""
import os
import re
import struct
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "probqa_amd", "libPqaCore.so")
READELF = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB):
    """the gfx950 ELF images in the library's offload bundles (header: magic, count, then {offset, size, triple} per image)"""
    data = open(path, "rb").read()
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size > 0:
                yield data[p + off:p + off + size]


def kernels(path=LIB):
    """{demangled kernel name: {vgpr, agpr, scratch, vgpr_spill, sgpr_spill}} over every gfx950 code object of the library"""
    recs = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", notes, re.S | re.M):
            meta = yaml.safe_load(doc)
            for k in (meta or {}).get("amdhsa.kernels", []):
                recs[k[".name"]] = {"vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                                    "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                                    "dynamic_stack": bool(k.get(".uses_dynamic_stack", False))}
    names = list(recs)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return {d: recs[n] for n, d in zip(names, dem)}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else "."
    for name, r in sorted(kernels().items()):
        if re.search(pat, name):
            print("%-120s v=%3d a=%3d scratch=%4d vS=%3d sS=%3d" % (name[:120], r["vgpr"], r["agpr"], r["scratch"], r["vgpr_spill"], r["sgpr_spill"]))
