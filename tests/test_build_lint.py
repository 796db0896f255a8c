"""No-GPU lint of the compiled form of kernels that issue vector-memory loads the compiler does not account for
(probqa_amd/csrc/cluster_kernels.hip: get_record_async / load_unit_async, waited for by hand).  Between such a load and its wait
the compiler believes the destination registers hold the value; spilling or copying them in that window would save stale contents
and free registers a landing load then overwrites.  The kernels are therefore built with registers to spare, and this test holds
the build to it: no scratch, no AGPRs, at least 16 of the 256 VGPRs of their occupancy unused.  (hipcc cross-compiles without a GPU.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "probqa_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def makefile_flags():
    text = open(os.path.join(CSRC, "Makefile")).read()
    return re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1).split()


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="no hipcc")
def test_kernels_with_hand_waited_loads_have_registers_to_spare():
    cmd = [HIPCC, "--offload-arch=gfx950"] + makefile_flags() + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c",
                                                                  os.path.join(CSRC, "cluster_kernels.hip"), "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    usage, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    five = {n: u for n, u in usage.items() if "eval_cluster_five_kernel" in n}
    assert len(five) == 2, sorted(usage)      # Float and Double engines
    for n, u in five.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (n, u)
        assert u["AGPRs"] == 0, (n, u)
        assert u["VGPRs"] <= 240, (n, u)
