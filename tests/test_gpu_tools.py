"""The measurement scripts under tools/ are not needed to build, test or bench -- but the numbers in DESIGN.md and profiles/ come from
them, so the suite runs each of the ones that take a size on a small cube: they must still run against the library as it is and
agree with themselves (same picks by every form they time)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args, env=None, timeout=120):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)] + [str(a) for a in args], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    return r.stdout


def test_sweep_timing_tool():
    out = run_tool("sweep_timing.py", "300x5x700")
    assert re.search(r"300x5x700 wg256_np\d+: sweep [\d.]+ us after StartQuiz; [\d.]+ us after", out), out


def test_pole_cost_and_late_tools():
    out = run_tool("pole_cost.py", "300x5x700", "late", 5)
    assert re.search(r"300x5x700 \S+ late .*: [\d.]+ us per sweep back to back", out), out
    out = run_tool("late_run.py", 0, 4)
    assert len(re.findall(r"worst relative deviation [\d.e+-]+", out)) == 2, out
    out = run_tool("late_probe.py", 6)
    assert "late0006" in out, out


def test_batch_bench_tool_same_pick_by_every_group_count():
    picks = set()
    for groups in (1, 4, 0):
        out = run_tool("batch_bench.py", 600, 5, 900, "f32", 20, 0, 1, 2, 0, groups)
        picks.add(re.search(r"pick0=(\d+)", out).group(1))
    assert len(picks) == 1, picks


def test_single_quiz_bench_tool_same_pick_by_both_cluster_forms():
    picks = set()
    for form in (1, 2):
        out = run_tool("f32_single_bench.py", 40, 5, 20000, 3, 0, form)
        lines = [l for l in out.splitlines() if "single quiz" in l]
        assert len(lines) == 2 and ("_ahead" in lines[0]) == (form == 2), out
        picks.add(tuple(re.search(r"pick=(\d+)", l).group(1) for l in lines))
    assert len(picks) == 1, picks


def test_sharded_bench_tool():
    out = run_tool("sharded_bench.py", 300, 5, 500, 50, env={"PQA_DEVICES": "0,0,0"})
    assert out.count("shards=3") == 2 and "argmax" in out and "sampled" in out, out


def test_isa_mix_tool_reads_a_compiler_listing(tmp_path):
    """(no GPU needed, but hipcc is: the listing of the smallest kernel file)"""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    listing = tmp_path / "select.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "probqa_amd", "csrc", "select_kernels.hip"), "-o", str(listing)], check=True, timeout=300,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sym = next(m.group(1) for m in (re.match(r"^(_Z\w+):", l) for l in open(listing)) if m and "kernel" in m.group(1))
    out = run_tool("isa_mix.py", listing, sym, 10)
    assert "whole kernel" in out
