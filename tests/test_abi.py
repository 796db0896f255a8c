"""No-GPU checks of the drop-in boundary: libPqaCore.so builds for gfx950, loads, exports every symbol that
include/PqaCInterop.h and include/PqaHipExt.h declare, the POD layouts match the reference's, and without a GPU the
factory fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re
import subprocess

import pytest

from probqa_amd import interop

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return re.findall(r"PQACORE_API\s+[\w\s\*]+?\b(\w+)\s*\(", text)


def test_headers_and_binding_agree():
    ref = declared_functions("PqaCInterop.h")
    ext = declared_functions("PqaHipExt.h")
    assert len(ref) == 40, ref  # the reference's PqaCInterop.h:48-108 declares 40 functions
    assert sorted(ref) == sorted(interop.REFERENCE_EXPORTS)
    assert sorted(ext) == sorted(interop.HIP_EXPORTS)


def test_library_loads_and_exports_every_declared_symbol(factory):
    lib = interop.load_library()
    for name in declared_functions("PqaCInterop.h") + declared_functions("PqaHipExt.h"):
        assert getattr(lib, name) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", interop.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(interop.REFERENCE_EXPORTS) <= exported and set(interop.HIP_EXPORTS) <= exported


def test_pod_layouts_match_reference():
    # reference PqaCInterop.h:9-42 with #pragma pack(8)
    d = interop.CiEngineDefinition
    assert ctypes.sizeof(d) == 48
    assert [getattr(d, f).offset for f in ("nAnswers", "nQuestions", "nTargets", "precType", "precExponent",
                                             "precMantissa", "initAmount", "memPoolMaxBytes")] == [0, 8, 16, 24, 26, 28, 32, 40]
    assert ctypes.sizeof(interop.CiAnsweredQuestion) == 16 and ctypes.sizeof(interop.CiEngineDimensions) == 24
    assert ctypes.sizeof(interop.CiRatedTarget) == 16 and interop.CiRatedTarget.prob.offset == 8
    assert ctypes.sizeof(interop.CiAddQorTParam) == 16
    assert ctypes.sizeof(interop.CiHipSelection) == 16 and ctypes.sizeof(interop.CiHipShard) == 24


def test_null_handles_follow_the_reference_convention(factory):
    lib = interop.load_library()
    # GET_ENGINE_OR_RET_ERR / _ASSIGN_ERR / _LOG_ERR of reference PqaCInterop.cpp:65-86
    err = lib.PqaEngine_RecordAnswer(None, 0, 0)
    assert err
    e = interop.PqaError(err)
    assert "Expected non-null argument" in e.to_string(True) and "IPqaEngine" in e.to_string(False)
    c_err = ctypes.c_void_p()
    assert lib.PqaEngine_NextQuestion(None, ctypes.byref(c_err), 0) == -1 and c_err.value
    interop.PqaError(c_err.value)
    dims = interop.CiEngineDimensions()
    assert lib.PqaEngine_CopyDims(None, ctypes.byref(dims)) == 0
    c_err = ctypes.c_void_p()
    assert lib.PqaEngineFactory_CreateCpuEngine(None, ctypes.byref(c_err), None) is None and c_err.value
    assert "IPqaEngineFactory" in interop.PqaError(c_err.value).to_string(True)
    lib.CiReleasePqaError(None)
    lib.CiReleasePqaEngine(None)
    import tempfile
    base = os.path.join(tempfile.mkdtemp(prefix="pqa_log_"), "x")   # (the log file <base>_<UTC time>_<pid>.log: not into the repository)
    assert lib.Logger_Init(ctypes.byref(ctypes.c_void_p()), base.encode()) == 1
    os.environ["PQA_TEST_LOG_BASE"] = base       # (tests that expect log entries later in this process look here)


def test_no_cpu_fallback_without_a_gpu(factory):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(5, 10, 10))
    assert eng is None and err is not None and "no CPU fallback" in err.to_string(True)


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under probqa_amd/ may reference it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "probqa_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pqa_oracle" not in text and "orclib" not in text, os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", interop.LIB_PATH], text=True)
    assert "oracle" not in out
