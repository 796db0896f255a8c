"""PqaHip_SelectArgmaxRccl: the shards' 16-byte winners through ONE RCCL all-gather on the engine's stream, for a process-per-GPU
host that owns the communicator and is not Python (north_star's collective inside the C ABI; probqa_amd/dist.py does the same
through torch.distributed).  A box has one GPU, so the communicator here has one rank -- made with RCCL's own C API through ctypes,
as a native host would make it; the collective, its buffers and the pick are the code every rank of N runs."""
import ctypes

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def rccl():
    for name in ("librccl.so.1", "librccl.so"):
        try:
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("no librccl.so on this box")


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]   # NCCL_UNIQUE_ID_BYTES


@pytest.mark.parametrize("i", [0, 1, 4])
def test_one_rank_communicator_gives_the_engines_own_pick(i, factory):
    import torch
    lib = rccl()
    torch.cuda.set_device(0)
    uid = UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        case = cases.small_cases()[i]
        eng, orc = case.make_engine(factory), case.make_oracle()
        quiz = eng.start_quiz()
        orc.start_quiz(cases.WORKERS)
        for step in range(len(case.answers) + 1):
            pri, idx = eng.select_argmax_rccl(quiz, comm.value, 1)
            _, opri = orc.eval(128)
            want = orc.select_argmax(opri)
            assert idx == want, (case.name, step, idx, want)
            if want >= 0:
                assert abs(pri - opri[want]) <= 1e-9 * abs(opri[want]), (case.name, step, pri, opri[want])
                assert idx == eng.next_question_argmax(quiz)
            if step < len(case.answers):
                q, a = case.answers[step]
                eng.set_active_question(quiz, q)
                eng.record_answer(quiz, a)
                orc.record_answer(q, a, cases.WORKERS - 1)
        eng.close()
    finally:
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclCommDestroy(comm)
