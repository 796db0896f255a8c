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


# ---- two ranks on two devices (skipped on a one-GPU box: RCCL refuses two ranks on one device) ----------------------------------
def _two_rank_worker(rank, world, port, ret):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import orclib  # noqa: F401  (the full cube's oracle: the single-process truth)
    from probqa_amd import dist as pdist
    from probqa_amd import interop

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    lib = rccl()
    uid = UniqueId()
    if rank == 0:
        assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    box = [ctypes.string_at(ctypes.addressof(uid), 128) if rank == 0 else None]   # (the raw 128 bytes: .value would stop at the first NUL)
    dist.broadcast_object_list(box, src=0)
    ctypes.memmove(ctypes.addressof(uid), box[0], 128)
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    try:
        case = cases.Case("rccl2", 5, 61, 900, seed=77, qgaps=[4], answers=[])
        A, D, B = case.kb()
        q0, q1 = pdist.shard_range(case.Q, world, rank)
        eng = interop.PqaEngineFactory().create_hip_engine(interop.EngineDefinition(case.K, q1 - q0, case.T, init_amount=case.init), q0, case.Q, rank)
        eng.set_option("workers", cases.WORKERS)
        eng.set_kb(A[q0:q1], D[q0:q1], B)
        gaps_here = [q for q in case.qgaps if q0 <= q < q1]
        if gaps_here:
            eng.set_question_gaps(gaps_here)
        full = case.make_oracle()
        quiz = eng.start_quiz()
        full.start_quiz(cases.WORKERS)
        device = torch.device("cuda", rank)
        sel = pdist.ShardedSelector(lambda out: eng.enqueue_select_argmax(quiz, out.data_ptr()), device)   # torch.distributed's RCCL all-gather
        picks = []
        for step in range(4):
            _, fpri = full.eval(8 * cases.WORKERS)
            want = full.select_argmax(fpri)
            pri_c, q_c = eng.select_argmax_rccl(quiz, comm.value, world)          # the C entry: ONE ncclAllGather on the engine's stream
            pri_t, q_t = sel.select()
            assert q_c == want == q_t, (rank, step, q_c, q_t, want)
            assert abs(pri_c - fpri[want]) <= 1e-9 * abs(fpri[want]) and pri_c == pri_t
            picks.append(q_c)
            # RecordAnswer: the owner updates its posterior, everybody receives it (8 ldT bytes, one broadcast)
            owner, ans = pdist.owner_of(want, case.Q, world), want % case.K
            eng.set_active_question(quiz, want)
            if rank == owner:
                eng.record_answer(quiz, ans)
            else:
                eng.record_answer_remote(quiz, ans)
            ptr, ld = eng.prior_device_ptr(quiz)
            eng.synchronize()                                                      # (the owner's posterior kernel has finished)
            pdist.broadcast_prior(pdist.tensor_from_device_ptr(ptr, ld, device), owner)
            torch.cuda.synchronize()
            full.record_answer(want, ans, cases.WORKERS - 1)
            assert np.array_equal(eng.get_priors(quiz), full.priors()), (rank, step)
        ret[rank] = picks
        eng.close()
    finally:
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclCommDestroy(comm)
        dist.destroy_process_group()


def test_two_ranks_on_two_devices():
    """PqaHip_SelectArgmaxRccl and probqa_amd.dist's selector over a communicator of TWO ranks, one per device: every rank returns the
    whole cube's pick (the oracle's) on every step, and the posterior broadcast after RecordAnswer keeps the shards bit-identical."""
    import socket

    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("one device: RCCL refuses two ranks on it (the one-rank communicator above runs the same code)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_rank_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] == ret[1] and len(ret[0]) == 4


def test_the_two_rank_worker_with_a_world_of_one():
    """The same worker as one process (what a one-GPU box can run of it): torch.distributed's nccl backend, a communicator made with
    RCCL's C API from a broadcast id, the sharded engine constructor, both selectors, the posterior hand-over."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_rank_worker, args=(1, port, ret), nprocs=1, join=True)
    assert len(ret[0]) == 4 and len(set(ret[0])) == 4
