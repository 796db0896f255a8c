"""The N>1 path on CPU: world_size-2 `gloo` processes run probqa_amd.dist (shard bounds, all-gather of the 16-byte
(priority, index) records, global pick, prior broadcast after RecordAnswer).  The local sweep is supplied by the CPU
oracle here (tests may use it as a stand-in selector); on GPUs the same ShardedSelector wraps
PqaHip_EnqueueSelectArgmax and RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from probqa_amd import dist as pdist  # noqa: E402


def test_shard_bounds_follow_calc_split():
    assert pdist.shard_bounds(10, 4) == [3, 6, 8, 10]
    assert pdist.shard_bounds(1000, 8) == [125 * (i + 1) for i in range(8)]
    assert pdist.shard_range(10, 4, 2) == (6, 8)
    assert [pdist.owner_of(q, 10, 4) for q in (0, 2, 3, 7, 9)] == [0, 0, 1, 2, 3]


def test_pick_global_ties_nan_and_empty():
    def rec(rows):
        t = torch.zeros(len(rows), 2, dtype=torch.float64)
        for i, (p, q) in enumerate(rows):
            t[i, 0] = p
            t[i, 1:].view(torch.int64)[0] = q
        return t

    assert pdist.pick_global(rec([(1.0, 5), (3.0, 9), (2.0, 1)])) == (3.0, 9)
    assert pdist.pick_global(rec([(3.0, 9), (3.0, 4)])) == (3.0, 4)           # tie -> lowest index
    assert pdist.pick_global(rec([(float("nan"), 2), (1.0, 7)])) == (1.0, 7)  # NaN never wins
    assert pdist.pick_global(rec([(0.0, -1), (2.0, 3)])) == (2.0, 3)          # shard without eligible questions
    assert pdist.pick_global(rec([(0.0, -1), (0.0, -1)]))[1] == -1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import orclib

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = cases.Case("dist", 5, 61, 90, seed=77, qgaps=[4], answers=[])
    A, D, B = case.kb()
    q0, q1 = pdist.shard_range(case.Q, world, rank)
    # this rank's shard of the KB, plus a full oracle as the single-process truth
    shard = orclib.Oracle(case.K, q1 - q0, case.T, case.init)
    shard.set_kb(A[q0:q1], D[q0:q1], B)
    shard.set_question_gaps([q - q0 for q in case.qgaps if q0 <= q < q1])
    full = case.make_oracle()
    shard.start_quiz(16)
    full.start_quiz(16)

    def local_select(out):
        _, pri = shard.eval(8)
        i = shard.select_argmax(pri)
        out[0] = pri[i] if i >= 0 else 0.0
        out[1:].view(torch.int64)[0] = (i + q0) if i >= 0 else -1

    sel = pdist.ShardedSelector(local_select, torch.device("cpu"))
    picks = []
    for _ in range(4):
        pri_g, q = sel.select()
        _, fpri = full.eval(8)
        want = full.select_argmax(fpri)
        assert q == want and pri_g == fpri[want], (rank, q, want)
        picks.append(q)
        # RecordAnswer: the owner updates the prior, everybody receives it (8*ldT-byte broadcast)
        owner = pdist.owner_of(q, case.Q, world)
        ans = q % case.K
        prior = torch.from_numpy(shard.mants)  # shares memory with the shard's quiz
        if rank == owner:
            shard.record_answer(q - q0, ans, 15)
        pdist.broadcast_prior(prior, owner)
        full.record_answer(q, ans, 15)
        assert np.array_equal(shard.priors(), full.priors())
    ret[rank] = picks
    dist.destroy_process_group()


def test_two_rank_sharded_selection_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] == ret[1] and len(ret[0]) == 4 and len(set(ret[0])) == 4


def test_pick_when_all_host_half_of_the_shm_exchange():
    """PqaHip_PickWhenAll: the host side of ShmSelector (no GPU involved): waits for every slot's step number, then the
    exact pick.  Writers are threads that publish late and out of order."""
    import ctypes
    import struct
    import threading
    import time

    from probqa_amd import interop

    world, slot = 5, 64
    buf = (ctypes.c_char * (world * slot))()
    base = ctypes.addressof(buf)
    records = [(2.0, 40), (7.5, 12), (float("nan"), 3), (7.5, 9), (0.0, -1)]   # tie on 7.5 -> index 9; NaN and -1 lose

    def publish(r, delay):
        time.sleep(delay)
        struct.pack_into("<dq", buf, r * slot, *records[r])
        struct.pack_into("<Q", buf, r * slot + 16, 17)

    threads = [threading.Thread(target=publish, args=(r, 0.02 * (world - r))) for r in range(world)]
    for t in threads:
        t.start()
    pri, idx = interop.pick_when_all(base, world, slot, 17, 5.0)
    for t in threads:
        t.join()
    assert (pri, idx) == (7.5, 9)
    with pytest.raises(interop.PqaException, match="Timed out"):
        interop.pick_when_all(base, world, slot, 18, 0.05)          # nobody publishes step 18
    struct.pack_into("<dq", buf, 3 * slot, 0.0, -1)
    struct.pack_into("<dq", buf, 1 * slot, 0.0, -1)
    struct.pack_into("<dq", buf, 0 * slot, 0.0, -1)
    assert interop.pick_when_all(base, world, slot, 17, 1.0)[1] == 3   # only the NaN shard is left: it still has a question


def test_pick_batch_ties_nan_and_empty():
    allw = np.zeros((3, 4, 2))
    allw[:, :, 1] = -1
    allw[0, 0], allw[1, 0], allw[2, 0] = (1.0, 5), (3.0, 9), (2.0, 1)            # plain maximum
    allw[0, 1], allw[1, 1] = (3.0, 9), (3.0, 4)                                   # tie -> lowest index
    allw[0, 2], allw[2, 2] = (float("nan"), 2), (1.0, 7)                          # NaN never wins
    assert pdist.pick_batch(allw) == [9, 4, 7, -1]                                # last quiz: no shard has a question


def _batch_worker(rank, world, port, ret):
    """BASELINE configs[4]'s N > 1 path (bench.py run_batched): every rank sweeps ITS shard for all B quizzes, the per-quiz
    winners meet in one all-gather, every rank makes the same picks -- the single-process oracle's argmaxes."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import orclib

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = cases.Case("distb", 5, 47, 70, seed=78, qgaps=[11], answers=[])
    A, D, B = case.kb()
    q0, q1 = pdist.shard_range(case.Q, world, rank)
    n_quizzes = 6
    histories = [[((7 * i + 3 * j) % case.Q, (i + j) % case.K) for j in range(i % 3)] for i in range(n_quizzes)]
    histories = [[(q, a) for q, a in h if q != 11] for h in histories]
    mine, want = np.zeros((n_quizzes, 2)), []
    for b, hist in enumerate(histories):
        full = case.make_oracle()
        full.start_quiz(16)
        for q, a in hist:
            full.record_answer(q, a, 15)
        _, fpri = full.eval(8)
        want.append(full.select_argmax(fpri))
        shard = orclib.Oracle(case.K, q1 - q0, case.T, case.init)      # the rank's shard with the SAME posterior and asked bits
        shard.set_kb(A[q0:q1], D[q0:q1], B)
        shard.set_question_gaps([q - q0 for q in case.qgaps if q0 <= q < q1])
        shard.start_quiz(16)
        shard.mants[: case.T] = full.priors()
        asked = np.ctypeslib.as_array(shard.quiz.contents.asked, shape=((q1 - q0 + 7) // 8,))
        for q, _ in hist:
            if q0 <= q < q1:
                asked[(q - q0) >> 3] |= 1 << ((q - q0) & 7)
        _, pri = shard.eval(8)
        i = shard.select_argmax(pri)
        mine[b] = (pri[i], i + q0) if i >= 0 else (0.0, -1)
    picks = pdist.select_batch(torch.from_numpy(mine))
    assert picks == want, (rank, picks, want)
    ret[rank] = picks
    dist.destroy_process_group()


def test_two_rank_batched_selection_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_batch_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] == ret[1] and len(ret[0]) == 6
