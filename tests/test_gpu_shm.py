"""Question-axis shards in separate processes, winners exchanged through host shared memory (probqa_amd/dist.py:
ShmSelector).  Two processes share the one GPU of the test box, each holding half of the questions; every step both must
return what a single engine holding the whole cube selects."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, Q, T, SEED, STEPS = 5, 90, 700, 77, 6


def _rank_main(rank, world, name, out):
    from probqa_amd import dist as pdist
    from probqa_amd import interop

    try:
        q_first, q_limit = pdist.shard_range(Q, world, rank)
        eng = interop.PqaEngineFactory().create_hip_engine(
            interop.EngineDefinition(K, q_limit - q_first, T, init_amount=0.1), q_first, Q, 0)
        eng.fill_synthetic(8.0, 0.5, SEED)
        quiz = eng.start_quiz()
        sel = pdist.ShmSelector(eng, quiz, rank, world, name, create=False)
        picks, gaps = [], []
        for _ in range(STEPS):
            pri, q = sel.select()
            picks.append((pri, q))
            gaps.append(q)                      # take the winner out on every shard and select again
            eng.set_question_gaps(gaps)
        sel.close()
        eng.close()
        out.put((rank, picks))
    except Exception as e:  # noqa: BLE001 - reported to the parent
        out.put((rank, repr(e)))


def test_two_processes_shared_memory_exchange(factory):
    from probqa_amd import interop

    whole = factory.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, 0)
    whole.fill_synthetic(8.0, 0.5, SEED)
    quiz = whole.start_quiz()
    want, gaps = [], []
    for _ in range(STEPS):
        pri = whole.eval_priorities(quiz)
        q = int(np.argmax(pri))
        want.append((float(pri[q]), q))
        gaps.append(q)
        whole.set_question_gaps(gaps)
    whole.close()

    world, name = 2, "test_%d" % os.getpid()
    path = "/dev/shm/pqa_select_%s" % name
    with open(path, "wb") as f:                 # the segment exists (zeroed) before either rank opens it
        f.write(b"\0" * (2 * world * 64))
    try:
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        procs = [ctx.Process(target=_rank_main, args=(r, world, name, out)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict(out.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
    finally:
        os.unlink(path)
    for r in range(world):
        assert isinstance(got[r], list), got[r]
        assert [q for _, q in got[r]] == [q for _, q in want], (r, got[r], want)
        assert np.allclose([p for p, _ in got[r]], [p for p, _ in want], rtol=1e-11, atol=0)
    assert len({q for _, q in want}) == STEPS   # six different winners, some from each shard
    assert {q < Q // 2 for _, q in want} == {True, False}
