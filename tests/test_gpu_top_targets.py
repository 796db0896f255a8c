"""ListTopTargets on the device against the oracle's restatement of the reference's algorithm
(oracle/pqa_oracle.c: orc_list_top_targets -- CEListTopTargetsAlgorithm::RunHeapifyBased, PqaCore/CEListTopTargetsAlgorithm.cpp:30-95,
over the pieces of CEHeapifyPriorsSubtaskMake.cpp:42-88): gaps and probabilities <= 0 are dropped, the rest comes by descending
probability.

Where all listed probabilities differ -- and the next one below the list differs from the last listed -- the listing is the reference's,
record for record (targets and the bits of the probabilities).  EQUAL probabilities the reference lists in the order its per-thread
heaps happen to hold them, which changes with the thread count of the machine (tests/test_oracle.py::
test_list_top_targets_tie_order_is_the_heaps); the engine lists them by ascending target.  `same_listing` below holds both
sides to everything that does not depend on that order: the sequence of probabilities bit for bit, every listed target holding the
probability it is listed with, no target twice, and per probability value the engine's targets = the lowest-numbered holders."""
import os
import time

import numpy as np
import pytest

import cases
import orclib
from probqa_amd import interop, synth

pytestmark = pytest.mark.gpu

W = cases.WORKERS


def listed(eng, quiz, n):
    return [(r.i_target, r.prob) for r in eng.list_top_targets(quiz, n)]


def same_listing(got, want, posterior, gaps=()):
    """got: the engine's [(target, prob)], want: the oracle's; identical wherever the probabilities are distinct."""
    assert [p for _, p in got] == [p for _, p in want], "probability sequences differ"
    gt, wt = (np.array([t for t, _ in side], dtype=np.int64) for side in (got, want))
    gp = np.array([p for _, p in got], dtype=np.float64)
    assert len(set(gt.tolist())) == len(gt) and len(set(wt.tolist())) == len(wt)
    live = posterior > 0
    live[list(gaps)] = False
    assert live[gt].all() and live[wt].all()
    assert np.array_equal(posterior[gt], gp) and np.array_equal(posterior[wt], gp)
    for v in np.unique(gp):
        holders = np.flatnonzero(live & (posterior == v))
        mine = gt[gp == v]
        assert mine.tolist() == holders[: len(mine)].tolist(), "equal probabilities: the lowest-numbered holders, ascending"
        if len(holders) == 1:
            assert wt[gp == v].tolist() == mine.tolist()


def make(factory, K, Q, T, seed=5, init=0.1, f32=False, noise=0.5):
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=init, **kw))
    assert err is None, err
    eng.set_option("workers", W)
    A, D, B = synth.synthetic_kb(K, Q, T, init, 8.0, noise, seed)
    eng.set_kb(A, D, B)
    if f32:
        A, D, B = (x.astype(np.float32).astype(np.float64) for x in (A, D, B))
    orc = orclib.Oracle(K, Q, T, init)
    orc.set_kb(A, D, B)
    return eng, orc


def answer_both(eng, orc, quiz, q, a):
    eng.set_active_question(quiz, q)
    eng.record_answer(quiz, a)
    orc.record_answer(q, a, W - 1)


@pytest.mark.parametrize("dims", [(4, 30, 500), (3, 12, 1000), (5, 9, 1025), (2, 6, 4097), (3, 5, 16384)], ids=str)
def test_listing_equals_the_reference_algorithm(factory, dims):
    """A noisy cube: all posteriors distinct, so the listing is the oracle's record for record -- through the cache RecordAnswer's
    kernel leaves, the quiz's own lines, the engine's lines (up to 256) and the host path (longer lists)."""
    K, Q, T = dims
    eng, orc = make(factory, K, Q, T, seed=T)
    gaps = [3, T // 2, T - 1]
    eng.set_target_gaps(gaps)
    orc.set_target_gaps(gaps)
    quiz = eng.start_quiz()
    orc.start_quiz(W)
    rng = np.random.default_rng(T)
    for step in range(5):
        answer_both(eng, orc, quiz, int(rng.integers(Q)) if step else 0, int(rng.integers(K)))
        post = eng.get_priors(quiz)
        assert np.array_equal(post, orc.priors())
        live = np.delete(post, gaps)
        distinct = len(np.unique(live)) == len(live)
        for n in (1, 5, 16, 40, 256, 300, T + 5):
            got, want = listed(eng, quiz, n), orc.list_top_targets(n, W)
            if distinct:
                assert got == want, (step, n)
            same_listing(got, want, post, gaps)
        assert distinct or step > 2, "the noisy cube is meant to give distinct posteriors"
    eng.close()
    orc.close()


def test_zero_probabilities_are_not_listed(factory):
    """Posterior elements that underflowed to exactly 0 under RecordAnswer, and elements NormalizePriors flushed under ResumeQuiz
    (PqaCore/CENormPriorsSubtaskCorrSum.cpp:32-36), are no candidates: `prob <= 0` is dropped (CEHeapifyPriorsSubtaskMake.cpp:47)."""
    K, Q, T = 3, 8, 300
    init = 0.1
    A, D, B = synth.synthetic_kb(K, Q, T, init, 8.0, 0.5, 21)
    tiny = list(range(40, 140))                     # for these targets answer 0 of every question is next to impossible
    A[:, 0, tiny] = 1e-200
    D[:] = A.sum(axis=1)
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=init))
    assert err is None
    eng.set_option("workers", W)
    eng.set_kb(A, D, B)
    orc = orclib.Oracle(K, Q, T, init)
    orc.set_kb(A, D, B)
    eng.set_target_gaps([7, 100])
    orc.set_target_gaps([7, 100])
    quiz = eng.start_quiz()
    orc.start_quiz(W)
    for q in (0, 1):                                # two answers: (1e-200)^2 underflows
        answer_both(eng, orc, quiz, q, 0)
    post = eng.get_priors(quiz)
    assert np.array_equal(post, orc.priors())
    zeros = [t for t in tiny if t != 100]
    assert all(post[t] == 0.0 for t in zeros) and (post > 0).sum() == T - 2 - len(zeros)
    for n in (1, 10, 32, 150, 250, 256, 290, 400):
        got, want = listed(eng, quiz, n), orc.list_top_targets(n, W)
        assert got == want, n                       # (distinct probabilities: record for record)
        assert len(got) == min(n, T - 2 - len(zeros)) and all(p > 0 for _, p in got)
        assert not {t for t, _ in got} & set(zeros + [7, 100])
    assert len(eng.list_top_targets_batch([quiz], 250)[0]) == T - 2 - len(zeros)
    # ResumeQuiz: four such answers put those targets 2^-2600 below the maximum; NormalizePriors flushes them to 0
    aqs = [(q, 0) for q in (2, 3, 4, 5)]
    for bug in (1, 0):
        eng.set_option("bug_compat", bug)
        rq = eng.resume_quiz([interop.AnsweredQuestion(q, a) for q, a in aqs])
        assert orc.resume_quiz(aqs, W, bool(bug)) == 0
        post = eng.get_priors(rq)
        assert np.array_equal(post, orc.priors()) and all(post[t] == 0.0 for t in zeros)
        for n in (1, 10, 200, 256, 300):
            got, want = listed(eng, rq, n), orc.list_top_targets(n, W)
            same_listing(got, want, post, [7, 100])
            assert len(got) == min(n, int((post > 0).sum()) - int(post[7] > 0) - int(post[100] > 0))
    # nothing positive but gaps: an empty listing, not an error
    eng.set_target_gaps([t for t in range(T) if t not in zeros])
    assert listed(eng, quiz, 5) == [] and eng.list_top_targets_batch([quiz, quiz], 5) == [[], []]
    eng.close()
    orc.close()


def test_equal_probabilities_come_by_ascending_target(factory):
    """A fresh knowledge base: every target holds the same probability.  The reference's order is its heaps' (a function of the
    thread count); the engine's is the target index.  Everything else agrees."""
    K, Q, T = 3, 5, 777
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.5))
    assert err is None
    eng.set_option("workers", W)
    orc = orclib.Oracle(K, Q, T, 0.5)
    eng.set_target_gaps([0, 5])
    orc.set_target_gaps([0, 5])
    quiz = eng.start_quiz()
    orc.start_quiz(W)
    post = eng.get_priors(quiz)
    assert np.array_equal(post, orc.priors())
    for n in (1, 7, 32, 100, 300, 775, 1000):
        got, want = listed(eng, quiz, n), orc.list_top_targets(n, W)
        assert [t for t, _ in got] == [t for t in range(T) if t not in (0, 5)][:n]
        same_listing(got, want, post, [0, 5])
    # after an answer on a fresh cube the probabilities still tie
    answer_both(eng, orc, quiz, 2, 1)
    post = eng.get_priors(quiz)
    same_listing(listed(eng, quiz, 10), orc.list_top_targets(10, W), post, [0, 5])
    eng.close()
    orc.close()


@pytest.mark.parametrize("dims,f32", [((2, 4, 16385), False), ((2, 3, 20000), False), ((2, 3, 100000), False), ((5, 4, 100000), True),
                                      ((2, 2, 300000), False)], ids=str)
def test_long_rows_are_listed_on_the_device(factory, dims, f32):
    """Rows beyond one workgroup's registers: chunk lists and their merge (two merge levels at 300000 targets x 256), single calls
    and batches, against the oracle."""
    K, Q, T = dims
    eng, orc = make(factory, K, Q, T, seed=T % 1000, f32=f32)
    gaps = [0, 4095, 4096, T // 3, T - 1]
    eng.set_target_gaps(gaps)
    orc.set_target_gaps(gaps)
    rng = np.random.default_rng(T)
    quizzes = []
    states = []
    for i in range(3):
        quiz = eng.start_quiz()
        orc.start_quiz(W)
        for step in range(i + 1):
            answer_both(eng, orc, quiz, (i + step) % Q, int(rng.integers(K)))
        post = eng.get_priors(quiz)
        assert np.array_equal(post, orc.priors())
        quizzes.append(quiz)
        states.append((post, {n: orc.list_top_targets(n, W) for n in (1, 10, 33, 256)}))
    for quiz, (post, want) in zip(quizzes, states):
        for n in (1, 10, 33, 256):
            got = listed(eng, quiz, n)
            same_listing(got, want[n], post, gaps)
            assert len(got) == n
    for n in (1, 10, 256):
        batch = eng.list_top_targets_batch(quizzes + quizzes[::-1], n)
        assert len(batch) == 6
        for lst, quiz in zip(batch, quizzes + quizzes[::-1]):
            assert [(r.i_target, r.prob) for r in lst] == listed(eng, quiz, n)
    got = listed(eng, quizzes[0], 300)                      # longer than the device path lists: host-side, same rule
    same_listing(got, orc_list_for(states[0][0], gaps, 300), states[0][0], gaps)
    eng.close()
    orc.close()


def orc_list_for(post, gaps, n):
    """(the oracle object holds the LAST quiz's state; an earlier state's long listing by definition: distinct values)"""
    live = [t for t in range(len(post)) if t not in gaps and post[t] > 0]
    return [(t, post[t]) for t in sorted(live, key=lambda t: (-post[t], t))[:n]]


def test_batched_listing_equals_quiz_by_quiz(factory):
    """PqaEngine_ListTopTargetsBatch: more quizzes than a launch sequence takes (256), every quiz in a state of its own."""
    K, Q, T = 4, 20, 5000
    eng, orc = make(factory, K, Q, T, seed=9)
    rng = np.random.default_rng(2)
    quizzes = eng.start_quiz_batch(300)
    for i, quiz in enumerate(quizzes):
        for step in range(i % 4):
            eng.set_active_question(quiz, int(rng.integers(Q)) if step else i % Q)
            eng.record_answer(quiz, int(rng.integers(K)))
    for n in (1, 10, 40):
        batch = eng.list_top_targets_batch(quizzes, n)
        for i in (0, 1, 2, 3, 129, 255, 256, 257, 299):
            assert [(r.i_target, r.prob) for r in batch[i]] == listed(eng, quizzes[i], n), (n, i)
        assert all(len(b) == n for b in batch)
    # the oracle on one of them
    orc.start_quiz(W)
    quiz = eng.start_quiz()
    answer_both(eng, orc, quiz, 3, 2)
    answer_both(eng, orc, quiz, 11, 0)
    assert [(r.i_target, r.prob) for r in eng.list_top_targets_batch([quiz], 25)[0]] == orc.list_top_targets(25, W)
    assert eng.list_top_targets_batch([], 5) == [] and eng.list_top_targets_batch(quizzes[:3], 0) == [[], [], []]
    with pytest.raises(interop.PqaException, match="absent|out of range|registry"):
        eng.list_top_targets_batch([quizzes[0], 10_000], 3)
    # a deferred RecordAnswer of a listed quiz is applied first
    eng.set_active_question(quizzes[5], 7)
    eng.record_answer(quizzes[5], 1)
    assert [(r.i_target, r.prob) for r in eng.list_top_targets_batch([quizzes[5]], 6)[0]] == listed(eng, quizzes[5], 6)
    eng.close()
    orc.close()


def test_batched_listing_through_the_sharded_engine(factory):
    from test_gpu_sharded import devices

    K, Q, T = 3, 12, 20000
    with devices("0,0,0"):
        sh, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    assert err is None and sh.get_option("shards") == 3
    sh.set_option("workers", W)
    A, D, B = synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, 4)
    sh.set_kb(A, D, B)
    orc = orclib.Oracle(K, Q, T, 0.1)
    orc.set_kb(A, D, B)
    a, b = sh.start_quiz(), sh.start_quiz()
    orc.start_quiz(W)
    for q, ans in ((1, 2), (6, 0), (11, 1)):        # questions of three different shards
        answer_both(sh, orc, a, q, ans)
    want = orc.list_top_targets(12, W)
    assert listed(sh, a, 12) == want
    batch = sh.list_top_targets_batch([b, a], 12)
    assert [(r.i_target, r.prob) for r in batch[1]] == want and [(r.i_target, r.prob) for r in batch[0]] == listed(sh, b, 12)
    sh.close()
    orc.close()


def test_configs4_listing_rate(factory):
    """256 quizzes x top-10 over 100000 targets (BASELINE configs[4]'s quiz batch on one shard) in one C-ABI call: the round's bar is
    1 ms (tools/top_targets_bench.py: ~0.1 ms; the Python wrapper's RatedTarget objects cost several times the device work, so the
    call is timed beneath it)."""
    import ctypes

    K, Q, T, NQ, N = 5, 4, 100000, 256, 10
    eng, _ = make(factory, K, Q, T, seed=1, f32=True)
    quizzes = eng.start_quiz_batch(NQ)
    rng = np.random.default_rng(0)
    eng.next_question_argmax_batch(quizzes)
    eng.record_answer_batch(quizzes, [int(x) for x in rng.integers(0, K, size=NQ)])
    want = eng.list_top_targets_batch(quizzes, N)
    c_quizzes = (ctypes.c_int64 * NQ)(*quizzes)
    c_counts = (ctypes.c_int64 * NQ)()
    c_dest = (interop.CiRatedTarget * (NQ * N))()
    times = []
    for _ in range(30):
        t0 = time.perf_counter()
        err = interop._lib.PqaEngine_ListTopTargetsBatch(eng.c_engine, NQ, c_quizzes, N, c_dest, c_counts)
        times.append(time.perf_counter() - t0)
        assert not err
    assert list(c_counts) == [N] * NQ
    assert [(c_dest[5 * N + j].iTarget, c_dest[5 * N + j].prob) for j in range(N)] == [(r.i_target, r.prob) for r in want[5]]
    med = sorted(times)[len(times) // 2]
    print("256 x top-10 at T=100000: median %.0f us per C-ABI call" % (med * 1e6))
    assert med < 1e-3
    eng.close()
