"""ListTopTargets on the device against the oracle's restatement of the reference's algorithm
(oracle/pqa_oracle.c: orc_list_top_targets -- CEListTopTargetsAlgorithm::RunHeapifyBased, PqaCore/CEListTopTargetsAlgorithm.cpp:30-95,
over the pieces of CEHeapifyPriorsSubtaskMake.cpp:42-88): gaps and probabilities <= 0 are dropped, the rest comes by descending
probability -- and EQUAL probabilities in the order the reference's per-worker heaps and head heap leave them, a function of the
worker count (engine option "workers" = the oracle's nWorkers).  Every listing here is the oracle's RECORD FOR RECORD: targets, the
bits of the probabilities, ties included.  (The engine lists by (probability, target) first; where that listing shows a tie among
the listed targets or at its boundary it reproduces the heaps on the device -- kb_kernels.hip: LaunchTopTargetsExact; engine option
top_exact = 0 keeps the index order, which `same_listing` describes.)"""
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
            assert got == want, (step, n)
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
            assert got == want, (bug, n)
            assert len(got) == min(n, int((post > 0).sum()) - int(post[7] > 0) - int(post[100] > 0))
    # nothing positive but gaps: an empty listing, not an error
    eng.set_target_gaps([t for t in range(T) if t not in zeros])
    assert listed(eng, quiz, 5) == [] and eng.list_top_targets_batch([quiz, quiz], 5) == [[], []]
    eng.close()
    orc.close()


def test_equal_probabilities_come_in_the_heaps_order(factory):
    """A fresh knowledge base: every target holds the same probability, and what the reference lists is decided by its heaps -- by the
    worker count.  The engine reproduces them (for 1, 4, 16 and 61 workers: different listings, none by index); with option
    top_exact = 0 it lists by ascending target, everything else the same."""
    K, Q, T = 3, 5, 777
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.5))
    assert err is None
    orc = orclib.Oracle(K, Q, T, 0.5)
    eng.set_target_gaps([0, 5])
    orc.set_target_gaps([0, 5])
    seen = set()
    for workers in (1, 4, 16, 61):
        eng.set_option("workers", workers)
        quiz = eng.start_quiz()
        orc.start_quiz(workers)
        post = eng.get_priors(quiz)
        assert np.array_equal(post, orc.priors())
        before = eng.get_option("top_exact_listings")
        for n in (1, 7, 32, 100, 256, 300, 775, 1000):
            got, want = listed(eng, quiz, n), orc.list_top_targets(n, workers)
            assert got == want, (workers, n)
        assert eng.get_option("top_exact_listings") > before
        seen.add(tuple(t for t, _ in listed(eng, quiz, 12)))
        eng.set_option("top_exact", 0)
        for n in (1, 7, 300):
            got = listed(eng, quiz, n)
            assert [t for t, _ in got] == [t for t in range(T) if t not in (0, 5)][:n]
            same_listing(got, orc.list_top_targets(n, workers), post, [0, 5])
        eng.set_option("top_exact", 1)
        # after an answer on a fresh cube the probabilities still tie
        eng.set_active_question(quiz, 2)
        eng.record_answer(quiz, 1)
        orc.record_answer(2, 1, max(1, workers - 1))
        assert np.array_equal(eng.get_priors(quiz), orc.priors())
        for n in (1, 10, 40):
            assert listed(eng, quiz, n) == orc.list_top_targets(n, workers), (workers, n)
        assert [[(r.i_target, r.prob) for r in lst] for lst in eng.list_top_targets_batch([quiz, quiz], 10)] == [orc.list_top_targets(10, workers)] * 2
    assert len(seen) == 4
    eng.close()
    orc.close()


def test_few_distinct_values_many_ties(factory):
    """Posteriors of a cube trained without noise: a handful of distinct values, hundreds of targets each -- ties inside the list, at
    its boundary, across pieces.  Single listings, batches, long rows; against the oracle record for record."""
    # (the last two: pieces of more than 8192 candidates -- their heaps live in global memory, not LDS)
    for (K, Q, T, workers) in ((5, 40, 1000, 16), (5, 12, 5000, 7), (4, 6, 40000, 16), (3, 4, 20000, 3), (4, 5, 40000, 3), (2, 3, 100000, 1),
                               (2, 3, 24577, 3)):     # (pieces of 8193 / 8192 / 8192: the largest decides for all of them)
        eng, orc = make(factory, K, Q, T, seed=3, noise=0.0)
        eng.set_option("workers", workers)
        gaps = [1, T // 2]
        eng.set_target_gaps(gaps)
        orc.set_target_gaps(gaps)
        quizzes = []
        orc.start_quiz(workers)
        quiz = eng.start_quiz()
        for step in range(4):
            q, a = (step * 7 + 3) % Q, step % K
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
            orc.record_answer(q, a, max(1, workers - 1))
            post = eng.get_priors(quiz)
            assert np.array_equal(post, orc.priors())
            assert len(np.unique(post)) < T // 4                       # (many ties)
            for n in (1, 3, 10, 33, 256):
                want = orc.list_top_targets(n, workers)
                assert listed(eng, quiz, n) == want, (T, step, n)
                assert [(r.i_target, r.prob) for r in eng.list_top_targets_batch([quiz], n)[0]] == want, (T, step, n)
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
            assert got == want[n], n
            assert len(got) == n
    for n in (1, 10, 256):
        batch = eng.list_top_targets_batch(quizzes + quizzes[::-1], n)
        assert len(batch) == 6
        for lst, quiz in zip(batch, quizzes + quizzes[::-1]):
            assert [(r.i_target, r.prob) for r in lst] == listed(eng, quiz, n)
    got = listed(eng, quizzes[0], 300)                      # longer than the device path lists: host-side
    assert got == orc_list_for(states[0][0], gaps, 300)     # (distinct values: the sort's order)
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


def test_batched_listing_of_tied_quizzes_beyond_one_launch(factory):
    """300 quizzes on a cube trained without noise, each a few answers in: tied listings in most of them, more than one launch sequence
    of the heaps' path; every one the oracle's."""
    K, Q, T = 5, 30, 3000
    eng, orc = make(factory, K, Q, T, seed=6, noise=0.0)
    quizzes = eng.start_quiz_batch(300)
    hist = {}
    for i, quiz in enumerate(quizzes):
        h = [((i + 3 * s) % Q, (i + s) % K) for s in range(i % 3)]
        for q, a in h:
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
        hist[quiz] = h
    before = eng.get_option("top_exact_listings")
    batch = eng.list_top_targets_batch(quizzes, 10)
    assert eng.get_option("top_exact_listings") - before > 256
    for i in (0, 1, 2, 17, 100, 255, 256, 257, 299):
        orc.start_quiz(W)
        for q, a in hist[quizzes[i]]:
            orc.record_answer(q, a, W - 1)
        assert [(r.i_target, r.prob) for r in batch[i]] == orc.list_top_targets(10, W), i
        assert listed(eng, quizzes[i], 10) == orc.list_top_targets(10, W), i
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
