"""The sweep ahead of its request (option "speculate", hip_engine.h: Speculation): StartQuiz, ResumeQuiz and RecordAnswer launch
the sweep the following NextQuestion would launch.  Same questions as an engine that does not speculate and as the oracle, on every step; the result is
dropped when anything touches the quiz, the cube or the gaps in between."""
import numpy as np
import pytest

import cases
from probqa_amd import interop

pytestmark = pytest.mark.gpu

SUBTASKS = 8 * cases.WORKERS


def make(case, factory, speculate, select, f32=False):
    if f32:
        import test_gpu_batch as tb
        eng, _ = tb.float_engine(case, factory)
    else:
        eng = case.make_engine(factory)
    eng.set_option("speculate", speculate)
    eng.set_option("select", select)
    eng.set_option("seed", 77)
    return eng


@pytest.mark.parametrize("select", [0, 1], ids=["sampled", "argmax"])
@pytest.mark.parametrize("dims", [(5, 40, 300), (5, 1000, 1000), (3, 64, 2500)], ids=lambda d: "%dx%dx%d" % (d[1], d[0], d[2]))
def test_same_questions_with_and_without(dims, select, factory):
    K, Q, T = dims
    case = cases.Case("spec", K, Q, T, seed=31)
    a, b = make(case, factory, 1, select), make(case, factory, 0, select)
    orc = case.make_oracle()
    rng = np.random.default_rng(5)
    qa, qb = a.start_quiz(), b.start_quiz()
    orc.start_quiz(cases.WORKERS)
    n_steps = min(12, Q - 1)
    for step in range(n_steps):
        if select == 1:
            ga, gb = a.next_question(qa), b.next_question(qb)
            run, opri = orc.eval(SUBTASKS)
            assert ga == gb == orc.select_argmax(opri), step
        else:
            rnd = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
            ga, gb = a.next_question_sampled(qa, rnd), b.next_question_sampled(qb, rnd)
            run, opri = orc.eval(SUBTASKS)
            assert ga == gb == orc.select_sampled(run, SUBTASKS, rnd), step
        ans = int(rng.integers(0, K))
        a.record_answer(qa, ans)
        b.record_answer(qb, ans)
        orc.record_answer(ga, ans, cases.WORKERS - 1)
        if step % 3 == 2:   # the cached top targets are RecordAnswer's own; asking for them leaves the speculation alone
            ta, tb = a.list_top_targets(qa, 3), b.list_top_targets(qb, 3)
            assert [(t.i_target, t.prob) for t in ta] == [(t.i_target, t.prob) for t in tb]
    assert np.array_equal(a.get_priors(qa), orc.priors())
    # every NextQuestion found its sweep launched (by StartQuiz, then by RecordAnswer); the last RecordAnswer's is still pending
    assert a.get_option("spec_hits") == n_steps
    assert b.get_option("spec_hits") == 0 and b.get_option("spec_dropped") == 0
    a.release_quiz(qa)
    assert a.get_option("spec_dropped") == 1
    a.close()
    b.close()


def test_dropped_when_something_intervenes(factory):
    """Between RecordAnswer and NextQuestion: another quiz's selection, training (the cube changes), new gaps, a changed
    posterior.  The question asked afterwards is the one an engine that never speculates asks in the same state."""
    case = cases.Case("spec_drop", 5, 48, 400, seed=32)
    a, b = make(case, factory, 1, 1), make(case, factory, 0, 1)

    def both(fn):
        ra, rb = fn(a), fn(b)
        assert ra == rb
        return ra

    q1 = both(lambda e: e.start_quiz())
    q2 = both(lambda e: e.start_quiz())
    first = both(lambda e: e.next_question(q1))
    both(lambda e: e.record_answer(q1, 2))
    both(lambda e: e.next_question(q2))                      # another quiz's sweep takes the hand-over buffers
    dropped = a.get_option("spec_dropped")
    assert dropped >= 1
    second = both(lambda e: e.next_question(q1))
    assert second != first
    both(lambda e: e.record_answer(q1, 1))
    both(lambda e: e.train([interop.AnsweredQuestion(3, 1), interop.AnsweredQuestion(7, 4)], 11, 5.0))       # the cube changes under the speculative result
    assert a.get_option("spec_dropped") == dropped + 1
    both(lambda e: e.next_question(q1))
    both(lambda e: e.record_answer(q1, 0))
    nxt = int(np.argmax(b.eval_priorities(q1)))
    both(lambda e: e.set_question_gaps([nxt]))               # the question the speculation has picked goes away
    got = both(lambda e: e.next_question(q1))
    assert got != nxt
    a.set_option("speculate", 1)                             # (six drops in a row: the engine had stopped speculating; start over)
    both(lambda e: e.record_answer(q1, 3))
    both(lambda e: e.record_answer(q2, 3))                   # the newer RecordAnswer's speculation replaces the older one
    both(lambda e: e.next_question(q2))                      # ... and serves that quiz
    assert a.get_option("spec_hits") == 1
    both(lambda e: e.next_question(q1))
    a.close()
    b.close()


def test_stops_speculating_for_a_client_that_never_follows_up(factory):
    case = cases.Case("spec_idle", 5, 32, 200, seed=33)
    eng = make(case, factory, 1, 1)
    quiz = eng.start_quiz()
    for q in range(30):            # answers to questions the client picks itself: no NextQuestion at all
        eng.set_active_question(quiz, q)
        eng.record_answer(quiz, q % 5)
    # after five drops in a row only every 32nd RecordAnswer speculates
    assert eng.get_option("spec_dropped") <= 8
    assert eng.get_option("spec_hits") == 0
    eng.close()


@pytest.mark.parametrize("select", [0, 1], ids=["sampled", "argmax"])
@pytest.mark.parametrize("f32,dims", [(True, (5, 48, 700)), (True, (5, 30, 5000)), (False, (5, 16, 20000)), (True, (3, 16, 17000))],
                         ids=["f32_48x5x700", "f32_30x5x5000", "f64_long_rows", "f32_long_rows"])
def test_float_engines_and_long_rows(f32, dims, select, factory):
    """Where the sweep has no finisher that hands its result over -- Float engines, rows beyond the register shapes -- RecordAnswer
    launches the sweep alone and the sampled selector's kernel follows at NextQuestion: the same questions as without."""
    K, Q, T = dims
    case = cases.Case("spec_f", K, Q, T, seed=41)
    a, b = make(case, factory, 1, select, f32), make(case, factory, 0, select, f32)
    rng = np.random.default_rng(6)
    qa, qb = a.start_quiz(), b.start_quiz()
    n_steps = min(8, Q - 1)
    for step in range(n_steps):
        if select == 1:
            ga, gb = a.next_question(qa), b.next_question(qb)
        else:
            rnd = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
            ga, gb = a.next_question_sampled(qa, rnd), b.next_question_sampled(qb, rnd)
        assert ga == gb, step
        ans = int(rng.integers(0, K))
        a.record_answer(qa, ans)
        b.record_answer(qb, ans)
    assert np.array_equal(a.get_priors(qa), b.get_priors(qb))
    assert a.get_option("spec_hits") == n_steps and b.get_option("spec_hits") == 0
    a.close()
    b.close()


@pytest.mark.parametrize("select", [0, 1], ids=["sampled", "argmax"])
@pytest.mark.parametrize("dims,workers", [((5, 1000, 1000), 16), ((5, 40, 300), 16), ((2, 77, 1024), 16), ((7, 50, 997), 5), ((3, 64, 700), 40)],
                         ids=lambda d: str(d).replace(" ", ""))
def test_update_inside_the_sweeps_launch(dims, workers, select, factory):
    """RecordAnswer's posterior update in the prologue of the speculative sweep (eval_kernels.hip: eval_questions_f64_upd; option
    fuse_update): against an engine that launches the posterior kernel first and against the oracle, on every step -- the
    posterior bit for bit, the listing of the best targets, the question selected next (the answered question is asked from the
    same launch on), with target and question gaps, 2 to 7 answers, rows that fill the 1024-target shape exactly, several worker
    counts (the subtasks of the reference's sum)."""
    K, Q, T = dims
    case = cases.Case("fuse", K, Q, T, seed=T + K, tgaps=[0, T // 2, T - 1], qgaps=[3])
    a, b = make(case, factory, 1, select), make(case, factory, 1, select)
    for e in (a, b):
        e.set_option("workers", workers)
    b.set_option("fuse_update", 0)
    orc = case.make_oracle()
    rng = np.random.default_rng(K + T)
    qa, qb = a.start_quiz(), b.start_quiz()
    orc.start_quiz(workers)
    n_steps = min(14, Q - 2)
    for step in range(n_steps):
        if select == 1:
            ga, gb = a.next_question(qa), b.next_question(qb)
            run, opri = orc.eval(SUBTASKS)
            assert ga == gb == orc.select_argmax(opri), step
        else:
            rnd = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
            ga, gb = a.next_question_sampled(qa, rnd), b.next_question_sampled(qb, rnd)
            run, opri = orc.eval(SUBTASKS)
            assert ga == gb == orc.select_sampled(run, SUBTASKS, rnd), step
        ans = int(rng.integers(0, K))
        a.record_answer(qa, ans)
        b.record_answer(qb, ans)
        orc.record_answer(ga, ans, max(1, workers - 1))
        if step % 2 == 0:
            ta, tb = a.list_top_targets(qa, 5), b.list_top_targets(qb, 5)
            assert [(t.i_target, t.prob) for t in ta] == [(t.i_target, t.prob) for t in tb], step
        if step % 4 == 1:       # (reading the posterior leaves the speculation alone)
            pa = a.get_priors(qa)
            assert np.array_equal(pa, b.get_priors(qb)) and np.array_equal(pa, orc.priors()), step
    assert np.array_equal(a.get_priors(qa), orc.priors())
    assert a.get_option("fused_updates") == n_steps and b.get_option("fused_updates") == 0
    assert a.get_option("spec_hits") == n_steps
    a.close(); b.close()


@pytest.mark.parametrize("dims,workers,why", [((5, 40, 300), 60, "more subtasks than the update's LDS scratch holds"),
                                              ((5, 30, 1500), 16, "rows beyond the 2-pair shape")],
                         ids=["60_workers", "1500_targets"])
def test_update_falls_back_to_its_own_kernel(dims, workers, why, factory):
    """Where the sweep's launch cannot take the update (eval_kernels.hip: EvalFusesUpdate) RecordAnswer launches the posterior
    kernel and the sweep behind it, as before: same posteriors and questions as the oracle, nothing counted as fused."""
    K, Q, T = dims
    case = cases.Case("nofuse", K, Q, T, seed=T + workers)
    eng = make(case, factory, 1, 1)
    eng.set_option("workers", workers)
    orc = case.make_oracle()
    qz = eng.start_quiz()
    orc.start_quiz(workers)
    for step in range(6):
        g = eng.next_question(qz)
        _, opri = orc.eval(SUBTASKS)
        assert g == orc.select_argmax(opri), step
        eng.record_answer(qz, step % K)
        orc.record_answer(g, step % K, max(1, workers - 1))
    assert np.array_equal(eng.get_priors(qz), orc.priors())
    assert eng.get_option("fused_updates") == 0 and eng.get_option("spec_hits") == 6, why
    eng.close()
