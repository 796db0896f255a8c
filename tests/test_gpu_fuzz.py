"""Seeded random cases through the same script as the fixtures (tests/test_gpu_parity.py: run_script): random dimensions
down to the reference's minimum (2 answers, 1 question, 2 targets), random target / question gaps, random answer scripts
that follow the engine's own argmax questions.  Posteriors bit-exact, priorities within the stated tolerance, argmax and the
reference's sampled selector identical -- on every step of every case."""
import numpy as np
import pytest

import cases
from test_gpu_parity import run_script

pytestmark = pytest.mark.gpu


def random_case(i):
    rng = np.random.default_rng(1000 + i)
    if i < 6:   # the corners first
        K, Q, T = [(2, 1, 2), (2, 2, 3), (9, 1, 17), (2, 40, 2), (3, 5, 64), (6, 3, 257)][i]
    else:
        K, Q = int(rng.integers(2, 10)), int(rng.integers(1, 60))
        T = int(rng.integers(2, 400)) if i % 4 else int(rng.integers(400, 3300))   # every 4th crosses kernel shapes
    n_tg = int(rng.integers(0, max(1, T // 4))) if T > 3 else 0
    n_qg = int(rng.integers(0, max(1, Q // 4))) if Q > 3 else 0
    tgaps = sorted(rng.choice(T, n_tg, replace=False).tolist())
    qgaps = sorted(rng.choice(Q, n_qg, replace=False).tolist())
    free_q = [q for q in range(Q) if q not in qgaps]
    n_ans = int(rng.integers(0, min(len(free_q), 5) + 1))
    qs = rng.choice(free_q, n_ans, replace=False).tolist() if n_ans else []
    answers = [(int(q), int(rng.integers(0, K))) for q in qs]
    return cases.Case("fuzz%02d_%dx%dx%d" % (i, Q, K, T), K, Q, T, seed=2000 + i, init=float(rng.choice([0.1, 0.5, 1.0])),
                      n_train=float(rng.choice([0.0, 2.0, 8.0])), noise=float(rng.choice([0.1, 0.5])),
                      tgaps=tgaps, qgaps=qgaps, answers=answers)


@pytest.mark.parametrize("i", range(120))
def test_random_case(i, factory):
    case = random_case(i)
    # every third case: two workgroups stream the questions (instead of one question per workgroup), every fifth through
    # the fused form of the sampled selector, every seventh with the resident sweep where its shape exists
    options = []
    if i % 3 == 0:
        options.append(("eval_max_grid", 2))
    if i % 5 == 0:
        options.append(("fused_sampled", 1))
    if i % 7 == 0:
        options.append(("server", 1))
    worst = max(run_script(case, factory, options))
    assert worst < 1e-9, (case.name, worst)


@pytest.mark.parametrize("i", [841, 2062, 6547])
def test_ill_conditioned_states(i, factory):
    """The offenders of round 3's soak beyond the suite's seeds (tools/fuzz_more.py 120..3770 and 5000..8000: these three of 13700
    runs): knowledge bases of 4 - 10 targets after several answers, a posterior element at p = 1 - 1e-7, where the reference's lack
    term -sum invD^2 / log2(p) has its pole and the last place of W_k -- its summation ORDER -- moves the priority by 1.3 - 2.7e-9.
    The sweeps watch for such rows and a kernel behind them (pole_kernels.hip) redoes the listed questions in the reference's exact
    order -- four serial Kahan lanes and PreciseSum for W_k, Log2Hot operation for operation on the row's largest element.  They are
    held to 1e-9 like every other state, with the launched sweep, two workgroups streaming all questions, and the resident sweep
    (which answers "redo" and hands the quiz to the launched path).  The late states of real quizzes are tests/test_gpu_late.py."""
    case = random_case(i)
    for options in ([], [("eval_max_grid", 2)], [("server", 1)]):
        steps = run_script(case, factory, options)
        assert max(steps) < 1e-9, (case.name, options, max(steps))


@pytest.mark.parametrize("i", range(48))
def test_random_case_batched(i, factory):
    """The same random cases through the row-sharing batched sweep: 1..130 quizzes in different states (prefixes of the case's
    answer script), Double engines (even i, bar 1e-9) and Float engines (odd i, the stated fp32 tolerance), random LDS tile."""
    import orclib
    import test_gpu_batch as tb

    case = random_case(i)
    rng = np.random.default_rng(5000 + i)
    if i % 2 == 0:
        eng, orc = case.make_engine(factory), case.make_oracle()
    else:
        eng, orc = tb.float_engine(case, factory)
    eng.set_option("batch_min", 1)
    eng.set_option("batch_tile", int(rng.choice([0, 64, 128])))
    eng.set_option("batch_qb", int(rng.choice([0, 1, 2])))
    n = int(rng.integers(1, 131))
    quizzes, hists = [], []
    for j in range(n):
        quiz = eng.start_quiz()
        hist = case.answers[: j % (len(case.answers) + 1)]
        for q, a in hist:
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
        quizzes.append(quiz)
        hists.append(hist)
    pri = eng.eval_priorities_batch(quizzes, case.Q)
    picks = eng.next_question_argmax_batch(quizzes)
    seen = {}
    for j in range(n):
        key = len(hists[j])
        if key not in seen:            # the oracle once per distinct state
            opri, opriors = tb.oracle_priorities(orc, hists[j])
            tol = 1e-9 if i % 2 == 0 else tb.f32_tolerance(orc, case)
            seen[key] = (opri, opriors, tol, orc.select_argmax(opri))
        opri, opriors, tol, want = seen[key]
        assert np.array_equal(eng.get_priors(quizzes[j]), opriors)
        rel = tb.rel_vec(pri[j], opri)
        assert (rel < tol).all(), (case.name, j, float(rel.max()))
        top = np.sort(opri)[::-1]
        margin = (top[0] - top[1]) / top[0] if len(top) > 1 and top[0] > 0 else 1.0
        if want < 0:
            assert picks[j] == -1
        elif margin > 10 * np.max(tol):
            assert picks[j] == want, (case.name, j)
    eng.close()
