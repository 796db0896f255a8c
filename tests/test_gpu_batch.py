"""The row-sharing batched sweep (probqa_amd/csrc/batch_kernels.hip: a lane is a quiz, the cube is read once per batch) and
the Float engine (TPqaPrecisionType::Float: fp32 cube, fp32 sweep arithmetic), through the C ABI.

Oracle: the fp64 CPU restatement (oracle/), run once per quiz -- for Float engines on the cube ROUNDED to fp32, which is the
cube the engine holds ("fp32 config: same generators evaluated in fp64 then rounded").

Tolerances.  fp64 batched sweep: PRIORITY_RTOL = 1e-9 like the single-quiz sweep (a different summation order again: one
serial sum per lane and tile, fp64 totals over the tiles).  fp32: F32_RTOL relative on the priorities of states where the
posterior is not concentrated (the lack term -sum invD^2 / log2 p has a pole at p -> 1, where an fp32 posterior carries 6e-8
of absolute error: SURVEY F3/F4); argmax asserted where the oracle's top-2 margin exceeds 10 x F32_RTOL.
"""
import numpy as np
import pytest

import cases
import orclib
from probqa_amd import interop, synth

pytestmark = pytest.mark.gpu

PRIORITY_RTOL = 1e-9
# fp32 sweep vs fp64 oracle on the same (rounded) cube.  An fp32 likelihood / posterior element carries 2^-23..2^-22 of relative
# rounding error, and how much of it reaches a priority is a property of the QUESTION AND QUIZ STATE, not of the kernel: the
# velocity sum of (posterior - prior)^2 cancels where an answer barely moves the posterior, vComp^9 = (ln sqrt2 - ln avgV + ..)^-9
# multiplies avgV's relative error by 9 vComp, the lack sum -sum invD^2 / log2 p has a pole at p -> 1.  The stated tolerance
# therefore is, per question,
#     F32_RTOL  +  F32_AMPLIFY x (relative change of the fp64 priority when every likelihood, and every answer row's 1/W_k, is
#                                  perturbed by 2^-23 relative)
# with the second term measured numerically (f32_tolerance: the priority formula of CEEvalQsSubtaskConsider.cpp:62-207 in numpy
# fp64, four random perturbations).  The tests print the measured maxima as fractions of this bound.
F32_RTOL = 5e-4
F32_AMPLIFY = 6.0

# the long-row sweep's forms (option cluster_form) and, for the form that runs ahead, its shapes (option cluster_shape:
# cluster_kernels.hip kAheadVariants)
CLUSTER_FORMS = [(1, 0), (2, 1), (2, 2)]
CLUSTER_FORM_IDS = ["by_question", "ahead", "ahead_256x2"]
CLUSTER_SHAPE_SUFFIX = {0: "", 1: "", 2: "_256x2"}



def scripted_quizzes(case, eng, n_quizzes, rng):
    """n_quizzes quizzes with different answer histories (0..3 answers); returns [(quiz id, [(q, a), ...])]."""
    out = []
    valid_q = [q for q in range(case.Q) if q not in case.qgaps]
    for i in range(n_quizzes):
        quiz = eng.start_quiz()
        hist = []
        for q in rng.choice(valid_q, size=min(i % 4, len(valid_q) - 1), replace=False):
            a = int(rng.integers(case.K))
            eng.set_active_question(quiz, int(q))
            eng.record_answer(quiz, a)
            hist.append((int(q), a))
        out.append((quiz, hist))
    return out


def oracle_priorities(orc, hist):
    orc.start_quiz(cases.WORKERS)
    for q, a in hist:
        orc.record_answer(q, a, cases.WORKERS - 1)
    _, pri = orc.eval(1)
    return pri, orc.priors()


def np_priorities(like, inv_d2, prior, n_valid, post_scale=None):
    """priority[q] from likelihoods like[q,k,t] (0 at gaps), invD^2[q,t] (0 at gaps), masked prior[t]: the formula of
    CEEvalQsSubtaskConsider.cpp:62-207 in numpy fp64 (library log2; gap / zero elements contribute nothing)."""
    w = like.sum(axis=2, keepdims=True)
    post = like / np.where(w > 0, w, 1)
    if post_scale is not None:
        post = post * post_scale                                        # a rounded 1/W_k moves a whole answer row together
    with np.errstate(divide="ignore", invalid="ignore"):
        l2 = np.where(post > 0, np.log2(np.where(post > 0, post, 1)), -1023.0)
        l2 = np.minimum(l2, -1e-19)                                     # Log2Hot(1) is (just) negative
        h = -(post * l2).sum(axis=2)                                      # [Q, K]
        lack = -(inv_d2[:, None, :] / l2).sum(axis=(1, 2))
    v = np.sqrt(((post - prior[None, None, :]) ** 2).sum(axis=2))
    wk = w[:, :, 0]
    tot = np.maximum(wk.sum(axis=1), 1e-300)
    avg_h, avg_v = (wk * h).sum(axis=1) / tot, (wk * v).sum(axis=1) / tot
    ln_sqrt2 = 0.34657359027997264
    vcomp = 1.0 / (ln_sqrt2 - np.log(np.maximum(avg_v, 1e-300)) + ln_sqrt2 / (n_valid + 1) ** 2)
    return lack * vcomp ** 9 / np.exp2(avg_h) ** 2


def f32_tolerance(orc, case):
    """Per-question tolerance of an fp32 sweep for the oracle's CURRENT quiz state (see F32_RTOL / F32_AMPLIFY)."""
    T = case.T
    prior = orc.priors().copy()
    inv_d = 1.0 / orc.D[:, :T]
    for t in case.tgaps:
        prior[t] = 0
        inv_d[:, t] = 0
    like = orc.A[:, :, :T] * inv_d[:, None, :] * prior[None, None, :]
    n_valid = T - len(case.tgaps)
    base = np_priorities(like, inv_d ** 2, prior, n_valid)
    rng = np.random.default_rng(12345)
    dev = np.zeros_like(base)
    for _ in range(4):
        noisy = like * (1 + 2.0 ** -23 * rng.uniform(-1, 1, size=like.shape))
        scale = 1 + 2.0 ** -23 * rng.uniform(-1, 1, size=like.shape[:2] + (1,))
        dev = np.maximum(dev, np.abs(np_priorities(noisy, inv_d ** 2, prior, n_valid, scale) - base) / np.maximum(np.abs(base), 1e-300))
    return F32_RTOL + F32_AMPLIFY * dev


def rel_vec(pri, opri):
    assert ((opri == 0) == (pri == 0)).all(), "gap / asked questions must have priority 0, and only they"
    return np.where(opri != 0, cases.rel_err(pri, np.where(opri != 0, opri, 1)), 0.0)


def rel_to(pri, opri):
    assert ((opri == 0) == (pri == 0)).all(), "gap / asked questions must have priority 0, and only they"
    nz = opri != 0
    return cases.rel_err(pri[nz], opri[nz]).max() if nz.any() else 0.0


@pytest.mark.parametrize("case", cases.small_cases(), ids=lambda c: c.name)
@pytest.mark.parametrize("n_quizzes,tile", [(70, 0), (5, 64), (256, 0)], ids=["70q", "5q_tile64", "256q"])
def test_fp64_batched_sweep_against_oracle_and_single(case, n_quizzes, tile, factory):
    """Every quiz of a batch gets the priorities the single-quiz sweep and the oracle give it; the batch's selections are
    the oracle's argmaxes.  tile=64 forces rows of > 64 targets through several LDS tiles (pass 2 re-stages)."""
    if n_quizzes == 256 and case.Q > 100:
        pytest.skip("256 oracle sweeps of the larger cubes are covered by the 70-quiz form")
    rng = np.random.default_rng(77)
    eng = case.make_engine(factory)
    eng.set_option("batch_min", 1)
    eng.set_option("batch_tile", tile)
    orc = case.make_oracle()
    quizzes = scripted_quizzes(case, eng, n_quizzes, rng)
    ids = [q for q, _ in quizzes]
    pri_b = eng.eval_priorities_batch(ids, case.Q)
    picks = eng.next_question_argmax_batch(ids)
    worst = 0.0
    for i, (quiz, hist) in enumerate(quizzes):
        if i % 7 == 0 or n_quizzes <= 8:       # the single-quiz sweep on a subset (it is itself held to the oracle elsewhere)
            assert rel_to(pri_b[i], eng.eval_priorities(quiz)) < PRIORITY_RTOL
        opri, opriors = oracle_priorities(orc, hist)
        assert np.array_equal(eng.get_priors(quiz), opriors)
        r = rel_to(pri_b[i], opri)
        worst = max(worst, r)
        assert r < PRIORITY_RTOL, f"quiz {i} ({hist}): {r:g}"
        top = np.sort(opri)[::-1]
        if top[0] > 0 and (top[0] - top[1]) / top[0] > 10 * PRIORITY_RTOL:
            assert picks[i] == orc.select_argmax(opri), f"quiz {i}: argmax"
    print(case.name, n_quizzes, "quizzes, tile", tile, ": max priority rel err %.2e" % worst)
    eng.close()


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("n_quizzes,groups,tile", [(5, 8, 0), (20, 8, 64), (32, 4, 0), (33, 4, 128), (64, 2, 0), (100, 2, 64), (128, 0, 0)],
                         ids=lambda v: str(v))
def test_small_batches_share_a_workgroup_between_question_groups(prec, n_quizzes, groups, tile, factory):
    """Batches of up to 128 quizzes: the lanes a small batch leaves over take further questions (batch_kernels.hip:
    eval_batch_kernel, lane = (question group, quiz)).  The automatic rule takes the grouped form only on cubes with hundreds of
    questions per CU-full of workgroups; here it is forced (option batch_groups) on cubes whose question counts leave the last
    block ragged, over several LDS tiles, gaps included: every quiz's priorities against the oracle, the picks its argmaxes."""
    case = cases.Case("groups_%s" % prec, 5, 117, 700, seed=5, qgaps=[0, 31, 116], tgaps=[3, 698])
    rng = np.random.default_rng(4242)
    if prec == "f64":
        eng, orc = case.make_engine(factory), case.make_oracle()
    else:
        eng, orc = float_engine(case, factory)
    eng.set_option("batch_min", 1)
    eng.set_option("batch_form", 2)
    eng.set_option("batch_groups", groups)
    eng.set_option("batch_tile", tile)
    quizzes = scripted_quizzes(case, eng, n_quizzes, rng)
    ids = [q for q, _ in quizzes]
    pri_b = eng.eval_priorities_batch(ids, case.Q)
    picks = eng.next_question_argmax_batch(ids)
    for i, (quiz, hist) in enumerate(quizzes):
        opri, opriors = oracle_priorities(orc, hist)
        assert np.array_equal(eng.get_priors(quiz), opriors)
        tol = PRIORITY_RTOL if prec == "f64" else f32_tolerance(orc, case)
        r = rel_vec(pri_b[i], opri)
        assert (r < tol).all(), f"quiz {i} ({hist}): {r.max():g}"
        top = np.sort(opri)[::-1]
        if top[0] > 0 and (top[0] - top[1]) / top[0] > 10 * np.max(tol):
            assert picks[i] == orc.select_argmax(opri), f"quiz {i}: argmax"
    eng.close()


def float_engine(case, factory):
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(case.K, case.Q, case.T, init_amount=case.init,
                                                                  prec_type=interop.PrecisionType.FLOAT, prec_exponent=8,
                                                                  prec_mantissa=24))
    assert err is None and eng is not None, err
    assert eng.get_option("precision") == 1
    A, D, B = case.kb()
    eng.set_kb(A, D, B)
    eng.set_option("workers", cases.WORKERS)
    if case.tgaps:
        eng.set_target_gaps(case.tgaps)
    if case.qgaps:
        eng.set_question_gaps(case.qgaps)
    orc = orclib.Oracle(case.K, case.Q, case.T, case.init)
    orc.set_kb(*(x.astype(np.float32).astype(np.float64) for x in (A, D, B)))
    orc.set_target_gaps(case.tgaps)
    orc.set_question_gaps(case.qgaps)
    return eng, orc


@pytest.mark.parametrize("case", cases.small_cases(), ids=lambda c: c.name)
def test_float_engine_against_fp64_oracle_on_the_rounded_cube(case, factory):
    """TPqaPrecisionType::Float: the cube is fp32 (GetKB returns the rounded values), posteriors are fp64 products of the
    rounded likelihood ratios (bit-identical to the oracle on the rounded cube), the sweep is fp32: priorities within
    F32_RTOL of the fp64 oracle, batched and single-quiz forms alike, argmax where the margin allows."""
    rng = np.random.default_rng(99)
    eng, orc = float_engine(case, factory)
    A, D, B = eng.get_kb(case.Q)
    assert np.array_equal(A, orc.A[:, :, : case.T]) and np.array_equal(D, orc.D[:, : case.T]) and np.array_equal(B, orc.B[: case.T])
    quizzes = scripted_quizzes(case, eng, 70, rng)
    ids = [q for q, _ in quizzes]
    pri_b = eng.eval_priorities_batch(ids, case.Q)
    picks = eng.next_question_argmax_batch(ids)
    worst_b = worst_s = worst_f = 0.0
    for i, (quiz, hist) in enumerate(quizzes):
        opri, opriors = oracle_priorities(orc, hist)
        assert np.array_equal(eng.get_priors(quiz), opriors), f"quiz {i}: posterior"
        tol = f32_tolerance(orc, case)
        rb = rel_vec(pri_b[i], opri)
        worst_b = max(worst_b, (rb / tol).max())
        assert (rb < tol).all(), f"quiz {i} ({hist}): batched fp32 sweep {rb.max():g}, {(rb / tol).max():g} of the tolerance"
        if i % 5 == 0:
            pri_s = eng.eval_priorities(quiz)
            rs = rel_vec(pri_s, opri)
            worst_s = max(worst_s, (rs / tol).max())
            worst_f = max(worst_f, (rel_vec(pri_s, pri_b[i]) / tol).max())
            assert (rs < tol).all() and (rel_vec(pri_s, pri_b[i]) < 2 * tol).all()
        top = np.sort(opri)[::-1]
        if top[0] > 0 and (top[0] - top[1]) / top[0] > 10 * tol.max():
            want = orc.select_argmax(opri)
            assert picks[i] == want, f"quiz {i}: batched argmax"
            if i % 5 == 0:
                assert eng.next_question_argmax(quiz) == want, f"quiz {i}: single-quiz argmax"
    print(case.name, "fp32 vs fp64 oracle, fraction of the per-question tolerance: batched %.2f, single %.2f; forms %.2f" % (worst_b, worst_s, worst_f))
    # the rest of the quiz surface on a Float engine: sampled selector, top targets, training, resume
    quiz, hist = quizzes[-1]
    q = eng.next_question(quiz)
    assert 0 <= q < case.Q and q not in case.qgaps and q not in [h[0] for h in hist]
    eng.record_answer(quiz, 0)
    orc.start_quiz(cases.WORKERS)
    for hq, ha in hist + [(q, 0)]:
        orc.record_answer(hq, ha, cases.WORKERS - 1)
    assert np.array_equal(eng.get_priors(quiz), orc.priors())
    top = eng.list_top_targets(quiz, 3)
    best = np.argsort(-orc.priors(), kind="stable")[:3]
    assert [t.i_target for t in top] == best.tolist()
    t0 = next(t for t in range(case.T) if t not in case.tgaps)
    eng.train([interop.AnsweredQuestion(hq, ha) for hq, ha in hist[:1]] or [interop.AnsweredQuestion(q, 0)], t0, 1.5)
    A2, D2, _ = eng.get_kb(case.Q)
    hq, ha = (hist[:1] or [(q, 0)])[0]
    want = (np.sqrt(A[hq, ha, t0]) * 3.0 + 2.25) + A[hq, ha, t0]
    assert abs(A2[hq, ha, t0] - want) <= 1.2e-7 * want and abs(D2[hq, t0] - (D[hq, t0] + (want - A[hq, ha, t0]))) <= 2.4e-7 * D2[hq, t0]
    eng.close()


def test_float_engine_kb_file_round_trip(factory, tmp_path):
    case = cases.small_cases()[1]
    eng, _ = float_engine(case, factory)
    path = str(tmp_path / "f32.kb")
    eng.save_kb(path, False)
    eng2, err = factory.load_cpu_engine(path)
    assert err is None and eng2.get_option("precision") == 1
    for a, b in zip(eng.get_kb(case.Q), eng2.get_kb(case.Q)):
        assert np.array_equal(a, b)
    q1, q2 = eng.start_quiz(), eng2.start_quiz()
    assert np.array_equal(eng.eval_priorities(q1), eng2.eval_priorities(q2))
    eng.close()
    eng2.close()


def test_batched_sweep_mid_size_fp64_and_fp32(factory):
    """300 x 5 x 1000 (ldT = 1008 / 1024: four LDS tiles per row by default): 128 quizzes, a seeded subset against the oracle."""
    case = cases.small_cases()[4]
    rng = np.random.default_rng(5)
    for prec in ("f64", "f32"):
        if prec == "f64":
            eng, orc = case.make_engine(factory), case.make_oracle()
            tol = PRIORITY_RTOL
        else:
            eng, orc = float_engine(case, factory)
            tol = None
        eng.set_option("batch_min", 1)
        quizzes = scripted_quizzes(case, eng, 128, rng)
        ids = [q for q, _ in quizzes]
        pri_b = eng.eval_priorities_batch(ids, case.Q)
        worst = 0.0
        for i in (0, 1, 2, 3, 63, 64, 65, 127):
            opri, _ = oracle_priorities(orc, quizzes[i][1])
            r = rel_vec(pri_b[i], opri)
            worst = max(worst, (r / (tol if tol else f32_tolerance(orc, case))).max())
        print("mid", prec, "max rel err as a fraction of the tolerance: %.3g" % worst)
        assert worst < 1
        eng.close()


@pytest.mark.parametrize("T,K,name", [(900, 5, "f32_wg256_nq1"), (1500, 3, "f32_wg256_nq2"), (2500, 5, "f32_wg256_nq3"),
                                     (4000, 5, "f32_wg256_nq4"), (4500, 5, "f32_wg320_nq4"), (6000, 2, "f32_wg384_nq4"),
                                     (8000, 5, "f32_wg512_nq4"), (9000, 7, "f32_wg512_nq5"), (10000, 5, "f32_wg512_nq5"),
                                     (11000, 16, "f32_wg704_nq4"), (12000, 5, "f32_wg768_nq4"),
                                     (16384, 5, "f32_wg1024_nq4"), (20000, 5, "f32_cluster7_x14"), (70000, 3, "f32_cluster18_x14"),
                                     (50000, 2, "f32_cluster"), (30000, 4, "f32_cluster"), (900, 17, "f32_stream")],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("cluster_form,cluster_shape", CLUSTER_FORMS, ids=CLUSTER_FORM_IDS)
def test_float_single_quiz_register_shapes(T, K, name, cluster_form, cluster_shape, factory):
    """The single-quiz sweep of a Float engine, every register shape (eval_f32_kernels.hip), the cluster form for long rows
    (cluster_kernels.hip) and the streaming form behind them:
    against the fp64 oracle on the rounded cube at the fp32 tolerance, after StartQuiz and after three answers, with target and
    question gaps, one question per workgroup and two workgroups streaming all questions; the streaming form (variant 99) on
    the same states within twice the tolerance of the register form."""
    Q = 14
    rng = np.random.default_rng(T)
    tgaps = sorted(rng.choice(T, 7, replace=False).tolist()) + [T - 1]
    case = cases.Case("f32shape_%d" % T, K, Q, T, seed=T, tgaps=sorted(set(tgaps)), qgaps=[3],
                      answers=[(5, 1), (0, K - 1), (13, 0)])
    if cluster_form == 2 and "cluster" not in name:
        pytest.skip("the two forms of the cluster sweep are for rows beyond the register shapes")
    eng, orc = float_engine(case, factory)
    eng.set_option("cluster_form", cluster_form)
    eng.set_option("cluster_shape", cluster_shape)
    if cluster_form == 2:       # (slices by the shape: threads x units per thread, one or two workgroups per CU)
        assert eng.eval_kernel_name().startswith("f32_cluster") and eng.eval_kernel_name().endswith("_ahead" + (CLUSTER_SHAPE_SUFFIX[cluster_shape] if 2 <= K <= 5 else ""))
    else:
        assert eng.eval_kernel_name() == name or (name == "f32_cluster" and eng.eval_kernel_name().startswith(name))
    quiz = eng.start_quiz()
    tol = f32_tolerance(orc, case)
    worst = 0.0
    for step in range(len(case.answers) + 1):
        hist = case.answers[:step]
        opri, opriors = oracle_priorities(orc, hist)
        assert np.array_equal(eng.get_priors(quiz), opriors)
        forms = {}
        for grid in (0, 2):
            eng.set_option("eval_max_grid", grid)
            forms[grid] = eng.eval_priorities(quiz)
            r = rel_vec(forms[grid], opri)
            worst = max(worst, (r / tol).max())
            assert (r < tol).all(), (name, step, grid, float((r / tol).max()))
        eng.set_option("eval_max_grid", 0)
        eng.set_option("eval_variant", 99)
        stream = eng.eval_priorities(quiz)
        eng.set_option("eval_variant", 0)
        assert (rel_vec(stream, forms[0]) < 2 * tol).all()
        assert forms[0][3] == 0 and all(forms[0][q] == 0 for q, _ in hist)      # gap and asked questions
        top = np.sort(opri)[::-1]
        if (top[0] - top[1]) / top[0] > 10 * tol.max():
            assert eng.next_question_argmax(quiz) == orc.select_argmax(opri)
        if step < len(case.answers):
            q, a = case.answers[step]
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
    print(name, "max rel err as a fraction of the tolerance: %.3g" % worst)
    eng.close()


@pytest.mark.parametrize("T,K,name", [(20000, 5, "f64_cluster20_x14"), (40000, 2, "f64_cluster40_x12"), (16500, 9, "f64_cluster26_x14"),
                                     (24000, 3, None), (33001, 4, None), (17000, 6, None)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("cluster_form,cluster_shape", CLUSTER_FORMS, ids=CLUSTER_FORM_IDS)
def test_double_long_rows_cluster_sweep(T, K, name, cluster_form, cluster_shape, factory):
    """Rows beyond the register shapes on a Double engine: the question split over a cluster of workgroups (cluster_kernels.hip).
    Against the oracle at the stated bar, after StartQuiz and after three answers, with gaps; the streaming form (variant 99) on
    the same states; argmax and the sampled selector's pick as the oracle's."""
    Q = 14
    rng = np.random.default_rng(T + K)
    tgaps = sorted(set(rng.choice(T, 9, replace=False).tolist() + [T - 1, 0]))
    case = cases.Case("f64long_%d" % T, K, Q, T, seed=T + 1, tgaps=tgaps, qgaps=[7], answers=[(5, 1), (0, K - 1), (13, 0)])
    eng, orc = case.make_engine(factory), case.make_oracle()
    eng.set_option("cluster_form", cluster_form)
    eng.set_option("cluster_shape", cluster_shape)
    if cluster_shape <= 1 and name is not None:
        assert eng.eval_kernel_name() == name + ("_ahead" if cluster_form == 2 else "")
    else:   # (256 x 2 is built for two to five answers)
        assert eng.eval_kernel_name().startswith("f64_cluster") and eng.eval_kernel_name().endswith(("_ahead" if cluster_form == 2 else "") + (CLUSTER_SHAPE_SUFFIX[cluster_shape] if 2 <= K <= 5 else ""))
    quiz = eng.start_quiz()
    worst = 0.0
    for step in range(len(case.answers) + 1):
        hist = case.answers[:step]
        opri, opriors = oracle_priorities(orc, hist)
        assert np.array_equal(eng.get_priors(quiz), opriors)
        pri = eng.eval_priorities(quiz)
        r = rel_vec(pri, opri)
        worst = max(worst, r.max())
        assert r.max() < PRIORITY_RTOL, (name, step, float(r.max()))
        eng.set_option("eval_variant", 99)
        assert rel_vec(eng.eval_priorities(quiz), pri).max() < PRIORITY_RTOL
        eng.set_option("eval_variant", 0)
        assert pri[7] == 0 and all(pri[q] == 0 for q, _ in hist)
        assert eng.next_question_argmax(quiz) == orc.select_argmax(opri)
        if step < len(case.answers):
            q, a = case.answers[step]
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
    print(name, "max rel err %.3g" % worst)
    eng.close()


@pytest.mark.parametrize("K", [2, 3, 4, 5])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_long_rows_many_questions_per_cluster(K, prec, factory):
    """The form of the long-row sweep that runs ahead (eval_cluster_five_kernel, built for two to five answers), several hundred
    questions per cluster -- the loop's steady state, its last two questions and the folds in between -- against the oracle, sweep
    after sweep on the same state and after answers; a ragged last slice and gaps."""
    Q, T = 700, 17000 + 37 * K
    case = cases.Case("manyq_%s_%d" % (prec, K), K, Q, T, seed=900 + K, tgaps=[0, 5000, T - 1], qgaps=[3, 350, Q - 1],
                      answers=[(5, 1), (Q - 2, K - 1)])
    if prec == "f32":
        eng, orc = float_engine(case, factory)
    else:
        eng, orc = case.make_engine(factory), case.make_oracle()
    assert eng.eval_kernel_name().endswith("_ahead_256x2"), eng.eval_kernel_name()
    quiz = eng.start_quiz()
    for step in range(len(case.answers) + 1):
        opri, opriors = oracle_priorities(orc, case.answers[:step])
        tol = f32_tolerance(orc, case) if prec == "f32" else PRIORITY_RTOL
        first = eng.eval_priorities(quiz)
        assert (rel_vec(first, opri) < tol).all(), (step, float((rel_vec(first, opri) / tol).max()))
        for _ in range(5):
            assert np.array_equal(eng.eval_priorities(quiz), first)     # (the same sums in the same order, whatever the members' timing)
        if step < len(case.answers):
            q, a = case.answers[step]
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
    eng.close()


@pytest.mark.parametrize("case", cases.small_cases() + [cases.Case("trained_200x5x400", 5, 200, 400, seed=61, n_train=8.0, noise=0.1)],
                         ids=lambda c: c.name)
def test_float_batched_argmax_is_the_fp64_argmax(case, factory):
    """north_star: "bit-exact for the argmax index".  A Float engine's batched NextQuestion re-ranks the fp32 sweep's 8 best
    questions per quiz in fp64 (eval_kernels.hip: batch_rerank_kernel), so its pick is the fp64 oracle's argmax on the rounded
    cube whenever that argmax is decided beyond fp64's own 1e-9 -- not only where the margin exceeds the fp32 tolerance, which is
    all round 2 could assert.  Without the re-rank (option rerank = 0) the fp32 argmax is allowed to differ; the share of equal
    picks is printed for both."""
    rng = np.random.default_rng(123)
    eng, orc = float_engine(case, factory)
    eng.set_option("batch_min", 1)
    quizzes = scripted_quizzes(case, eng, 96, rng)
    ids = [q for q, _ in quizzes]
    picks = eng.next_question_argmax_batch(ids)
    eng.set_option("rerank", 0)
    picks32 = eng.next_question_argmax_batch(ids)
    eng.set_option("rerank", 1)
    decided = same = same32 = 0
    for i, (quiz, hist) in enumerate(quizzes):
        opri, _ = oracle_priorities(orc, hist)
        want = orc.select_argmax(opri)
        top = np.sort(opri)[::-1]
        margin = (top[0] - top[1]) / top[0] if len(top) > 1 and top[0] > 0 else 1.0
        same32 += int(picks32[i] == want)
        same += int(picks[i] == want)
        if want >= 0 and margin > PRIORITY_RTOL:
            decided += 1
            assert picks[i] == want, f"quiz {i} ({hist}): picked {picks[i]}, fp64 argmax {want} (margin {margin:.3g})"
        # whatever was picked is one of the oracle's best: its fp64 priority is within 1e-9 of the maximum
        if want >= 0:
            assert opri[picks[i]] >= top[0] * (1 - PRIORITY_RTOL)
    print(case.name, ": fp64 argmax picked in %d / %d quizzes with the re-rank (%d decided beyond 1e-9), %d / %d by fp32 alone"
          % (same, len(quizzes), decided, same32, len(quizzes)))
    eng.close()


@pytest.mark.parametrize("dims", [(5, 40, 300), (4, 30, 20000), (3, 20, 1100)], ids=lambda d: "%dx%dx%d" % (d[1], d[0], d[2]))
def test_batched_posterior_updates_are_bit_identical(dims, factory):
    """PqaEngine_StartQuizBatch / PqaEngine_RecordAnswerBatch: n quizzes' StartQuiz / RecordAnswer in ONE
    launch each (grid.x = quiz; the workgroup code and summation order of the one-quiz kernels), against the same calls one by
    one AND the oracle: posteriors bit for bit, top targets, and the deferred form the engine uses for concurrent clients.
    Rows of 20000 targets: beyond the LDS staging and the in-kernel top listing."""
    K, Q, T = dims
    case = cases.Case("updbatch", K, Q, T, seed=T, qgaps=[2], tgaps=[1, T - 2])
    eng = case.make_engine(factory)
    orc = case.make_oracle()
    n = 150
    rng = np.random.default_rng(T)
    batch = eng.start_quiz_batch(n)
    singles = [eng.start_quiz() for _ in range(n)]
    assert len(set(batch + singles)) == 2 * n
    orc.start_quiz(cases.WORKERS)
    p0 = orc.priors()
    for z in (batch[0], batch[77], batch[-1], singles[3]):
        assert np.array_equal(eng.get_priors(z), p0)
    valid = [q for q in range(Q) if q not in case.qgaps]
    for step in range(2):
        qs = [int(valid[int(rng.integers(len(valid)))]) for _ in range(n)]
        ans = [int(rng.integers(K)) for _ in range(n)]
        for i in range(n):
            for z in (batch[i], singles[i]):
                if step == 1 and qs[i] in [q for q, _ in hist[i]]:
                    qs[i] = next(q for q in valid if q not in [x for x, _ in hist[i]])
                eng.set_active_question(z, qs[i])
        if step == 0:
            hist = [[] for _ in range(n)]
        eng.record_answer_batch(batch, ans)
        for i in range(n):
            eng.record_answer(singles[i], ans[i])
            hist[i].append((qs[i], ans[i]))
        for i in range(0, n, 13):
            a, b = eng.get_priors(batch[i]), eng.get_priors(singles[i])
            assert np.array_equal(a, b), f"step {step} quiz {i}"
            _, opriors = oracle_priorities(orc, hist[i])
            assert np.array_equal(a, opriors), f"step {step} quiz {i}: oracle"
            ta, tb2 = eng.list_top_targets(batch[i], 5), eng.list_top_targets(singles[i], 5)
            assert [(t.i_target, t.prob) for t in ta] == [(t.i_target, t.prob) for t in tb2]
    assert eng.get_option("update_max_flush") >= n
    with pytest.raises(interop.PqaException, match="active question"):
        eng.record_answer_batch([batch[0]], [0])            # no active question any more
    eng.close()


@pytest.mark.parametrize("workers", [16, 2, 7], ids=lambda w: "%dworkers" % w)
@pytest.mark.parametrize("T", [70001, 16390], ids=lambda t: "%dtargets" % t)
def test_long_row_posterior_kernels_are_bit_identical(T, workers, factory):
    """StartQuiz / RecordAnswer over rows beyond 16384 targets (one workgroup per subtask of the reference's sum,
    prior_kernels.hip: long_row_stage_kernel + long_row_divide_kernel) against the one-workgroup kernels (option long_row_form = 0)
    and the oracle, bit for bit: the subtasks' values in LDS (16 workers) and in memory (2 workers: one subtask is the whole row),
    a row that is not a multiple of four targets, target gaps at both ends."""
    K, Q = 3, 12
    case = cases.Case("longrow", K, Q, T, seed=T + workers, qgaps=[4], tgaps=[0, 5, T - 1])
    eng = case.make_engine(factory)
    eng.set_option("workers", workers)
    orc = case.make_oracle()
    quizzes = []
    for form in (1, 0):
        eng.set_option("long_row_form", form)
        quizzes.append(eng.start_quiz())
    orc.start_quiz(workers)
    for z in quizzes:
        assert np.array_equal(eng.get_priors(z), orc.priors())
    for step, (q, a) in enumerate([(3, 1), (0, 2), (11, 0)]):
        for form, z in zip((1, 0), quizzes):
            eng.set_option("long_row_form", form)
            eng.set_active_question(z, q)
            eng.record_answer(z, a)
        orc.record_answer(q, a, max(1, workers - 1))
        p1, p0 = eng.get_priors(quizzes[0]), eng.get_priors(quizzes[1])
        assert np.array_equal(p1, p0), f"step {step}: the two forms"
        assert np.array_equal(p1, orc.priors()), f"step {step}: oracle"
    eng.close()


@pytest.mark.parametrize("dims", [(37, 5, 101), (300, 5, 1000), (64, 5, 50), (90, 5, 2000), (40, 3, 300), (50, 8, 700), (33, 2, 64)],
                         ids=lambda d: "%dx%dx%d" % d)
@pytest.mark.parametrize("n_quizzes", [3, 8, 16, 17, 33, 70], ids=lambda n: "%dq" % n)
def test_fp64_midbatch_sweep_against_oracle(dims, n_quizzes, factory):
    """The sweep for a few dozen quizzes (batch_kernels.hip: eval_midbatch_kernel -- a lane is a (quiz, chunk of the row); what the
    engine's combined sweeps for concurrent clients use): every quiz's priorities against the oracle (1e-9) and against the
    row-sharing sweep (1e-11), the selections = the oracle's argmaxes, with target and question gaps, two to eight answers, 16 / 32 / 64
    quiz slots per wave, rows that do not fill their last chunk, and workgroups that sweep many questions (eval_max_grid)."""
    Q, K, T = dims
    rng = np.random.default_rng(Q + n_quizzes)
    case = cases.Case("mid_%dx%dx%d" % dims, K, Q, T, seed=Q + T, tgaps=sorted(rng.choice(T, 3, replace=False).tolist()),
                      qgaps=sorted(rng.choice(Q, 2, replace=False).tolist()))
    eng = case.make_engine(factory)
    orc = case.make_oracle()
    quizzes = scripted_quizzes(case, eng, n_quizzes, rng)
    ids = [q for q, _ in quizzes]
    pri_rs = eng.eval_priorities_batch(ids, case.Q)                 # row-sharing
    for max_grid in (0, 3):
        eng.set_option("eval_max_grid", max_grid)
        eng.set_option("batch_form", 3)
        pri = eng.eval_priorities_batch(ids, case.Q)
        picks = eng.next_question_argmax_batch(ids)
        eng.set_option("batch_form", 0)
        assert cases.rel_err(pri, pri_rs).max() < 1e-11
        for i, (quiz, hist) in enumerate(quizzes):
            if i % 5 and n_quizzes > 20:
                continue
            opri, opriors = oracle_priorities(orc, hist)
            assert np.array_equal(eng.get_priors(quiz), opriors)
            assert rel_to(pri[i], opri) < PRIORITY_RTOL, f"quiz {i} ({hist})"
            top = np.sort(opri)[::-1]
            if top[0] > 0 and (top[0] - top[1]) / top[0] > 10 * PRIORITY_RTOL:
                assert picks[i] == orc.select_argmax(opri), f"quiz {i}: argmax"
    eng.close()
