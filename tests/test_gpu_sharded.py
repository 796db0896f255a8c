"""One knowledge base over several shards of ONE process behind the plain C ABI (probqa_amd/csrc/sharded_engine.cpp):
PQA_DEVICES lists a device per shard and PqaEngineFactory_CreateCpuEngine -- the only factory call the reference's wrappers can
make -- returns an engine whose every call fans out over the shards.  On a single-GPU box the ordinal 0 is listed several times
(several shards on one device): the same code path as eight GPUs, minus the peer copies crossing xGMI.

Held to: a whole-cube engine (same selections, bit-identical posteriors) AND the oracle, step by step."""
import os

import numpy as np
import pytest

import cases
import orclib
from probqa_amd import interop

pytestmark = pytest.mark.gpu

PRIORITY_RTOL = 1e-9
SUBTASKS = 8 * cases.WORKERS


class devices:
    def __init__(self, spec):
        self.spec = spec

    def __enter__(self):
        self.saved = os.environ.get("PQA_DEVICES")
        os.environ["PQA_DEVICES"] = self.spec

    def __exit__(self, *a):
        if self.saved is None:
            os.environ.pop("PQA_DEVICES", None)
        else:
            os.environ["PQA_DEVICES"] = self.saved


@pytest.mark.parametrize("case", [cases.small_cases()[1], cases.small_cases()[2], cases.small_cases()[4]], ids=lambda c: c.name)
@pytest.mark.parametrize("spec", ["0,0", "0,0,0,0,0"], ids=["2shards", "5shards"])
def test_sharded_engine_behind_the_plain_abi(case, spec, factory):
    with devices(spec):
        sh = case.make_engine(factory)            # PqaEngineFactory_CreateCpuEngine, nothing else
    assert sh.get_option("shards") == spec.count(",") + 1
    whole = case.make_engine(factory)
    assert whole.get_option("shards") == -1
    orc = case.make_oracle()
    dims = sh.copy_dims()
    assert (dims.n_answers, dims.n_questions, dims.n_targets) == (case.K, case.Q, case.T)
    for a, b in zip(sh.get_kb(case.Q), whole.get_kb(case.Q)):
        assert np.array_equal(a, b)
    qs, qw = sh.start_quiz(), whole.start_quiz()
    orc.start_quiz(cases.WORKERS)
    valid = [q for q in range(case.Q) if q not in case.qgaps]
    rng = np.random.default_rng(3)
    for step in range(6):
        assert np.array_equal(sh.get_priors(qs), orc.priors()), f"step {step}: posterior"
        pri = sh.eval_priorities(qs, case.Q)
        run, opri = orc.eval(SUBTASKS)
        assert ((opri == 0) == (pri == 0)).all() and cases.rel_err(pri[opri != 0], opri[opri != 0]).max() < PRIORITY_RTOL
        assert np.array_equal(pri, whole.eval_priorities(qw))          # same kernels over the same rows
        want = orc.select_argmax(opri)
        assert sh.next_question_argmax(qs) == want == whole.next_question_argmax(qw)
        assert sh.get_active_question_id(qs) == want
        for rnd in (0, 1, 2**63, 2**64 - 1, 0x9E3779B97F4A7C15):
            assert sh.next_question_sampled(qs, rnd) == orc.select_sampled(run, SUBTASKS, rnd) == whole.next_question_sampled(qw, rnd), hex(rnd)
        assert sh.next_question_argmax_batch([qs]) == [want]
        # answer a question of a (mostly) different shard every step
        q = valid[(step * len(valid)) // 6 + int(rng.integers(3))]
        while q in [x for x, _ in orc.answers]:
            q = valid[(valid.index(q) + 1) % len(valid)]
        a = int(rng.integers(case.K))
        for e, z in ((sh, qs), (whole, qw)):
            e.set_active_question(z, q)
            e.record_answer(z, a)
        orc.record_answer(q, a, cases.WORKERS - 1)
        ts, tw = sh.list_top_targets(qs, 4), whole.list_top_targets(qw, 4)
        assert [(t.i_target, t.prob) for t in ts] == [(t.i_target, t.prob) for t in tw]
    # ResumeQuiz across shards (the answered questions' rows live on different shards), both settings of the :53 switch
    aqs = list(orc.answers)
    for bug in (1, 0):
        sh.set_option("bug_compat", bug)
        o2 = case.make_oracle()
        assert o2.resume_quiz(aqs, cases.WORKERS, bool(bug)) == 0
        q2 = sh.resume_quiz([interop.AnsweredQuestion(q, a) for q, a in aqs])
        assert np.array_equal(sh.get_priors(q2), o2.priors())
        _, opri = o2.eval(SUBTASKS)
        pri = sh.eval_priorities(q2, case.Q)
        assert ((opri == 0) == (pri == 0)).all() and cases.rel_err(pri[opri != 0], opri[opri != 0]).max() < PRIORITY_RTOL
        sh.release_quiz(q2)
    # training reaches every shard's rows and every vB replica
    t0 = next(t for t in range(case.T) if t not in case.tgaps)
    train = [(valid[0], 1), (valid[-1], 0), (valid[len(valid) // 2], 2), (valid[0], 1)]
    sh.train([interop.AnsweredQuestion(q, a) for q, a in train], t0, 0.7)
    orc.train(train, t0, 0.7, cases.WORKERS)
    sh.record_quiz_target(qs, t0, 1.1)
    orc.record_quiz_target(t0, 1.1)
    A, D, B = sh.get_kb(case.Q)
    assert np.array_equal(A, orc.A[:, :, : case.T]) and np.array_equal(D, orc.D[:, : case.T]) and np.array_equal(B, orc.B[: case.T])
    q3 = sh.start_quiz()
    orc.start_quiz(cases.WORKERS)
    assert np.array_equal(sh.get_priors(q3), orc.priors())
    # many quizzes at once: every shard sweeps the batch, the winners are merged per quiz
    quizzes = [q3] + [sh.start_quiz() for _ in range(39)]
    _, opri = orc.eval(SUBTASKS)
    assert sh.next_question_argmax_batch(quizzes) == [orc.select_argmax(opri)] * 40
    sh.start_maintenance(True)                      # maintenance edits of the dimensions are not sharded
    e = sh.add_qs_ts([interop.AddQuestionParam(1.0)], [], throw=False)
    assert e is not None and "sharded engine" in e.to_string(True)
    sh.finish_maintenance()
    sh.close()
    whole.close()


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_kb_file_crosses_between_sharded_and_whole_engines(prec, factory, tmp_path):
    """A .kb file written by a sharded engine loads into a whole-cube engine and the other way round (same byte layout: the
    file orders its rows by question, every shard streams its own block), gaps and id maps included."""
    import test_gpu_batch as tb

    case = cases.small_cases()[1]            # 37 x 5 x 101 with target and question gaps
    def make():
        if prec == "f32":
            return tb.float_engine(case, factory)[0]
        return case.make_engine(factory)
    with devices("0,0,0"):
        sh = make()
    whole = make()
    train = [(5, 1), (30, 2), (17, 0)]
    for e in (sh, whole):
        e.train([interop.AnsweredQuestion(q, a) for q, a in train], 7, 1.3)
    p_sh, p_wh = str(tmp_path / "sharded.kb"), str(tmp_path / "whole.kb")
    sh.save_kb(p_sh, False)
    whole.save_kb(p_wh, False)
    assert open(p_sh, "rb").read() == open(p_wh, "rb").read()           # byte-identical files
    from_sharded, err = factory.load_cpu_engine(p_sh)                    # sharded file -> whole engine
    assert err is None and from_sharded.get_option("shards") == -1
    with devices("0,0"):
        from_whole, err = factory.load_cpu_engine(p_wh)                  # whole file -> sharded engine (another split)
    assert err is None and from_whole.get_option("shards") == 2
    ref = whole.get_kb(case.Q)
    for e in (from_sharded, from_whole):
        for a, b in zip(ref, e.get_kb(case.Q)):
            assert np.array_equal(a, b)
        assert e.get_total_questions_asked() == whole.get_total_questions_asked()
        assert e.question_perm_from_comp([0, 5, 36]) == whole.question_perm_from_comp([0, 5, 36])
        assert e.target_perm_from_comp([0, 50, 100]) == whole.target_perm_from_comp([0, 50, 100])
        qa, qb = e.start_quiz(), whole.start_quiz()
        assert np.array_equal(e.get_priors(qa), whole.get_priors(qb))
        assert np.array_equal(e.eval_priorities(qa, case.Q), whole.eval_priorities(qb))
        whole.release_quiz(qb)
    for e in (sh, whole, from_sharded, from_whole):
        e.close()


def test_sharded_engine_exhausts_questions_like_the_whole_one(factory):
    case = cases.small_cases()[0]        # 8 questions over 3 shards (3 + 3 + 2)
    with devices("0,0,0"):
        sh = case.make_engine(factory)
    sh.set_option("select", 1)
    quiz = sh.start_quiz()
    asked = []
    for _ in range(case.Q):
        q = sh.next_question(quiz)
        assert q not in asked
        asked.append(q)
        sh.record_answer(quiz, 0)
    assert sorted(asked) == list(range(case.Q))
    with pytest.raises(interop.PqaException, match="run out of questions"):
        sh.next_question(quiz)
    sh.close()


def test_config3_shape_eight_shards_on_one_device(factory):
    """BASELINE configs[3] -- 10000Q x 5A x 10000T fp64 over 8 shards -- with all eight shards on the one device of this box:
    the one-process sharded engine's selections (argmax and the reference's sampled selector) and posteriors equal the
    whole-cube engine's over a short quiz."""
    Q, K, T = 10000, 5, 10000
    with devices("0,0,0,0,0,0,0,0"):
        sh, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    assert err is None and sh.get_option("shards") == 8
    whole, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    assert err is None
    for e in (sh, whole):
        e.set_option("workers", cases.WORKERS)
        e.fill_synthetic(8.0, 0.5, 20260928)
    qs, qw = sh.start_quiz(), whole.start_quiz()
    for step in range(3):
        a, b = sh.next_question_argmax(qs), whole.next_question_argmax(qw)
        assert a == b, (step, a, b)
        for rnd in (1, 2**63 + 12345, 2**64 - 1):
            assert sh.next_question_sampled(qs, rnd) == whole.next_question_sampled(qw, rnd)
        for e, z in ((sh, qs), (whole, qw)):
            e.set_active_question(z, (a + 1250 * step) % Q)      # a question of another shard each step
            e.record_answer(z, step % K)
        assert np.array_equal(sh.get_priors(qs), whole.get_priors(qw))
    sh.close()
    whole.close()


def test_config4_shape_sharded_batch_on_one_device(factory):
    """BASELINE configs[4]'s shape -- rows of 100000 targets, fp32, 256 quizzes batched, question axis over 8 shards -- scaled
    to 8 x 250 questions on the one device of this box: the sharded engine's batched selections equal the whole-cube Float
    engine's, and a sample of priorities agrees with the fp64 oracle on the rounded rows within the fp32 tolerance."""
    import orclib
    import test_gpu_batch as tb
    from probqa_amd import synth

    Q, K, T, B = 2000, 5, 100000, 256
    kw = dict(init_amount=0.1, prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24)
    with devices("0,0,0,0,0,0,0,0"):
        sh, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, **kw))
    assert err is None and sh.get_option("shards") == 8 and sh.get_option("precision") == 1
    whole, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, **kw))
    assert err is None
    quizzes = []
    for e in (sh, whole):
        e.set_option("workers", cases.WORKERS)
        e.fill_synthetic(8.0, 0.5, 20260928)
        qz = [e.start_quiz() for _ in range(B)]
        for i, z in enumerate(qz):
            if i % 3:
                e.set_active_question(z, (37 * i) % Q)
                e.record_answer(z, i % K)
        quizzes.append(qz)
    picks_s, picks_w = sh.next_question_argmax_batch(quizzes[0]), whole.next_question_argmax_batch(quizzes[1])
    assert picks_s == picks_w
    assert all(0 <= p < Q for p in picks_s)
    # a sample against the oracle: 6 questions' rows rebuilt on the host (rounded to fp32), quiz 1's posterior
    sample = [0, 249, 250, 1007, 1750, 1999]
    orc = orclib.Oracle(K, len(sample), T, 0.1)
    for j, q in enumerate(sample):
        Aq, Dq, Bq = synth.synthetic_kb(K, 1, T, 0.1, 8.0, 0.5, 20260928, q_offset=q, q_total=Q)
        orc.A[j, :, :T], orc.D[j, :T] = Aq[0].astype(np.float32), Dq[0].astype(np.float32)
    orc.B[:T] = Bq.astype(np.float32)
    orc.mants[:T] = sh.get_priors(quizzes[0][1])
    _, opri = orc.eval_avx2(8)
    pri = sh.eval_priorities_batch(quizzes[0][:64], Q)[1][sample]
    keep = pri != 0                                       # (quiz 1 has asked question 37)
    rel = np.abs(pri[keep] - opri[keep]) / opri[keep]
    print("config-4 shape, sharded batch: max rel err vs fp64 oracle on the sample %.2e" % rel.max())
    assert rel.max() < 5e-4
    sh.close()
    whole.close()
