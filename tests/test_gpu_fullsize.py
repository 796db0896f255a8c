"""BASELINE-size cubes on the device (filled there: probqa_amd/synth.py's generator, bit-identical on both sides), checked
against the oracle on SAMPLES of their questions -- the oracle gets the sampled questions' rows from the same generator and the
engine's own posterior -- and through properties that need no oracle: two sweeps of different construction agree on EVERY question,
quizzes in the same state get the same bits, asked questions get 0, the pick is the argmax of the priorities."""
import collections

import numpy as np
import pytest

import cases
import orclib
import test_gpu_batch as tb
from probqa_amd import interop, synth

pytestmark = pytest.mark.gpu

SEED = 20260928
Shape = collections.namedtuple("Shape", "T tgaps")


def make_engine(factory, K, Q, T, f32):
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
    assert err is None and eng is not None, err
    eng.set_option("workers", cases.WORKERS)
    eng.fill_synthetic(8.0, 0.5, SEED)
    return eng


def sample_oracle(K, Q, T, q0, n, f32, posterior):
    """(priorities, fp32 tolerance) of questions q0 .. q0 + n - 1 of the synthetic Q x K x T cube for a quiz with this posterior."""
    A, D, B = synth.synthetic_kb(K, n, T, 0.1, 8.0, 0.5, SEED, q_offset=q0, q_total=Q)
    if f32:
        A, D, B = (x.astype(np.float32).astype(np.float64) for x in (A, D, B))
    orc = orclib.Oracle(K, n, T, 0.1)
    orc.set_kb(A, D, B)
    orc.mants[:T] = posterior
    _, pri = orc.eval(1)
    return pri, (tb.f32_tolerance(orc, Shape(T, [])) if f32 else None)


@pytest.mark.parametrize("prec,Q,T", [("f32", 700, 100000), ("f64", 400, 50000)], ids=["f32_700x5x100000", "f64_400x5x50000"])
def test_long_rows_many_questions_per_cluster(prec, Q, T, factory):
    """Rows of BASELINE configs[4]'s length through the cluster sweep (cluster_kernels.hip) with every cluster sweeping dozens
    of questions one after the other (the exchange buffers alternate, the members take turns folding): all questions against the
    streaming form -- a kernel that shares nothing with it but the element arithmetic -- and samples against the oracle."""
    K, f32 = 5, prec == "f32"
    eng = make_engine(factory, K, Q, T, f32)
    name = eng.eval_kernel_name()
    assert name.startswith(prec + "_cluster"), name
    n_clusters = int(name.split("_x")[1].split("_")[0])
    assert Q >= 20 * n_clusters, (name, "every cluster must sweep many questions")
    quiz = eng.start_quiz()
    for step, (q, a) in enumerate([(None, None), (Q // 3, 1), (Q - 1, 4)]):
        if q is not None:
            eng.set_active_question(quiz, q)
            eng.record_answer(quiz, a)
        pri = eng.eval_priorities(quiz)
        eng.set_option("eval_variant", 99)
        stream = eng.eval_priorities(quiz)
        eng.set_option("eval_variant", 0)
        asked = [Q // 3, Q - 1][:step]
        assert all(pri[i] == 0 and stream[i] == 0 for i in asked) and (np.delete(pri, asked) > 0).all()
        rel = np.abs(pri - stream) / np.maximum(stream, 1e-300)
        assert rel.max() < (2e-4 if f32 else 1e-9), (name, step, float(rel.max()), int(rel.argmax()))
        assert eng.next_question_argmax(quiz) == int(np.argmax(pri))
        posterior = eng.get_priors(quiz)
        for q0 in (0, Q // 2 - 2, Q - 4):
            opri, tol = sample_oracle(K, Q, T, q0, 4, f32, posterior)
            for i in range(4):
                if q0 + i in asked:
                    continue
                r = abs(pri[q0 + i] - opri[i]) / opri[i]
                assert r < (tol[i] if f32 else 1e-9), (name, step, q0 + i, r)
    eng.close()


def test_configs4_shard_256_quizzes(factory):
    """BASELINE configs[4], one GPU's shard: 12500 x 5 x 100000 fp32 (30 GB), 256 quizzes in one batched sweep."""
    K, Q, T, B = 5, 12500, 100000, 256
    eng = make_engine(factory, K, Q, T, True)
    quizzes = [eng.start_quiz() for _ in range(B)]
    hists = {3: [(17, 2)], 77: [(6000, 0), (12499, 4)], 255: [(1, 1)]}
    for i, hist in hists.items():
        for q, a in hist:
            eng.set_active_question(quizzes[i], q)
            eng.record_answer(quizzes[i], a)
    pri = eng.eval_priorities_batch(quizzes, Q)
    picks = eng.next_question_argmax_batch(quizzes)
    fresh = [i for i in range(B) if i not in hists]
    for i in fresh[1:]:
        assert np.array_equal(pri[i], pri[fresh[0]])          # the same state: the same bits, whichever lane and wave
    for i in range(B):
        assert picks[i] == int(np.argmax(pri[i]))
        asked = [q for q, _ in hists.get(i, [])]
        assert all(pri[i][q] == 0 for q in asked) and (np.delete(pri[i], asked) > 0).all()
    worst = 0.0
    for i in (0, 77, 255):
        posterior = eng.get_priors(quizzes[i])
        for q0 in (0, 6001, Q - 3):
            opri, tol = sample_oracle(K, Q, T, q0, 3, True, posterior)
            keep = np.array([q0 + j not in [q for q, _ in hists.get(i, [])] for j in range(3)])   # (the sample oracle has no asked bits)
            r = np.abs(pri[i][q0:q0 + 3] - opri) / opri
            worst = max(worst, float((r / tol)[keep].max()))
            assert (r < tol)[keep].all(), (i, q0, r, tol)
    print("configs[4] shard, 256 quizzes: sampled priorities at most %.2f of the fp32 tolerance" % worst)
    eng.close()
