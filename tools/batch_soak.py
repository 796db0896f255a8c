"""Soak of the row-sharing batched sweep: random (K, Q, T, number of quizzes, tile, questions per block), quizzes in different
states; every quiz's batched priorities against the single-quiz sweep's (another kernel), the batched picks against their argmax.
batch_soak.py first last [f32]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop
first, last = int(sys.argv[1]), int(sys.argv[2])
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
factory = interop.PqaEngineFactory()
bad, t0 = 0, time.time()
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 10)); Q = int(rng.integers(1, 50))
    T = int(rng.choice([rng.integers(2, 300), rng.integers(250, 1100), rng.integers(1000, 5000)]))
    B = int(rng.choice([1, 2, 63, 64, 65, 128, 200, 256, int(rng.integers(1, 257))]))
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    e, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
    e.fill_synthetic(8.0, 0.5, 300 + seed); e.set_option("workers", 16); e.set_option("select", 1)
    e.set_option("batch_min", 1); e.set_option("batch_tile", int(rng.choice([0, 64, 128, 192, 512]))); e.set_option("batch_qb", int(rng.choice([0, 1, 2, 4])))
    if T > 8: e.set_target_gaps(sorted(set(rng.choice(T, int(rng.integers(0, min(T // 4, 12))), replace=False).tolist())))
    if Q > 3: e.set_question_gaps(sorted(set(rng.choice(Q, int(rng.integers(0, Q // 3)), replace=False).tolist())))
    try:
        quizzes = []
        for i in range(B):
            z = e.start_quiz()
            for _ in range(int(rng.integers(0, 3)) if Q > 3 else 0):
                try:
                    e.next_question(z); e.record_answer(z, int(rng.integers(0, K)))
                except interop.PqaException:
                    break
            quizzes.append(z)
        pri = e.eval_priorities_batch(quizzes, Q)
        picks = e.next_question_argmax_batch(quizzes)
        for i in sorted(set([0, B - 1] + rng.choice(B, min(B, 6), replace=False).tolist())):
            one = e.eval_priorities(quizzes[i])
            assert ((one == 0) == (pri[i] == 0)).all(), (seed, i, "zeros")
            rel = np.where(one != 0, np.abs(pri[i] - one) / np.where(one != 0, np.abs(one), 1), 0)
            assert rel.max() < (3e-3 if f32 else 1e-9), (seed, (K, Q, T, B), i, float(rel.max()))
            if (pri[i] > 0).any():
                assert pri[i][picks[i]] == pri[i].max(), (seed, i, "pick")
            else:
                assert picks[i] == -1
    except AssertionError as ex:
        bad += 1; print("FAIL", ex)
    e.close()
print("seeds %d..%d%s: %d failures, %.0f s" % (first, last, " f32" if f32 else "", bad, time.time() - t0))
