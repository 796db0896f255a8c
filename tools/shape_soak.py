"""Soak of the single-quiz sweeps' shapes: random (K, Q, T) -- T anywhere up to 60000, i.e. through every register shape of both
precisions, the cluster form and odd row lengths -- with gaps and a few answers; the engine's default kernel against its streaming
form (variant 99) on ALL questions.  shape_soak.py first last [f32]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
factory = interop.PqaEngineFactory()
bad, t0, names = 0, time.time(), {}
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 9))
    T = int(rng.choice([rng.integers(2, 1100), rng.integers(1000, 17000), rng.integers(16000, 60000)]))
    Q = int(rng.integers(1, 40)) if T < 20000 else int(rng.integers(1, 12))
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    e, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
    e.fill_synthetic(float(rng.choice([2.0, 8.0])), 0.5, 700 + seed); e.set_option("workers", 16); e.set_option("select", 1)
    if T > 8: e.set_target_gaps(sorted(set(rng.choice(T, int(rng.integers(0, min(T // 4, 20))), replace=False).tolist())))
    if Q > 3: e.set_question_gaps(sorted(set(rng.choice(Q, int(rng.integers(0, Q // 3)), replace=False).tolist())))
    if rng.integers(0, 3) == 0: e.set_option("eval_max_grid", int(rng.integers(1, 4)))
    name = e.eval_kernel_name(); names[name.split("_x")[0]] = names.get(name.split("_x")[0], 0) + 1
    try:
        quiz = e.start_quiz()
        for step in range(3):
            pri = e.eval_priorities(quiz)
            e.set_option("eval_variant", 99); ref = e.eval_priorities(quiz); e.set_option("eval_variant", 0)
            assert ((pri == 0) == (ref == 0)).all(), (seed, name, step, "zeros")
            rel = np.where(ref != 0, np.abs(pri - ref) / np.where(ref != 0, np.abs(ref), 1), 0)
            assert rel.max() < (2e-3 if f32 else 1e-9), (seed, name, (K, Q, T), step, float(rel.max()), int(rel.argmax()))
            if (pri > 0).any():
                q = e.next_question(quiz)
                assert pri[q] >= pri.max() * (1 - (1e-3 if f32 else 1e-9)), (seed, name, step, "argmax")
                e.record_answer(quiz, int(rng.integers(0, K)))
    except interop.PqaException as ex:
        if "run out of questions" not in str(ex):
            bad += 1; print("FAIL seed", seed, name, (K, Q, T), repr(ex)[:200])
    except AssertionError as ex:
        bad += 1; print("FAIL", ex)
    e.close()
print("seeds %d..%d%s: %d failures, %.0f s; kernels: %s" % (first, last, " f32" if f32 else "", bad, time.time() - t0, sorted(names.items())))
