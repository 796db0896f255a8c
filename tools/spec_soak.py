"""Soak of the speculative sweep: the same random interleaving of quiz calls (several quizzes, both selectors, training, gaps of
questions, option changes) on an engine that speculates and on one that does not -- every returned question, listing and posterior
must be the same.  spec_soak.py first_seed last_seed [f32]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
factory = interop.PqaEngineFactory()
bad, t0, hits = 0, time.time(), 0
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    K, Q = int(rng.integers(2, 7)), int(rng.integers(8, 80))
    T = int(rng.choice([50, 300, 1000, 1500, 5000, 17000]))
    kb = synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, 500 + seed)
    engs = []
    for spec in (1, 0):
        kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
        e, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
        e.set_kb(*kb); e.set_option("workers", 16); e.set_option("speculate", spec); e.set_option("seed", 9)
        engs.append(e)
    a, b = engs
    select = int(rng.integers(0, 2))
    for e in engs: e.set_option("select", select)
    live, pending = [], {}
    try:
        for step in range(200):
            op = rng.choice(["start", "next", "next_rnd", "answer", "top", "priors", "release", "train", "gap", "select", "own"],
                            p=[0.08, 0.25, 0.1, 0.25, 0.1, 0.04, 0.04, 0.04, 0.02, 0.03, 0.05])
            def both(fn):
                out = []
                for e in engs:
                    try:
                        out.append(("ok", fn(e)))
                    except interop.PqaException as ex:
                        out.append(("raised", str(ex)[:60]))
                assert out[0] == out[1], (seed, step, op, out)
                if out[0][0] == "raised":
                    raise interop.PqaException(out[0][1])
                return out[0][1]
            if op == "start" and len(live) < 4:
                live.append(both(lambda e: e.start_quiz()))
            elif op in ("next", "next_rnd") and live:
                quiz = int(rng.choice(live))
                try:
                    if op == "next":
                        got = both(lambda e: e.next_question(quiz))
                    else:
                        rnd = int(rng.integers(0, 2**63)) * 2
                        got = both(lambda e: e.next_question_sampled(quiz, rnd))
                    pending[quiz] = got
                except interop.PqaException:
                    pass    # (questions exhausted: both raise -- `both` compares only returns)
            elif op == "answer" and pending:
                quiz = int(rng.choice(sorted(pending))); pending.pop(quiz)
                ans = int(rng.integers(0, K))
                try:
                    both(lambda e: e.record_answer(quiz, ans))
                except interop.PqaException:
                    pass    # (the question became a gap meanwhile: both refuse)
            elif op == "own" and live:      # the client picks the question itself
                quiz, q = int(rng.choice(live)), int(rng.integers(0, Q))
                ans = int(rng.integers(0, K))
                for e in engs:
                    e.set_active_question(quiz, q, throw=False)
                ra, rb = [e.record_answer(quiz, ans, throw=False) for e in engs]
                assert (ra is None) == (rb is None), (seed, step, op)
                pending.pop(quiz, None)
            elif op == "top" and live:
                quiz, n = int(rng.choice(live)), int(rng.choice([1, 3, 10]))
                both(lambda e: [(r.i_target, r.prob) for r in e.list_top_targets(quiz, n)])
            elif op == "priors" and live:
                quiz = int(rng.choice(live))
                assert np.array_equal(a.get_priors(quiz), b.get_priors(quiz)), (seed, step, op)
            elif op == "release" and live:
                quiz = int(rng.choice(live)); live.remove(quiz); pending.pop(quiz, None)
                both(lambda e: e.release_quiz(quiz))
            elif op == "train" and live:
                quiz, t = int(rng.choice(live)), int(rng.integers(0, T))
                try:
                    both(lambda e: e.record_quiz_target(quiz, t, 1.0))
                except interop.PqaException:
                    pass    # (an answered question became a gap meanwhile: both refuse)
            elif op == "gap":
                q = int(rng.integers(0, Q))
                for e in engs: e.set_question_gaps([q])
            elif op == "select":
                select ^= 1
                for e in engs: e.set_option("select", select)
        hits += a.get_option("spec_hits")
    except BaseException as ex:  # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, (K, Q, T), repr(ex)[:300])
    for e in engs: e.close()
print("seeds %d..%d%s: %d failures, %d speculative sweeps used, %.0f s" % (first, last, " f32" if f32 else "", bad, hits, time.time() - t0))
