"""Soak of the maintenance-mode edits (AddQsTs / RemoveQuestions / RemoveTargets / Compact, reference PqaCore/CpuEngine.cpp:468-658,
GapTracker.h, PermanentIdManager.cpp) against a numpy model: random sequences, the cube / vB of the live questions and targets, the
dimensions and the permanent <-> compact id maps compared after every edit; then a quiz on the edited KB against a KB created with
those numbers.  kb_soak.py first_seed last_seed [f32]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
factory = interop.PqaEngineFactory()
r = (lambda x: float(np.float32(x))) if f32 else (lambda x: x)
def make(K, Q, T):
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
    e, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
    assert err is None, err
    e.set_option("workers", 16)
    return e
bad, t0, edits = 0, time.time(), 0
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    K, Q, T = int(rng.integers(2, 6)), int(rng.integers(3, 30)), int(rng.integers(3, 70))
    eng = make(K, Q, T)
    A, D, B = synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, 900 + seed)
    eng.set_kb(A, D, B)
    if f32: A, D, B = (x.astype(np.float32).astype(np.float64) for x in (A, D, B))
    qgaps, tgaps = [], []                      # LIFO lists, like GapTracker
    fresh = set()      # questions (re)initialised by AddQsTs: every answer equally likely given every target -- the posterior equals the
                       # prior, the velocity term is 0 or 1e-34 by the rounding of the priors' sum, and the priority follows that coin
    permQ, permT = list(range(Q)), list(range(T))   # compact -> permanent (-1 at gaps)
    nextPQ, nextPT = Q, T
    def quiz_check(where):
        # a quiz on the edited KB (gaps and all) = a quiz on a KB created with the live numbers
        liveQ = [q for q in range(Q) if q not in qgaps]; liveT = [t for t in range(T) if t not in tgaps]
        if len(liveQ) >= 1 and len(liveT) >= 2:
            ref = make(K, len(liveQ), len(liveT))
            ref.set_kb(np.ascontiguousarray(A[liveQ][:, :, liveT]), np.ascontiguousarray(D[liveQ][:, liveT]), np.ascontiguousarray(B[liveT]))
            q1, q2 = eng.start_quiz(), ref.start_quiz()
            p1, p2 = eng.get_priors(q1)[liveT], ref.get_priors(q2)
            assert np.allclose(p1, p2, rtol=1e-12, atol=0), (seed, where, "priors of the edited KB")
            e1, e2 = eng.eval_priorities(q1)[liveQ], ref.eval_priorities(q2)
            rel = np.abs(e1 - e2) / np.maximum(np.abs(e2), 1e-300)
            rel[[i for i, q in enumerate(liveQ) if q in fresh]] = 0
            assert rel.max() < (2e-3 if f32 else 1e-9), (seed, where, "priorities of the edited KB", float(rel.max()), (Q, T), sorted(qgaps), sorted(tgaps))
            eng.release_quiz(q1)
            ref.close()
    try:
        eng.start_maintenance(False)
        for step in range(int(rng.integers(3, 14))):
            op = rng.choice(["rq", "rt", "add", "compact"], p=[0.3, 0.3, 0.3, 0.1])
            liveQ = [q for q in range(Q) if q not in qgaps]; liveT = [t for t in range(T) if t not in tgaps]
            if op == "rq" and len(liveQ) > 2:
                ids = rng.choice(liveQ, int(rng.integers(1, min(4, len(liveQ) - 1))), replace=False).tolist()
                eng.remove_questions(ids)
                if os.environ.get("KB_SOAK_EACH"): print(seed, step, "remove_questions", ids)
                for q in ids: qgaps.append(q); permQ[q] = -1
            elif op == "rt" and len(liveT) > 2:
                ids = rng.choice(liveT, int(rng.integers(1, min(4, len(liveT) - 1))), replace=False).tolist()
                eng.remove_targets(ids)
                if os.environ.get("KB_SOAK_EACH"): print(seed, step, "remove_targets", ids)
                for t in ids: tgaps.append(t); permT[t] = -1
            elif op == "add":
                nq, nt = int(rng.integers(0, 4)), int(rng.integers(0, 4))
                aq = [interop.AddQuestionParam(float(rng.choice([0.25, 0.5, 1.0, 2.0]))) for _ in range(nq)]
                at = [interop.AddTargetParam(float(rng.choice([0.3, 0.7, 1.5]))) for _ in range(nt)]
                eng.add_qs_ts(aq, at)
                if os.environ.get("KB_SOAK_EACH"): print(seed, step, "add", [p.init_amount for p in aq], [p.init_amount for p in at], "->", [p.i_question for p in aq], [p.i_target for p in at])
                qids, tids = [], []
                for _ in range(nq):
                    if qgaps: qids.append(qgaps.pop())
                    else: qids.append(Q); Q += 1
                for _ in range(nt):
                    if tgaps: tids.append(tgaps.pop())
                    else: tids.append(T); T += 1
                assert [p.i_question for p in aq] == qids and [p.i_target for p in at] == tids, (seed, step, "ids", qids, tids)
                if Q > A.shape[0]:
                    A = np.concatenate([A, np.zeros((Q - A.shape[0], K, A.shape[2]))], axis=0)
                    D = np.concatenate([D, np.zeros((Q - D.shape[0], D.shape[1]))], axis=0)
                if T > A.shape[2]:
                    A = np.concatenate([A, np.zeros((A.shape[0], K, T - A.shape[2]))], axis=2)
                    D = np.concatenate([D, np.zeros((D.shape[0], T - D.shape[1]))], axis=1)
                    B = np.concatenate([B, np.zeros(T - B.shape[0])])
                permQ += [-1] * (Q - len(permQ)); permT += [-1] * (T - len(permT))
                for t, p in zip(tids, at):          # target columns first (over the questions not re-initialised just now) ...
                    A[:, :, t], D[:, t], B[t] = r(p.init_amount ** 2), r(p.init_amount ** 2 * K), r(p.init_amount)
                    permT[t] = nextPT; nextPT += 1
                for q, p in zip(qids, aq):          # ... then whole questions, every column
                    A[q], D[q] = r(p.init_amount ** 2), r(p.init_amount ** 2 * K)
                    permQ[q] = nextPQ; nextPQ += 1
                    fresh.add(q)
            elif op == "compact":
                old_q, old_t = eng.compact()
                if os.environ.get("KB_SOAK_EACH"): print(seed, step, "compact")
                keepQ = [q for q in range(Q) if q not in qgaps]; keepT = [t for t in range(T) if t not in tgaps]
                assert sorted(old_q) == keepQ and sorted(old_t) == keepT, (seed, step, "compact maps")
                A, D, B = A[old_q][:, :, old_t], D[old_q][:, old_t], B[old_t]
                permQ, permT = [permQ[q] for q in old_q], [permT[t] for t in old_t]
                fresh = {i for i, q in enumerate(old_q) if q in fresh}
                Q, T, qgaps, tgaps = len(old_q), len(old_t), [], []
            else:
                continue
            edits += 1
            d = eng.copy_dims()
            assert (d.n_questions, d.n_targets) == (Q, T), (seed, step, op, "dims")
            liveQ = [q for q in range(Q) if q not in qgaps]; liveT = [t for t in range(T) if t not in tgaps]
            A2, D2, B2 = eng.get_kb()
            assert np.array_equal(A2[liveQ][:, :, liveT], A[liveQ][:, :, liveT]) and np.array_equal(D2[liveQ][:, liveT], D[liveQ][:, liveT]) \
                and np.array_equal(B2[liveT], B[liveT]), (seed, step, op, "cube")
            assert eng.question_perm_from_comp(list(range(Q))) == permQ, (seed, step, op, "question ids", eng.question_perm_from_comp(list(range(Q))), permQ)
            assert eng.target_perm_from_comp(list(range(T))) == permT, (seed, step, op, "target ids")
            lp = [p for p in permQ if p >= 0]
            assert eng.question_comp_from_perm(lp) == [permQ.index(p) for p in lp], (seed, step, op, "inverse map")
            if os.environ.get("KB_SOAK_EACH"):
                eng.finish_maintenance()
                quiz_check("step %d after %s" % (step, op))
                eng.start_maintenance(False)
        eng.finish_maintenance()
        quiz_check("end")
    except BaseException as ex:  # noqa: BLE001
        bad += 1
        print("FAIL seed", seed, (K, Q, T), repr(ex)[:400])
    eng.close()
print("seeds %d..%d%s: %d failures, %d edits checked, %.0f s" % (first, last, " f32" if f32 else "", bad, edits, time.time() - t0))
