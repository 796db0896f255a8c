"""Model-based stress test of the engine's stream-ordered paths.  A random interleaving of every quiz-level call over
several quizzes at once -- StartQuiz / ResumeQuiz / NextQuestion (launch, graph replay, batch) / RecordAnswer /
ListTopTargets (cache hit and miss) / GetPriors / ReleaseQuiz (buffer reuse) / training -- is replayed against one CPU
oracle per quiz.  Posteriors must stay bit-identical and every selection / listing must equal the oracle's, whatever was
enqueued before it."""
import numpy as np
import pytest

import cases
import orclib
from probqa_amd import interop, synth

pytestmark = pytest.mark.gpu

K, Q, T = 4, 48, 300


class Shadow:
    """One oracle (its own copy of the KB) per live quiz."""

    def __init__(self, kb_arrays, asked=()):
        self.orc = orclib.Oracle(K, Q, T, 0.1)
        self.orc.set_kb(*kb_arrays)
        if asked:
            assert self.orc.resume_quiz(list(asked), cases.WORKERS, True) == 0
        else:
            self.orc.start_quiz(cases.WORKERS)
        self.asked = {q for q, _ in asked}

    def argmax(self):
        _, pri = self.orc.eval(8 * cases.WORKERS)
        return self.orc.select_argmax(pri)

    def top(self, n):
        p = self.orc.priors()
        order = sorted(range(T), key=lambda t: (-p[t], t))[:n]
        return [(t, p[t]) for t in order]


@pytest.mark.parametrize("seed,mode", [(1, "plain"), (2, "plain"), (3, "plain"), (4, "resident"), (5, "resident"),
                                       (6, "row_sharing_batches"), (7, "three_shards"), (8, "three_shards"),
                                       (20, "resident"), (30, "resident"), (51, "resident")],   # (found by tools/stress_more.py)
                         ids=["seed1", "seed2", "seed3", "seed4_resident_sweep", "seed5_resident_sweep", "seed6_row_sharing_batches",
                              "seed7_three_shards", "seed8_three_shards", "seed20_resident_sweep", "seed30_resident_sweep",
                              "seed51_resident_sweep"])
def test_random_interleaving_against_per_quiz_oracles(factory, seed, mode):
    import os

    rng = np.random.default_rng(seed)
    kb = list(synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, 100 + seed))
    saved = os.environ.get("PQA_DEVICES")
    if mode == "three_shards":     # the same calls through the one-process sharded engine (three shards on this box's one device)
        os.environ["PQA_DEVICES"] = "0,0,0"
    try:
        eng, err = factory.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    finally:
        if saved is None:
            os.environ.pop("PQA_DEVICES", None)
        else:
            os.environ["PQA_DEVICES"] = saved
    assert err is None
    assert eng.get_option("shards") == (3 if mode == "three_shards" else -1)
    eng.set_kb(*kb)
    eng.set_option("workers", cases.WORKERS)
    eng.set_option("select", 1)
    resident = mode == "resident"
    if mode == "row_sharing_batches":
        eng.set_option("batch_min", 1)   # every batch through the lane-per-quiz sweep
    if resident:
        # plain selections through the resident kernel (stopped and restarted by the launches, the training and the idle
        # time in between), five workgroups streaming the questions
        eng.set_option("server", 1)
        eng.set_option("server_idle_us", 300)
        eng.set_option("eval_max_grid", 5)
    live = {}          # quiz id -> Shadow
    pending = {}       # quiz id -> question handed out and not yet answered
    for step in range(260):
        op = rng.choice(["start", "resume", "next", "next_graph", "batch", "answer", "top", "priors", "release", "train"],
                        p=[0.08, 0.04, 0.2, 0.08, 0.08, 0.22, 0.14, 0.06, 0.05, 0.05])
        ids = sorted(live)
        if op == "start" and len(live) < 6:
            live[eng.start_quiz()] = Shadow(kb)
        elif op == "resume" and len(live) < 6:
            qs = rng.choice(Q, int(rng.integers(1, 4)), replace=False).tolist()
            aqs = [(int(q), int(rng.integers(0, K))) for q in qs]
            live[eng.resume_quiz([interop.AnsweredQuestion(q, a) for q, a in aqs])] = Shadow(kb, aqs)
        elif op in ("next", "next_graph") and ids:
            quiz = int(rng.choice(ids))
            want = live[quiz].argmax()
            if want < 0:
                continue
            eng.set_option("use_graph", 1 if op == "next_graph" else 0)
            assert eng.next_question(quiz) == want, (step, op, quiz)
            eng.set_option("use_graph", 0)
            pending[quiz] = want
        elif op == "batch" and len(ids) >= 2:
            sub = [int(q) for q in rng.choice(ids, int(rng.integers(2, len(ids) + 1)), replace=False)]
            want = [live[q].argmax() for q in sub]
            assert eng.next_question_argmax_batch(sub) == want, (step, sub)
            for q, w in zip(sub, want):
                if w >= 0:
                    pending[q] = w
        elif op == "answer" and pending:
            quiz = int(rng.choice(sorted(pending)))
            q, a = pending.pop(quiz), int(rng.integers(0, K))
            eng.record_answer(quiz, a)
            live[quiz].orc.record_answer(q, a, cases.WORKERS - 1)
        elif op == "top" and ids:
            quiz, n = int(rng.choice(ids)), int(rng.choice([1, 3, 10, 25]))
            got = [(r.i_target, r.prob) for r in eng.list_top_targets(quiz, n)]
            assert got == live[quiz].top(n), (step, quiz, n)
        elif op == "priors" and ids:
            quiz = int(rng.choice(ids))
            assert np.array_equal(eng.get_priors(quiz), live[quiz].orc.priors()), (step, quiz)
        elif op == "release" and ids:
            quiz = int(rng.choice(ids))
            eng.release_quiz(quiz)
            live.pop(quiz).orc.close()
            pending.pop(quiz, None)
        elif op == "train" and ids:
            # RecordQuizTarget changes the KB under every quiz: mirror it into the shared arrays and every oracle
            quiz = int(rng.choice(ids))
            aqs = list(live[quiz].orc.answers)
            if not aqs:
                continue
            target = int(rng.integers(0, T))
            eng.record_quiz_target(quiz, target, 1.0)
            for sh in live.values():
                sh.orc.train(aqs, target, 1.0)
            A, D, B = eng.get_kb()
            assert np.array_equal(A, next(iter(live.values())).orc.A[:, :, :T])
            kb[0], kb[1], kb[2] = A, D, B
    for sh in live.values():
        sh.orc.close()
    eng.close()
