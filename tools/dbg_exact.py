import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import cases, test_gpu_fuzz as tf
from probqa_amd import interop
factory = interop.PqaEngineFactory()
i, stop = int(sys.argv[1]), int(sys.argv[2])
case = tf.random_case(i)
orc = case.make_oracle(); eng = case.make_engine(factory)
quiz = eng.start_quiz(); orc.start_quiz(cases.WORKERS)
for step in range(stop + 1):
    if step == stop:
        pri = eng.eval_priorities(quiz); run, opri = orc.eval(8 * cases.WORKERS)
        rel = cases.rel_err(pri, opri); rel[opri == 0] = 0
        qbad = int(np.argmax(rel))
        print("step", step, "worst question", qbad, "rel", rel[qbad], "n above 1e-11:", int((rel > 1e-11).sum()), "of", len(rel))
        A, D, _ = case.kb()
        pr = np.array(orc.priors()); 
        valid = np.ones(case.T, bool); valid[list(case.tgaps)] = False
        lh = (A[qbad] / D[qbad][None, :]) * np.where(valid, pr, 0)[None, :]
        W = lh.sum(axis=1, keepdims=True)
        p = lh / W
        print("priors", pr)
        for k in range(case.K):
            print(" k", k, "W", W[k, 0], "1-pmax %.3g" % (1 - p[k].max()), "p", p[k])
    q, a = case.answers[step] if step < len(case.answers) else (None, None)
    if q is None: break
    eng.set_active_question(quiz, q); eng.record_answer(quiz, a); orc.record_answer(q, a, cases.WORKERS - 1)
