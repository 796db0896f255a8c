"""Conditioning of a fuzz case's failing step: how close to 1 the largest posterior element of the worst question is."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import cases, test_gpu_fuzz as tf
from probqa_amd import interop
i = int(sys.argv[1])
case = tf.random_case(i)
factory = interop.PqaEngineFactory()
orc, eng = case.make_oracle(), case.make_engine(factory)
quiz = eng.start_quiz(); orc.start_quiz(cases.WORKERS)
for step in range(len(case.answers) + 1):
    pri = eng.eval_priorities(quiz); _, opri = orc.eval(8 * cases.WORKERS)
    rel = np.where(opri != 0, np.abs(pri - opri) / np.where(opri != 0, opri, 1), 0)
    q = int(rel.argmax())
    T = case.T
    prior = orc.priors().copy()
    for t in case.tgaps: prior[t] = 0
    like = orc.A[q, :, :T] / orc.D[q, :T][None, :] * prior[None, :]
    post = like / like.sum(axis=1, keepdims=True)
    l2 = np.log2(np.where(post > 0, post, 1))
    nz = l2[l2 < 0]
    print("%s step %d: worst question %d rel %.3g; its smallest |log2 p| = %.3g (p = 1 - %.3g); 1e-16 / that = %.3g"
          % (case.name, step, q, rel.max(), np.abs(nz).min() if nz.size else 0, 1 - post.max(), 1e-16 / max(np.abs(nz).min(), 1e-300) if nz.size else 0))
    if step < len(case.answers):
        qq, a = case.answers[step]
        eng.set_active_question(quiz, qq); eng.record_answer(quiz, a); orc.record_answer(qq, a, cases.WORKERS - 1)
