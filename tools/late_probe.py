"""Where a late-quiz-state case (tests/test_gpu_late.py) deviates from the oracle: per step the worst questions, and for each of
them what its answer rows look like (numpy fp64): the largest posterior element, how far the answer moves it, its share of the
velocity sum.  usage: python tools/late_probe.py CASE [CASE ...] [name=value ...]   (engine options after the cases)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import test_gpu_late as tl  # noqa: E402
from probqa_amd import interop  # noqa: E402


def rows_of(case, q, prior):
    A, D, _ = case.kb()
    pr = prior.copy()
    pr[list(case.tgaps)] = 0.0
    lh = (A[q] / D[q][None, :]) * pr[None, :]
    W = lh.sum(axis=1)
    out = []
    for k in range(case.K):
        p = lh[k] / W[k]
        t = int(p.argmax())
        d = p - pr
        V = float((d * d).sum())
        out.append("k=%d W=%.3e pmax=1-%.3e at %d  d_h=%.3e  d_h^2/V=%.4f  V=%.3e" % (k, W[k], 1 - p[t], t, d[t], d[t] ** 2 / V if V else 0, V))
    return out


def main():
    idx = [int(a) for a in sys.argv[1:] if "=" not in a]
    opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
    factory = interop.PqaEngineFactory()
    for i in idx:
        leg, case, options = tl.late_case(i)
        print("====", case.name, options + opts, "answers", case.answers)
        orc, eng = case.make_oracle(), case.make_engine(factory)
        for n, v in options + opts:
            eng.set_option(n, v)
        quiz = eng.start_quiz()
        orc.start_quiz(cases.WORKERS)
        for step in range(len(case.answers) + 1):
            pri = eng.eval_priorities(quiz)
            _, opri = orc.eval(128)
            rel = cases.rel_err(pri, opri)
            rel[opri == 0] = 0
            prior = orc.priors()
            print("step %d: worst %.3e  1-pmax(prior)=%.3e" % (step, rel.max(), 1 - prior.max()))
            if rel.max() > 3e-10:
                for q in np.argsort(rel)[::-1][:3]:
                    print("  question %d rel %.3e" % (q, rel[q]))
                    for line in rows_of(case, int(q), prior):
                        print("     ", line)
            if step < len(case.answers):
                q, a = case.answers[step]
                eng.set_active_question(quiz, q)
                eng.record_answer(quiz, a)
                orc.record_answer(q, a, cases.WORKERS - 1)
        eng.close()


if __name__ == "__main__":
    main()
