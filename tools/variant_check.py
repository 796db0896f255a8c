"""Parity + timing of one sweep shape (tools): variant_check.py <variant> [allk_per_cu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import cases
from probqa_amd import interop
variant = int(sys.argv[1])
f = interop.PqaEngineFactory()
worst = 0.0
for case in cases.small_cases() + [cases.Case("k5_1000", 5, 97, 1000, seed=3, tgaps=[5, 77, 500], qgaps=[4], answers=[(3, 1), (50, 4)])]:
    if case.K != 5 or case.T > 1024:
        continue
    eng, orc = case.make_engine(f), case.make_oracle()
    for mg in (0, 3):
        eng.set_option("eval_max_grid", mg)
        quiz = eng.start_quiz()
        orc.start_quiz(cases.WORKERS)
        for step in range(len(case.answers) + 1):
            eng.set_option("eval_variant", variant)
            pri = eng.eval_priorities(quiz)
            sel = eng.next_question_argmax(quiz)
            samp = eng.next_question_sampled(quiz, 0x9E3779B97F4A7C15)
            eng.set_option("eval_variant", 0)
            assert np.array_equal(eng.eval_priorities(quiz) == 0, pri == 0)
            run, opri = orc.eval(128)
            nz = opri != 0
            worst = max(worst, float(np.max(np.abs(pri[nz] - opri[nz]) / opri[nz])) if nz.any() else 0)
            assert sel == orc.select_argmax(opri) and samp == orc.select_sampled(run, 128, 0x9E3779B97F4A7C15), (case.name, step)
            if step < len(case.answers):
                q, a = case.answers[step]
                eng.set_active_question(quiz, q); eng.record_answer(quiz, a); orc.record_answer(q, a, cases.WORKERS - 1)
    eng.close()
print("variant", variant, "max priority rel err vs oracle: %.2e" % worst)
