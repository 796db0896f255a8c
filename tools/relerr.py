import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cases, orclib
from probqa_amd import interop
f=interop.PqaEngineFactory()
for case in cases.small_cases():
    orc=case.make_oracle(); eng=case.make_engine(f)
    quiz=eng.start_quiz(); orc.start_quiz(16)
    for q,a in case.answers:
        eng.set_active_question(quiz,q); eng.record_answer(quiz,a); orc.record_answer(q,a,15)
    pri=eng.eval_priorities(quiz); _,opri=orc.eval(128)
    m=opri!=0
    print(case.name, "max rel", cases.rel_err(pri[m],opri[m]).max(), "prior max", orc.priors().max())
