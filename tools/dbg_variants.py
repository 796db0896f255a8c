import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cases, orclib
from probqa_amd import interop
f=interop.PqaEngineFactory()
for T in (1000, 1500, 2500, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 10000):
    case=cases.Case("dbg",5,12,T,seed=3)
    orc=case.make_oracle(); eng=case.make_engine(f)
    quiz=eng.start_quiz(); orc.start_quiz(16)
    _,opri=orc.eval(128)
    for v in tuple(range(1, 25)) + (99,):
        eng.set_option("eval_variant", v)
        try:
            pri=eng.eval_priorities(quiz)
            print(T, v, eng.eval_kernel_name(), "max rel", cases.rel_err(pri,opri).max())
        except Exception as e:
            print(T, v, "n/a")
