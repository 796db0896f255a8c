# debugging aid: per-question parity of one eval kernel shape against the oracle, on fresh (cache-cold) engines
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cases, orclib
from probqa_amd import interop
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 30
f=interop.PqaEngineFactory()
tot=0; nbad=0
for T,Q in ((1000,12),(10000,300),(4000,40)):
    case=cases.Case("dbg",5,Q,T,seed=3)
    orc=case.make_oracle(); orc.start_quiz(16)
    _,opri=orc.eval(128)
    for trial in range(8):
        eng=case.make_engine(f)
        quiz=eng.start_quiz()
        eng.set_option("eval_variant", variant)
        for rep in range(2):
            pri=eng.eval_priorities(quiz)
            r=pri/opri
            bad=np.where(np.abs(r-1)>1e-9)[0]
            tot+=Q; nbad+=len(bad)
            if len(bad): print(T,Q,"trial",trial,"rep",rep,"bad",len(bad),"idx",bad[:10],"ratios",r[bad[:5]])
        eng.close()
print("variant",variant,"total question evals",tot,"bad",nbad)
