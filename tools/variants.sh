set -x
python - <<'PY'
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
PY
python bench.py --steps 500 --warmup 50 --cpu-seconds 2
for v in 1 2 3 8 9; do python bench.py --steps 300 --warmup 20 --variant $v --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['eval_kernel'], d['roofline']['kernel_us'], d['value'], d['pipelined_selections_per_sec'])"; done
for v in 0 5 7 99; do python bench.py --config M --steps 20 --warmup 3 --variant $v --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['eval_kernel'], d['roofline']['kernel_us'], d['roofline']['achieved'], d['value'])"; done
