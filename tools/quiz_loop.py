# latency of every call of a quiz step on the 1000 x 5 x 1000 cube, through the Python wrapper of the C ABI
import sys, time; sys.path.insert(0,'.')
import numpy as np
from probqa_amd import interop
f=interop.PqaEngineFactory()
K,Q,T=5,1000,1000
eng=f.create_hip_engine(interop.EngineDefinition(K,Q,T,init_amount=0.1),0,Q,0)
eng.fill_synthetic(8.0,0.5,1)
rng=np.random.default_rng(0)
acc={k:0.0 for k in ("start_quiz","next_question(sampled)","next_question(argmax)","record_answer","list_top_targets(10)","record_quiz_target","release_quiz")}
n_quiz, n_steps = 40, 30
for mode in (0,1):
    eng.set_option("select", mode)
    key="next_question(argmax)" if mode else "next_question(sampled)"
    t_all=time.perf_counter()
    for _ in range(n_quiz):
        t=time.perf_counter(); quiz=eng.start_quiz(); acc["start_quiz"]+=time.perf_counter()-t
        for _ in range(n_steps):
            t=time.perf_counter(); q=eng.next_question(quiz); acc[key]+=time.perf_counter()-t
            t=time.perf_counter(); eng.record_answer(quiz,int(rng.integers(0,K))); acc["record_answer"]+=time.perf_counter()-t
            t=time.perf_counter(); top=eng.list_top_targets(quiz,10); acc["list_top_targets(10)"]+=time.perf_counter()-t
        t=time.perf_counter(); eng.record_quiz_target(quiz, top[0].i_target); acc["record_quiz_target"]+=time.perf_counter()-t
        t=time.perf_counter(); eng.release_quiz(quiz); acc["release_quiz"]+=time.perf_counter()-t
    dt=time.perf_counter()-t_all
    print("select=%d: %.0f quiz steps/s (question + answer + top-10), %.0f quizzes/s of %d steps incl. training" % (mode, n_quiz*n_steps/dt, n_quiz/dt, n_steps))
cnt={"start_quiz":2*n_quiz,"release_quiz":2*n_quiz,"record_quiz_target":2*n_quiz,"record_answer":2*n_quiz*n_steps,"list_top_targets(10)":2*n_quiz*n_steps,"next_question(sampled)":n_quiz*n_steps,"next_question(argmax)":n_quiz*n_steps}
for k,v in acc.items(): print("%-24s %7.1f us/call" % (k, 1e6*v/cnt[k]))
