# fixed per-selection overhead of the synchronous path: a KB so small that the sweep itself is negligible
import sys, time; sys.path.insert(0,'.')
import torch
from probqa_amd import interop
f=interop.PqaEngineFactory()
for (K,Q,T) in ((2,4,16),(5,256,1000),(5,1000,1000)):
    eng=f.create_hip_engine(interop.EngineDefinition(K,Q,T,init_amount=0.1),0,Q,0)
    eng.set_option("select",1); eng.fill_synthetic(8.0,0.5,1)
    s=torch.cuda.Stream(); torch.cuda.set_stream(s); eng.set_stream(s.cuda_stream)
    quiz=eng.start_quiz()
    for _ in range(200): eng.next_question_argmax(quiz)
    torch.cuda.synchronize()
    n=3000; t0=time.perf_counter()
    for _ in range(n): eng.next_question_argmax(quiz)
    dt=(time.perf_counter()-t0)/n
    t0=time.perf_counter()
    for _ in range(n): eng.enqueue_select_argmax(quiz)
    t_enq=(time.perf_counter()-t0)/n
    torch.cuda.synchronize()
    t_tot=(time.perf_counter()-t0)/n
    print("K=%d Q=%d T=%d: sync step %.2f us | enqueue-only host cost %.2f us | pipelined %.2f us/step" % (K,Q,T,dt*1e6,t_enq*1e6,t_tot*1e6))
    eng.close()
