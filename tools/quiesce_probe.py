"""How long the bracket of bench.py's timed region takes with the resident sweep alive (PqaHip_Quiesce against PqaHip_Synchronize)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probqa_amd import interop
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
e.set_option("select", 1)
e.fill_synthetic(8.0, 0.5, 1)
own = len(sys.argv) > 1 and sys.argv[1] == "own"
if not own:
    st = torch.cuda.Stream()
    e.set_stream(st.cuda_stream)
q = e.start_quiz()
e.set_option("server", 1)
for _ in range(5):
    e.next_question_argmax(q)
for name, fn in (("quiesce", e.quiesce), ("quiesce", e.quiesce), ("synchronize", e.synchronize)):
    for _ in range(3):
        e.next_question_argmax(q)
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
    e.next_question_argmax(q); t2 = time.perf_counter()
    e.next_question_argmax(q); t3 = time.perf_counter()
    print("%s: %.1f us; the step after it %.1f us, the next %.1f us" % (name, 1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2)))
