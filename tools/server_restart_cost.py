"""What a selection costs right after the resident sweep was told to leave (PqaHip_Synchronize), against the steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop
e, err = interop.PqaEngineFactory().create_cpu_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 1)
e.set_option("select", 1); e.set_option("server", 1)
q = e.start_quiz()
for _ in range(200): e.next_question(q)
first, second, steady = [], [], []
for rep in range(50):
    e.synchronize()
    t0 = time.perf_counter(); e.next_question(q); t1 = time.perf_counter(); e.next_question(q); t2 = time.perf_counter()
    for _ in range(20): e.next_question(q)
    t3 = time.perf_counter(); e.next_question(q); t4 = time.perf_counter()
    first.append(t1 - t0); second.append(t2 - t1); steady.append(t4 - t3)
med = lambda v: sorted(v)[len(v) // 2] * 1e6
print("first selection after the kernel left: %.1f us; the one after it: %.1f us; steady: %.1f us" % (med(first), med(second), med(steady)))
