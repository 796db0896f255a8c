"""Three shards of one engine on ONE device, rows of 20000 targets: three cluster sweeps in flight at once on three streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from probqa_amd import interop
K, Q, T = 5, 240, 20000
f = interop.PqaEngineFactory()
whole, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
whole.fill_synthetic(8.0, 0.5, 3); whole.set_option("select", 1)
os.environ["PQA_DEVICES"] = sys.argv[1] if len(sys.argv) > 1 else "0,0,0"
sh, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
del os.environ["PQA_DEVICES"]
sh.fill_synthetic(8.0, 0.5, 3); sh.set_option("select", 1)
print("shards", sh.get_option("shards"), "kernel of the whole engine:", whole.eval_kernel_name())
qa, qb = whole.start_quiz(), sh.start_quiz()
t0 = time.time()
for step in range(12):
    a, b = whole.next_question(qa), sh.next_question(qb)
    assert a == b, (step, a, b)
    whole.record_answer(qa, step % K); sh.record_answer(qb, step % K)
assert np.array_equal(whole.get_priors(qa), sh.get_priors(qb))
print("12 selections agree, %.2f s" % (time.time() - t0))
