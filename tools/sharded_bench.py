"""Timing of the one-process sharded engine (tools): PQA_DEVICES=<spec> python tools/sharded_bench.py Q K T [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop
Q, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
assert err is None, err
e.set_option("select", 1)
e.fill_synthetic(8.0, 0.5, 20260928)
qz = e.start_quiz()
for name, fn in (("argmax", lambda: e.next_question(qz)), ("sampled", lambda: e.next_question_sampled(qz, 12345))):
    for _ in range(50):
        p = fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        p = fn()
    dt = time.perf_counter() - t0
    print("PQA_DEVICES=%s %dx%dx%d %s: %.1f us/step, %.0f selections/s, shards=%d, pick=%d"
          % (os.environ.get("PQA_DEVICES"), Q, K, T, name, 1e6 * dt / steps, steps / dt, e.get_option("shards"), p))
e.close()
