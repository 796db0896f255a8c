"""Timing of the sampled NextQuestion forms (tools): host_sampled 1 (one launch + host selector) vs 0 (sweep + selector kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop
Q, K, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1000, 5, 1000)))
f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 20260928)
qz = e.start_quiz()
for hs in (1, 0, 1, 0):
    e.set_option("host_sampled", hs)
    picks = []
    for i in range(200):
        picks.append(e.next_question_sampled(qz, 0x9E3779B97F4A7C15 * (i + 1) % 2**64))
    t0 = time.perf_counter()
    n = 3000
    for i in range(n):
        e.next_question_sampled(qz, 12345 + i)
    dt = time.perf_counter() - t0
    print("host_sampled=%d: %.1f us per sampled selection; picks[:8]=%s" % (hs, 1e6 * dt / n, picks[:8]))
e.close()
# the same through the resident sweep
e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 20260928)
qz = e.start_quiz()
e.set_option("server", 1)
if e.get_option("server_active") == 1:
    picks = [e.next_question_sampled(qz, 0x9E3779B97F4A7C15 * (i + 1) % 2**64) for i in range(200)]
    t0 = time.perf_counter()
    for i in range(3000):
        e.next_question_sampled(qz, 12345 + i)
    dt = time.perf_counter() - t0
    print("resident + host selector: %.1f us per sampled selection; picks[:8]=%s" % (1e6 * dt / 3000, picks[:8]))
e.close()
