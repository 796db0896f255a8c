import sys, time
sys.path.insert(0, '.')
from probqa_amd import interop
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
e.fill_synthetic(8.0, 0.5, 20260928); e.set_option("seed", 5)
quiz = e.start_quiz()
for rep in range(3):
    for fused in (0, 1):
        e.set_option("fused_sampled", fused)
        for _ in range(300): e.next_question_sampled(quiz, 123456789)
        t0 = time.perf_counter()
        for i in range(2000): q = e.next_question_sampled(quiz, 0x9E3779B97F4A7C15 * (i + 1) % 2**64)
        dt = time.perf_counter() - t0
        print("fused=%d: %.1f us per sampled NextQuestion (last pick %d)" % (fused, 1e6 * dt / 2000, q))
e.set_option("select", 1)
t0 = time.perf_counter()
for i in range(2000): q = e.next_question_argmax(quiz)
print("argmax: %.1f us" % (1e6 * (time.perf_counter() - t0) / 2000))
