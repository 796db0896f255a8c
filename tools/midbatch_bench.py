"""Batched argmax selections at S for a few dozen quizzes: grid.y = quiz, the (quiz, chunk) form, row-sharing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop
f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1))
e.fill_synthetic(8.0, 0.5, 20260928); e.set_option("select", 1)
qz = [e.start_quiz() for _ in range(256)]
for i, q in enumerate(qz):
    e.set_active_question(q, (37 * i) % 1000); e.record_answer(q, i % 5)
ns = [int(x) for x in sys.argv[1:]] or [4, 8, 16, 24, 32, 48, 64, 128, 256]
for n in ns:
    out = []
    for form in (1, 3, 2):
        e.set_option("batch_form", form)
        for _ in range(3): p = e.next_question_argmax_batch(qz[:n])
        t0 = time.perf_counter(); reps = 30
        for _ in range(reps): p = e.next_question_argmax_batch(qz[:n])
        dt = (time.perf_counter() - t0) / reps
        out.append("%s %.1f us/batch %.2f us/quiz" % ({1: "grid.y", 3: "mid", 2: "rowsh"}[form], dt * 1e6, dt * 1e6 / n))
    print("n=%3d: %s" % (n, " | ".join(out)))
