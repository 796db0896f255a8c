"""tools/server_check.py -- resident sweep (option "server") against the launch-per-selection path: same selections, same
top targets, timings of both.  Run on a GPU box under `timeout`."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from probqa_amd import interop

f = interop.PqaEngineFactory()
Q = K5 = None
def make(server):
    e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
    e.set_option("select", 1)
    e.fill_synthetic(8.0, 0.5, 20260928)
    e.set_option("server", server)
    return e

def script(e, n_quiz=3, n_steps=12, sleep_at=()):
    out = []
    for z in range(n_quiz):
        quiz = e.start_quiz()
        for i in range(n_steps):
            q = e.next_question(quiz)
            if i in sleep_at:
                time.sleep(0.02)      # lets the resident kernel time out and leave
            e.record_answer(quiz, (q * 7 + i + z) % 5)
            top = e.list_top_targets(quiz, 5)
            out.append((q, tuple((t.i_target, t.prob) for t in top)))
        e.release_quiz(quiz)
    return out

a = script(make(0))
e1 = make(1)
print("server_active", e1.get_option("server_active"))
b = script(e1, sleep_at=(3, 7))
same = a == b
print("selections + top targets identical:", same)
if not same:
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print("first difference at step", i, x, y); break

def rate(e, n=5000):
    quiz = e.start_quiz()
    for _ in range(500): e.next_question_argmax(quiz)
    t0 = time.perf_counter()
    for _ in range(n): s = e.next_question_argmax(quiz)
    dt = time.perf_counter() - t0
    return n / dt, s

def quiz_rate(e, n_quiz=20, n_steps=30):
    t0 = time.perf_counter(); steps = 0
    for z in range(n_quiz):
        quiz = e.start_quiz()
        for i in range(n_steps):
            q = e.next_question(quiz); e.record_answer(quiz, (q + i) % 5); e.list_top_targets(quiz, 10); steps += 1
        e.release_quiz(quiz)
    return steps / (time.perf_counter() - t0)

for srv in (0, 1, 0, 1):
    e = make(srv)
    r, s = rate(e)
    print("server=%d: %.0f selections/s (question %d); quiz steps/s %.0f" % (srv, r, s, quiz_rate(e)))
    e.close()
print("done")
