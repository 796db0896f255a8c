"""Launched sweep time (HIP events, back-to-back) for a list of cube shapes, the resident step at S, and a late quiz state's sweep:
sweep_timing.py [QxKxT ...] [option=value ...]   (engine options, e.g. pole_fix=0)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from probqa_amd import interop
f = interop.PqaEngineFactory()
opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "=" not in a] or [(1000, 5, 1000), (2000, 5, 2000), (4000, 5, 4000), (8000, 5, 8000), (10000, 5, 10000)]
st = torch.cuda.Stream()
for Q, K, T in shapes:
    e = f.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, 0)
    e.set_option("select", 1)
    for name, value in opts:
        e.set_option(name, value)
    e.fill_synthetic(8.0, 0.5, 20260928)
    e.set_stream(st.cuda_stream)
    q = e.start_quiz()
    def kernel_us(qz, n):
        for _ in range(5):
            e.enqueue_eval(qz)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(n):
            e.enqueue_eval(qz)
        b.record(st)
        torch.cuda.synchronize()
        return 1e3 * a.elapsed_time(b) / n
    n = 200 if Q * T <= 4e6 else 40
    fresh = kernel_us(q, n)
    # a late quiz: answers consistent with one target until the posterior sits on it
    guess = int(0.37 * T)
    width = max(1, 32 * T // 1000)
    late = None
    for step in range(40):
        qq = e.next_question_argmax(q)
        x = qq * T // Q
        a = 0 if guess < x - width else 1 if guess < x else 2 if guess == x else 3 if guess <= x + width else 4
        e.record_answer(q, a)
        top = e.list_top_targets(q, 1)
        if top and top[0].prob > 1 - 1e-6:
            late = (step + 1, top[0].prob)
            break
    late_us = kernel_us(q, n)
    print("%dx%dx%d %s: sweep %.1f us after StartQuiz; %.1f us after %s" % (Q, K, T, e.eval_kernel_name(), fresh, late_us,
          "%d answers (top posterior 1 - %.2g)" % (late[0], 1 - late[1]) if late else "40 answers (never concentrated)"))
    e.close()
