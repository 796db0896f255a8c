"""One quiz state's sweeps, for a kernel trace: pole_cost.py QxKxT fresh|late N [option=value ...]
(rocprofv3 --kernel-trace --stats -- python tools/pole_cost.py ...: the averages of the sweep and of the fix behind it, apart)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from probqa_amd import interop
Q, K, T = (int(x) for x in sys.argv[1].split("x"))
state, n = sys.argv[2], int(sys.argv[3])
opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[4:]]
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, 0)
e.set_option("select", 1)
for name, value in opts:
    e.set_option(name, value)
e.fill_synthetic(8.0, 0.5, 20260928)
st = torch.cuda.Stream()
e.set_stream(st.cuda_stream)
q = e.start_quiz()
top = None
if state == "late":
    guess, width = int(0.37 * T), max(1, 32 * T // 1000)
    for step in range(40):
        qq = e.next_question_argmax(q)
        x = qq * T // Q
        a = 0 if guess < x - width else 1 if guess < x else 2 if guess == x else 3 if guess <= x + width else 4
        e.record_answer(q, a)
        top = e.list_top_targets(q, 1)
        if top and top[0].prob > 1 - 1e-6:
            break
for _ in range(3):
    e.enqueue_eval(q)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(n):
    e.enqueue_eval(q)
b.record(st)
torch.cuda.synchronize()
print("%dx%dx%d %s %s %s: %.1f us per sweep back to back" % (Q, K, T, e.eval_kernel_name(), state, opts, 1e3 * a.elapsed_time(b) / n),
      "(top posterior 1 - %.2g)" % (1 - top[0].prob) if top else "")
e.close()
