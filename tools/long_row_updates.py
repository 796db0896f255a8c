"""Posterior updates at BASELINE configs[4]'s row length (100000 targets): one-quiz kernels and the batched launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probqa_amd import interop

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 64, T, init_amount=0.1, prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24), 0, 64, 0)
e.fill_synthetic(8.0, 0.5, 1)
e.set_option("speculate", 0)
st = torch.cuda.Stream(); e.set_stream(st.cuda_stream)
def ev(fn, reps=10):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); st.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
z = e.start_quiz()
print("T=%d: start_quiz kernel %.1f us" % (T, ev(lambda: e.release_quiz(e.start_quiz()))))
i = [0]
def rec():
    e.set_active_question(z, i[0] % 64); i[0] += 1
    e.record_answer(z, i[0] % 5)
print("record_answer kernel %.1f us" % ev(rec))
n = 256
t0 = time.perf_counter(); qs = e.start_quiz_batch(n); e.synchronize(); t1 = time.perf_counter()
print("start_quiz_batch(%d): %.2f ms wall" % (n, 1e3 * (t1 - t0)))
for r in range(3):
    for k, q in enumerate(qs):
        e.set_active_question(q, (k + r) % 64)
    e.synchronize()
    t0 = time.perf_counter(); e.record_answer_batch(qs, [(k + r) % 5 for k in range(n)]); e.synchronize(); t1 = time.perf_counter()
    print("record_answer_batch(%d): %.2f ms wall" % (n, 1e3 * (t1 - t0)))
