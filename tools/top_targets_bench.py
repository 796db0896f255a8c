"""ListTopTargets on the device: the batched listing of BASELINE configs[4]'s quiz batch (256 quizzes x top-N over 100000 targets)
and single listings over long rows.  Wall time of the C-ABI call (ctypes included) and, by HIP events on the engine's stream, the
device time of its launches.
  python tools/top_targets_bench.py [T=100000] [quizzes=256] [N=10] [only]     (only: just the batched top-N, for a profiler)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from probqa_amd import interop

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
K, Q = 5, 8
f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24))
assert err is None, err
e.fill_synthetic(8.0, 0.5, 20260928)
st = torch.cuda.Stream()
e.set_stream(st.cuda_stream)
quizzes = e.start_quiz_batch(NQ)
rng = np.random.default_rng(0)
for step in range(2):
    for quiz in quizzes:
        e.set_active_question(quiz, step)
    e.record_answer_batch(quizzes, [int(x) for x in rng.integers(0, K, size=NQ)])


def timed(fn, reps=30):
    fn()
    wall, dev = [], []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        t0 = time.perf_counter()
        fn()
        wall.append((time.perf_counter() - t0) * 1e6)
        b.record(st)
        st.synchronize()
        dev.append(a.elapsed_time(b) * 1e3)
    wall.sort()
    dev.sort()
    return wall[len(wall) // 2], dev[len(dev) // 2]


import ctypes

ONLY = len(sys.argv) > 4
c_quizzes = (ctypes.c_int64 * NQ)(*quizzes)
c_counts = (ctypes.c_int64 * NQ)()
for n in ([N] if ONLY else sorted({1, N, 32, 256})):
    # the C-ABI call itself (the wrapper's list of Python objects per record costs more than the device work)
    c_dest = (interop.CiRatedTarget * (NQ * n))()
    w, d = timed(lambda: interop._check(interop._lib.PqaEngine_ListTopTargetsBatch(e.c_engine, NQ, c_quizzes, n, c_dest, c_counts)))
    assert all(c == n for c in c_counts)
    print("batched: %d quizzes x top-%d over %d targets: C-ABI call %.0f us (events on the stream: %.0f us) = %.2f us per quiz" % (NQ, n, T, w, d, w / NQ))
if ONLY:
    e.close()
    sys.exit(0)
for n in sorted({1, N, 32, 256}):
    w, d = timed(lambda: e.list_top_targets(quizzes[0], n))
    print("single : top-%d over %d targets: call %.0f us (events: %.0f us)" % (n, T, w, d))
# ... and where every probability ties (quizzes just started on a knowledge base with equal target counts): the reference's order among
# equals is its heaps' -- reproduced on the device (LaunchTopTargetsExact) behind the fast listing that found the tie
fresh, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24))
assert err is None, err
fresh.set_stream(st.cuda_stream)
fq = fresh.start_quiz_batch(NQ)
c_fq = (ctypes.c_int64 * NQ)(*fq)
for n in sorted({1, N}):
    c_dest = (interop.CiRatedTarget * (NQ * n))()
    before = fresh.get_option("top_exact_listings")
    w, d = timed(lambda: interop._check(interop._lib.PqaEngine_ListTopTargetsBatch(fresh.c_engine, NQ, c_fq, n, c_dest, c_counts)))
    assert fresh.get_option("top_exact_listings") > before
    print("all equal: %d quizzes x top-%d over %d targets: C-ABI call %.0f us (the fast listing, then the heaps' order on the device)" % (NQ, n, T, w))
    w1, d1 = timed(lambda: fresh.list_top_targets(fq[0], n))
    print("all equal: one quiz, top-%d: call %.0f us" % (n, w1))
fresh.close()
w, d = timed(lambda: e.list_top_targets(quizzes[0], 300), reps=5)
print("single : top-300 (host path: the posterior is copied): call %.0f us" % w)
e.close()
