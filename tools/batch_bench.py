"""Timing of the batched sweeps (tools, not a test): PqaEngine_NextQuestionArgmaxBatch through the C ABI.
usage: batch_bench.py Q K T prec(f32|f64) B [tile] [batch_min] [reps] [qb] [groups] [tail]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop

Q, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec, B = sys.argv[4], int(sys.argv[5])
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 0
bmin = int(sys.argv[7]) if len(sys.argv) > 7 else 1
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
qb = int(sys.argv[9]) if len(sys.argv) > 9 else 0
groups = int(sys.argv[10]) if len(sys.argv) > 10 else 0
tail = int(sys.argv[11]) if len(sys.argv) > 11 else 1
f = interop.PqaEngineFactory()
if prec == "f32":
    d = interop.EngineDefinition(K, Q, T, init_amount=0.1, prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24)
else:
    d = interop.EngineDefinition(K, Q, T, init_amount=0.1)
eng, err = f.create_cpu_engine(d)
assert err is None, err
eng.fill_synthetic(8.0, 0.5, 20260928)
eng.set_option("batch_min", bmin)
eng.set_option("batch_tile", tile)
eng.set_option("batch_qb", qb)
eng.set_option("batch_groups", groups)
eng.set_option("batch_tail", tail)
quizzes = [eng.start_quiz() for _ in range(B)]
picks = eng.next_question_argmax_batch(quizzes)
t0 = time.perf_counter()
for _ in range(reps):
    picks = eng.next_question_argmax_batch(quizzes)
dt = (time.perf_counter() - t0) / reps
el = Q * K * T * B
print("%dx%dx%d %s B=%d tile=%d min=%d qb=%d groups=%d tail=%d: %.3f ms/batch, %.0f selections/s, %.3g element-evals/s, pick0=%d"
      % (Q, K, T, prec, B, tile, bmin, qb, groups, tail, dt * 1e3, B / dt, el / dt, picks[0]))
eng.close()
