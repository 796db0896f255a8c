"""The gated fix (engine option pole_gate): a late quiz state's synchronous argmax selection with every listed question redone
(pole_gate = 0) and with only those that can still win (1), per cube shape; the picks side by side.
  python tools/gate_probe.py [QxKxT ...]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from probqa_amd import interop
f = interop.PqaEngineFactory()
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1000, 5, 1000), (4000, 5, 4000), (10000, 5, 10000)]
for Q, K, T in shapes:
    e = f.create_hip_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1), 0, Q, 0)
    e.set_option("select", 1)
    e.set_option("speculate", 0)
    e.fill_synthetic(8.0, 0.5, 20260928)
    q = e.start_quiz()
    guess, width = int(0.37 * T), max(1, 32 * T // 1000)
    reps = 300 if Q * T <= 4e6 else 40
    def timed():
        for _ in range(3):
            e.next_question_argmax(q)
        t0 = time.perf_counter()
        for _ in range(reps):
            pick = e.next_question_argmax(q)
        return (time.perf_counter() - t0) / reps * 1e6, pick
    line = []
    for step in range(40):
        e.set_option("pole_gate", 1)
        qq = e.next_question_argmax(q)
        x = qq * T // Q
        a = 0 if guess < x - width else 1 if guess < x else 2 if guess == x else 3 if guess <= x + width else 4
        e.record_answer(q, a)
        top = e.list_top_targets(q, 1)
        if step in (0, 3, 6) or (top and top[0].prob > 1 - 1e-6):
            res = {}
            for gate in (0, 1):
                e.set_option("pole_gate", gate)
                res[gate] = timed()
            line.append("after %d answers (top 1 - %.1e): all redone %.1f us, gated %.1f us, picks %d %d" % (
                step + 1, 1 - top[0].prob, res[0][0], res[1][0], res[0][1], res[1][1]))
            assert res[0][1] == res[1][1]
            if top[0].prob > 1 - 1e-6 and step > 6:
                break
    print("%dx%dx%d %s:\n  " % (Q, K, T, e.eval_kernel_name()) + "\n  ".join(line))
    e.close()
