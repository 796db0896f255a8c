"""Where the time of bench.py's quiz loop goes, call by call (wall clock around each C-ABI call, 1000 x 5 x 1000)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from probqa_amd import interop, synth

K, Q, T, SEED = 5, 1000, 1000, 1234
eng, err = interop.PqaEngineFactory().create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
eng.fill_synthetic(8.0, 0.5, SEED)
eng.set_option("select", 0)
eng.set_option("seed", SEED)
for name, val in [a.split("=") for a in sys.argv[1:]]:
    eng.set_option(name, int(val))
acc = collections.defaultdict(lambda: [0.0, 0])
def timed(name, fn, *a):
    t0 = time.perf_counter()
    r = fn(*a)
    d = acc[name]; d[0] += time.perf_counter() - t0; d[1] += 1
    return r
for rep in range(2):
    acc.clear()
    rng = np.random.default_rng(SEED)
    asked = 0
    t00 = time.perf_counter()
    for _ in range(600):
        guess = int(rng.integers(T))
        qz = timed("start_quiz", eng.start_quiz)
        for j in range(30):
            qq = timed("next_question[first]" if j == 0 else "next_question", eng.next_question, qz)
            asked += 1
            ans = synth.dichotomy_answer(qq * T // Q, guess, max(1, 32 * T // 1000))
            timed("record_answer", eng.record_answer, qz, ans)
            top1 = timed("list_top_targets", eng.list_top_targets, qz, 1)
            if top1 and top1[0].i_target == guess:
                break
        timed("record_quiz_target", eng.record_quiz_target, qz, guess)
        timed("release_quiz", eng.release_quiz, qz)
    total = time.perf_counter() - t00
print("questions/s %.0f  (%d questions, %.1f us each all in)" % (asked / total, asked, total / asked * 1e6))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-24s %6d calls %8.2f us/call %6.1f %% of the loop" % (k, n, t / n * 1e6, 100 * t / total))
print("spec_hits", eng.get_option("spec_hits"), "dropped", eng.get_option("spec_dropped"))
