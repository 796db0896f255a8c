import sys, time
sys.path.insert(0, '.')
from probqa_amd import interop
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
e.set_option("select", 1); e.fill_synthetic(8.0, 0.5, 20260928); e.set_option("server", int(sys.argv[1]))
T = {"start": [], "next": [], "rec": [], "top": [], "release": []}
for z in range(6):
    t = time.perf_counter(); quiz = e.start_quiz(); T["start"].append(time.perf_counter() - t)
    for i in range(20):
        t0 = time.perf_counter(); q = e.next_question(quiz)
        t1 = time.perf_counter(); e.record_answer(quiz, (q + i) % 5)
        t2 = time.perf_counter(); e.list_top_targets(quiz, 10)
        t3 = time.perf_counter()
        T["next"].append(t1 - t0); T["rec"].append(t2 - t1); T["top"].append(t3 - t2)
    t = time.perf_counter(); e.release_quiz(quiz); T["release"].append(time.perf_counter() - t)
for k, v in T.items():
    v = sorted(v[2:]); print(k, "p50 %.1f us  max %.1f us" % (1e6 * v[len(v) // 2], 1e6 * v[-1]))
e.close()
