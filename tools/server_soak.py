"""tools/server_soak.py -- the leave/post race of the resident sweep, hammered: the idle time is set to 100 us and requests
arrive after random pauses of 0..300 us, so that the kernel is leaving about as often as a request arrives.  Every
selection must be the right one and none may time out."""
import random, sys, time
sys.path.insert(0, '.')
from probqa_amd import interop
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 400, 800, init_amount=0.1), 0, 400, 0)
e.set_option("select", 1); e.fill_synthetic(8.0, 0.5, 7)
quiz = e.start_quiz()
want = e.next_question_argmax(quiz)
e.set_option("server", 1); e.set_option("server_idle_us", 100)
rnd = random.Random(1)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 20
n = bad = 0
worst = 0.0
while time.time() < t_end:
    pause = rnd.random() * 300e-6
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < pause:
        pass
    t1 = time.perf_counter()
    got = e.next_question_argmax(quiz)
    worst = max(worst, time.perf_counter() - t1)
    n += 1
    bad += got != want
print("selections %d, wrong %d, slowest %.1f us" % (n, bad, 1e6 * worst))
e.close()
sys.exit(1 if bad else 0)
