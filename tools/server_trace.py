"""per-step device timestamps of the resident sweep (build with -DPQA_SERVER_TRACE); 100 MHz ticks"""
import sys, time, ctypes
sys.path.insert(0, '.')
from probqa_amd import interop
f = interop.PqaEngineFactory()
e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
e.set_option("select", 1); e.fill_synthetic(8.0, 0.5, 20260928); e.set_option("server", 1)
quiz = e.start_quiz()
e.next_question_argmax(quiz)
mb = e.get_option("debug_mailbox")
pad = (ctypes.c_uint64 * 16).from_address(mb)   # whole 128-byte mailbox as qwords; pad[] starts at qword 11
rows = []
for i in range(3000):
    t0 = time.perf_counter(); e.next_question_argmax(quiz); t1 = time.perf_counter()
    if i > 500:
        tA0, tB0, tAl, tBl, tdet = pad[11], pad[12], pad[13], pad[14], pad[15]
        rows.append((1e6 * (t1 - t0), (tA0 - tdet) / 100, (tB0 - tA0) / 100, (tAl - tdet) / 100, (tBl - tAl) / 100))
import statistics
names = ["host step us", "wg0 detect->sweep", "wg0 sweep+finish", "lastwg detect(wg0)->sweep", "lastwg sweep"]
for k, n in enumerate(names):
    col = sorted(r[k] for r in rows); print("%-28s p50 %.2f  p90 %.2f" % (n, col[len(col) // 2], col[len(col) * 9 // 10]))
e.close()
