import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop, synth
K, Q, T = 4, 48, 300
def run(first_graph, max_grid, vram, resume1=True):
    kb = synth.synthetic_kb(K, Q, T, 0.1, 8.0, 0.5, 130)
    e, err = interop.PqaEngineFactory().create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1))
    e.set_kb(*kb); e.set_option("workers", 16); e.set_option("select", 1)
    e.set_option("server_vram_mailbox", vram)
    e.set_option("server", 1); e.set_option("server_idle_us", 300); e.set_option("eval_max_grid", max_grid)
    AQ = interop.AnsweredQuestion
    q0 = e.resume_quiz([AQ(27, 0), AQ(11, 3)])
    e.set_option("use_graph", 1 if first_graph else 0)
    a = e.next_question(q0)
    e.set_option("use_graph", 0)
    q1 = e.resume_quiz([AQ(19, 2)]) if resume1 else e.start_quiz()
    b = e.next_question(q1)
    e.set_option("server", 0)
    want = int(np.argmax(e.eval_priorities(q1)))
    print("first via %s, max_grid %d, vram %d, quiz1 %s: got %d want %d %s" % ("graph" if first_graph else "server", max_grid, vram, "resumed" if resume1 else "started", b, want, "OK" if b == want else "WRONG"))
    e.close()
for fg in (1, 0):
    for mg in (5, 0):
        for vram in (1, 0):
            run(fg, mg, vram)
run(1, 5, 1, False)
