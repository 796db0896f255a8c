"""kernel time of RecordAnswer's launch (HIP events on the engine's stream), with and without the top-10 listing"""
import sys
sys.path.insert(0, '.')
import torch
from probqa_amd import interop
f = interop.PqaEngineFactory()
for small in (0, 1):
    for top in (0, 10):
        e = f.create_hip_engine(interop.EngineDefinition(5, 1000, 1000, init_amount=0.1), 0, 1000, 0)
        e.set_option("select", 1); e.fill_synthetic(8.0, 0.5, 20260928); e.set_option("top_cache", top)
        e.set_option("server", small)      # only for the 256-thread forms of the posterior kernels; no selection is made
        st = torch.cuda.Stream(); e.set_stream(st.cuda_stream)
        quiz = e.start_quiz()
        ts = []
        for i in range(60):
            e.set_active_question(quiz, i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); e.record_answer(quiz, i % 5); b.record(st); st.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts = sorted(ts[10:])
        print("256-thread form=%d top_cache=%2d: record kernel p50 %.1f us  p10 %.1f  p90 %.1f" % (small, top, ts[len(ts)//2], ts[len(ts)//10], ts[len(ts)*9//10]))
        e.close()
