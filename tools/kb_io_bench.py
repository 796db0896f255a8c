""".kb save / load rates of a device-resident knowledge base (hip_engine_kb.cpp: IoRows -- two pinned staging buffers in turn, the
file's I/O of one batch under the copies of the other).  The file goes to --dir (default /dev/shm: memory-backed, so that the rate
is the engine's and not a disk's; pass a disk path to see that instead).
  python tools/kb_io_bench.py [QxKxT=10000x5x10000] [f32] [--dir PATH]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from probqa_amd import interop
args = [a for a in sys.argv[1:] if not a.startswith("--")]
where = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else "/dev/shm"
if "--dir" in sys.argv:
    args = [a for a in args if a != where]
dims = [a for a in args if "x" in a]
Q, K, T = (int(x) for x in (dims[0] if dims else "10000x5x10000").split("x"))
f32 = "f32" in args
kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if f32 else {}
f = interop.PqaEngineFactory()
e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
assert err is None, err
e.fill_synthetic(8.0, 0.5, 20260928)
quiz = e.start_quiz()
want = e.eval_priorities(quiz)
path = os.path.join(where, "pqa_io_bench_%d.kb" % os.getpid())
try:
    t0 = time.perf_counter()
    e.save_kb(path, False)
    t_save = time.perf_counter() - t0
    size = os.path.getsize(path)
    e.close()
    t0 = time.perf_counter()
    e2, err = f.load_cpu_engine(path)
    assert err is None, err
    t_load = time.perf_counter() - t0
    got = e2.eval_priorities(e2.start_quiz())
    assert np.array_equal(got, want), "the loaded knowledge base gives other priorities"
    print("%dx%dx%d %s: file %.2f GB in %s; save %.2f s = %.2f GB/s, load %.2f s = %.2f GB/s (engine creation and cube allocation included); priorities identical" % (
        Q, K, T, "fp32" if f32 else "fp64", size / 1e9, where, t_save, size / 1e9 / t_save, t_load, size / 1e9 / t_load))
    e2.close()
finally:
    if os.path.exists(path):
        os.unlink(path)
