"""Timing of the single-quiz sweep of a Float engine and of a Double engine (tools): f32_single_bench.py Q K T [steps] [variant] [cluster_form] [cluster_shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probqa_amd import interop
Q, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cluster_form = int(sys.argv[6]) if len(sys.argv) > 6 else 0
cluster_shape = int(sys.argv[7]) if len(sys.argv) > 7 else 0
f = interop.PqaEngineFactory()
for prec in ("f32", "f64"):
    kw = dict(prec_type=interop.PrecisionType.FLOAT, prec_exponent=8, prec_mantissa=24) if prec == "f32" else {}
    e, err = f.create_cpu_engine(interop.EngineDefinition(K, Q, T, init_amount=0.1, **kw))
    assert err is None, err
    e.set_option("select", 1)
    if prec == "f32":
        e.set_option("eval_variant", variant)
    e.set_option("cluster_form", cluster_form)
    e.set_option("cluster_shape", cluster_shape)
    e.fill_synthetic(8.0, 0.5, 20260928)
    qz = e.start_quiz()
    for _ in range(10):
        p = e.next_question(qz)
    t0 = time.perf_counter()
    for _ in range(steps):
        p = e.next_question(qz)
    dt = time.perf_counter() - t0
    s = 4 if prec == "f32" else 8
    print("%dx%dx%d %s single quiz: %.1f us/selection, %.0f selections/s, %.0f GB/s of cube, kernel %s, pick=%d"
          % (Q, K, T, prec, 1e6 * dt / steps, steps / dt, Q * (K + 1) * T * s / (dt / steps) / 1e9, e.eval_kernel_name(), p))
    e.close()
