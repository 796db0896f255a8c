"""Worker process of tests/test_gpu_parity.py::test_full_size_M_every_question_against_the_oracle: the oracle's priorities of one
slice of the synthetic cube (rows rebuilt on the host by the generator).  CPU only: imports neither the engine nor HIP."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_slice(args):
    q0, n, K, Q, T, seed, prior_bytes = args
    import numpy as np

    import orclib
    from probqa_amd import synth

    A, D, B = synth.synthetic_kb(K, n, T, 0.1, 8.0, 0.5, seed, q_offset=q0, q_total=Q)
    orc = orclib.Oracle(K, n, T, 0.1)
    orc.set_kb(A, D, B)
    orc.mants[:T] = np.frombuffer(prior_bytes, dtype=np.float64)
    _, opri = orc.eval_avx2(2)
    return q0, opri
