"""How often the sweep re-evaluates questions in the reference's exact order, what it costs, and how close to the oracle it gets:
the three offenders of round 3's soak, and a late quiz state at S."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import cases, test_gpu_fuzz as tf
from test_gpu_parity import run_script
from probqa_amd import interop
factory = interop.PqaEngineFactory()
for i in (841, 2062, 6547):
    case = tf.random_case(i)
    for options in ([], [("eval_max_grid", 2)], [("server", 1)], [("pole_fix", 0)]):
        try:
            steps = run_script(case, factory, options)
            print(case.name, options, "max rel err per step:", ["%.2g" % s for s in steps])
        except AssertionError as ex:
            print(case.name, options, "FAIL", str(ex)[:200])
