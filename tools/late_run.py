"""The worst step of late-quiz-state cases (tests/test_gpu_late.py: late_case(i)) by number: late_run.py I [I ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_late as tl  # noqa: E402
from probqa_amd import interop  # noqa: E402

factory = interop.PqaEngineFactory()
for i in [int(x) for x in sys.argv[1:]]:
    leg, case, worst = tl.run_late_case(i, factory)
    print("%d %s %s worst relative deviation %.3g" % (i, leg, case.name, worst), flush=True)
