"""More seeds of tests/test_gpu_fuzz.py than the suite runs: fuzz_more.py first last   (= pytest -m gpu tests/test_gpu_soak.py
--soak N --soak-first FIRST, without pytest around it)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_soak as soak
from probqa_amd import interop
first, last = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = soak.fuzz_cases(first, last, interop.PqaEngineFactory())
print("cases %d..%d: %d failures, %.0f s" % (first, last, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
