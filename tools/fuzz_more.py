"""More seeds of tests/test_gpu_fuzz.py than the suite runs (a soak, not a test): fuzz_more.py first last"""
import os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_fuzz as tf
from probqa_amd import interop
first, last = int(sys.argv[1]), int(sys.argv[2])
factory = interop.PqaEngineFactory()
bad = 0
t0 = time.time()
for i in range(first, last):
    for name, fn in (("single", tf.test_random_case), ("batched", tf.test_random_case_batched)):
        try:
            fn(i, factory)
        except BaseException as ex:  # noqa: BLE001
            if type(ex).__name__ in ("Skipped",):
                continue
            bad += 1
            print("FAIL", name, i, tf.random_case(i).name, repr(ex)[:300])
            traceback.print_exc(limit=2)
print("cases %d..%d: %d failures, %.0f s" % (first, last, bad, time.time() - t0))
