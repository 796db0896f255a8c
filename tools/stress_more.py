"""More seeds of tests/test_gpu_stress.py's model-based random interleaving, in every mode (a soak, not a test)."""
import os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_stress as ts
from probqa_amd import interop
first, last = int(sys.argv[1]), int(sys.argv[2])
factory = interop.PqaEngineFactory()
bad, t0 = 0, time.time()
for seed in range(first, last):
    for mode in ("plain", "resident", "row_sharing_batches", "three_shards"):
        try:
            ts.test_random_interleaving_against_per_quiz_oracles(factory, seed, mode)
        except BaseException as ex:  # noqa: BLE001
            bad += 1
            print("FAIL", seed, mode, repr(ex)[:300])
            traceback.print_exc(limit=3)
print("seeds %d..%d x 4 modes: %d failures, %.0f s" % (first, last, bad, time.time() - t0))
