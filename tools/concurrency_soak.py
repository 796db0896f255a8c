"""Soak of the concurrent-client paths (combined sweeps, posted operations, group commit): learner threads of random counts, both
selectors, with and without training, against the one-thread digest where the transcript is interleaving-independent.
usage: python tools/concurrency_soak.py [rounds]      (every round is bounded by the client's own watchdog; the same loop runs in
the suite: tests/test_gpu_soak.py)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_soak as soak
from probqa_amd import interop
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0 = time.time()
bad = soak.concurrency_rounds(rounds, interop.PqaEngineFactory(), report=lambda s: print(s, flush=True))
print("soak %s: %d rounds in %.0f s" % ("passed" if not bad else "FAILED %r" % (bad,), rounds, time.time() - t0))
sys.exit(1 if bad else 0)
