import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases
from probqa_amd import interop
f = interop.PqaEngineFactory()
case = cases.Case("conc", 3, 40, 100, seed=3, qgaps=[5])
eng = case.make_engine(f); eng.set_option("select", 1)
eng.set_option("combine", 0)
os.environ["PQA_CLIENT_VERBOSE"] = "1"
print(interop.run_learners(eng, 16, 48, 8, seed=2, train=False))
