import sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import test_gpu_late as t
from probqa_amd import interop
f=interop.PqaEngineFactory()
for i in [int(x) for x in sys.argv[1:]]:
    print(i, t.run_late_case(i, f)[2], flush=True)
