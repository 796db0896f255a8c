# how the device's division-free Log2Hot compares with the oracle's operation-for-operation restatement
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, orclib
from probqa_amd import interop
lib = orclib.lib()
f = interop.PqaEngineFactory()
eng, err = f.create_cpu_engine(interop.EngineDefinition(2, 2, 4, init_amount=1.0))
rng = np.random.default_rng(7)
print("Log2Hot(1): device %.4g oracle %.4g" % (eng.log2hot(np.array([1.0]))[0], lib.orc_log2hot(1.0)))
for name, p in (("uniform (0,1)", rng.random(400000)), ("1 - 1e-6*U", 1.0 - rng.random(100000) * 1e-6), ("1e-12*U", rng.random(100000) * 1e-12)):
    dev = eng.log2hot(p); orc = np.array([lib.orc_log2hot(float(v)) for v in p])
    d = np.abs(dev - orc)
    print("%-14s differing %.4f%%  max abs diff %.3g  max diff/ulp(result) %.2f" % (name, 100 * (d != 0).mean(), d.max(), (d / np.spacing(np.abs(orc))).max()))
