"""probqa_amd -- MI355X-native implementation of ProbQA's question-evaluation hot path.

Only what the path needs lives here: `csrc/` (gfx950 HIP kernels, the host engine and the PqaCore C ABI ->
libPqaCore.so), `interop.py` (the reference Python wrapper's API over that ABI), `synth.py` (synthetic KBs) and
`dist.py` (question-axis sharding over torch.distributed / RCCL).
"""
