"""lambda_amd -- MI355X-native seed-extension engine for lambda3's hot path.

Only what the path needs lives here: ``csrc/`` (gfx950 HIP kernels, the C ABI of include/lambda_ext.h and the C++
host mirror of the reference's extension driver), ``capi`` (ctypes binding of that C ABI), ``synth`` (synthetic
seed batches for tests and bench.py) and ``build`` (hipcc driver).
"""
__all__ = ["capi", "synth", "build"]
