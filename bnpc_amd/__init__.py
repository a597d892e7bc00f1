"""bnpc_amd: MI355X-native Bernoulli log-likelihood hot path of BnpC.

Holds only what the path needs:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI (libbnpc_hip.so)
  _lib.py    ctypes binding of include/bnpc_hip.h
  model.py   host-side mirror of the reference's CRP / CRP_errors_learning
  mcmc.py    the sampler driver (the caller of the hot path)
"""
__version__ = '0.1.0'
