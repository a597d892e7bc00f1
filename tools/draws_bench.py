"""Rate of the three draws of a parameter batch (bnpc_mt_mh_draws: choice(sd),
truncnorm's uniforms, the acceptance uniforms) on this host, one thread:
python3 tools/draws_bench.py [G M]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np                                          # noqa: E402
from bnpc_amd import _lib                                   # noqa: E402

G, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 5000)
lib = _lib.load()
np.random.seed(1)
sd = np.zeros(G * M, np.int32)
U = np.zeros(G * M)
u = np.zeros(G * M)
with _lib.NumpyStream() as rng:
    for n_sd in (3, 4):
        best = 1e9
        for rep in range(5):
            t = time.perf_counter()
            for _ in range(100):
                lib.bnpc_mt_mh_draws(rng, G, M, n_sd, _lib.ptr(sd, C.c_int32),
                    _lib.ptr(U, C.c_double), _lib.ptr(u, C.c_double))
            best = min(best, (time.perf_counter() - t) / 100 * 1e6)
        words = G * M * (4 + (4 / 3 if n_sd == 3 else 1))
        print(f'{G} x {M}, {n_sd} proposal widths: {best:.1f} us per batch '
              f'= {best * 1e3 / words:.3f} ns per 32-bit word')
a = np.empty(G * M * 2)
t = time.perf_counter()
for _ in range(100):
    np.random.random_sample(G * M * 2)
print(f'numpy random_sample of the same doubles: '
      f'{(time.perf_counter() - t) / 100 * 1e6:.1f} us')
print(open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0])

# the same draws into a destination that is not in the cache (64 MB walked part
# by part), ordinary and non-temporal stores
f = getattr(lib, '_Z19bnpc_mt_mh_draws_toP12bnpc_mt19937lllPiPdS2_b', None)
if f is not None:
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                  C.c_void_p, C.c_void_p, C.c_bool]
    E = G * M
    parts = max(2, (8 << 20) // E)
    sd_big = np.zeros(E * parts, np.int32)
    U_big = np.zeros(E * parts)
    u_big = np.zeros(E * parts)
    with _lib.NumpyStream() as rng:
        for stream in (False, True):
            t = time.perf_counter()
            for rep in range(3):
                for p in range(parts):
                    f(C.cast(rng, C.c_void_p), G, M, 3,
                      sd_big[p * E:].ctypes.data, U_big[p * E:].ctypes.data,
                      u_big[p * E:].ctypes.data, stream)
            per = (time.perf_counter() - t) / (3 * parts) * 1e6
            print(f'cold destination, non-temporal stores {stream}: '
                  f'{per:.1f} us per batch')
