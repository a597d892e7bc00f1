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
