#!/usr/bin/env python3
"""Where the wall time of _lib.mh_batch goes: the C call vs the Python wrapper,
with the draws taken in the call and with given draws.  (dev tool)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib, hostkernels  # noqa: E402
from bnpc_amd.model import TMIN, TMAX  # noqa: E402

kt = hostkernels.table()
lib = _lib.load()
real = lib.bnpc_mh_batch
inner = [0.0]


class Timed:
    def __call__(self, *a):
        t = time.perf_counter()
        r = real(*a)
        inner[0] += time.perf_counter() - t
        return r


lib.bnpc_mh_batch = Timed()
rng = np.random.RandomState(0)
sd = np.array([.1, .25, .5])
for G, M in ((3, 1000), (10, 1000)):
    old = np.clip(rng.uniform(size=(G, M)), TMIN, TMAX).astype(np.float32)
    n1 = rng.randint(0, 500, (G, M)).astype(np.int32)
    n0 = rng.randint(0, 500, (G, M)).astype(np.int32)
    draws = _lib.mh_draws(G, M, 3)
    for label, kw in (('draws in the call', {}), ('given draws', {'draws': draws})):
        for _ in range(50):
            _lib.mh_batch(kt, old, n1, n0, sd, TMIN, TMAX, .01, .2, .25, .25,
                False, False, want_prior=True, **kw)
        inner[0] = 0.0
        reps = 500
        t = time.perf_counter()
        for _ in range(reps):
            _lib.mh_batch(kt, old, n1, n0, sd, TMIN, TMAX, .01, .2, .25, .25,
                False, False, want_prior=True, **kw)
        total = time.perf_counter() - t
        print(f'G={G} M={M} {label:18s}: wall {1e6 * total / reps:6.1f} us, '
            f'C call {1e6 * inner[0] / reps:6.1f} us, wrapper '
            f'{1e6 * (total - inner[0]) / reps:5.1f} us', flush=True)
