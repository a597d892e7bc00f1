#!/usr/bin/env python3
"""Time individual bnpc_mh_batch calls (dev tool): shapes x threads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib, hostkernels  # noqa: E402
from bnpc_amd.model import TMIN, TMAX  # noqa: E402

kt = hostkernels.table()
rng = np.random.RandomState(0)
sd = np.array([.1, .25, .5])
for G, M in ((20, 2000), (50, 5000)):
    old = np.clip(rng.uniform(size=(G, M)), TMIN, TMAX).astype(np.float32)
    n1 = rng.randint(0, 500, (G, M)).astype(np.int32)
    n0 = rng.randint(0, 500, (G, M)).astype(np.int32)
    for thr in (16, 32, 64):
        ts = []
        for i in range(8):
            t = time.perf_counter()
            _lib.mh_batch(kt, old, n1, n0, sd, TMIN, TMAX, .01, .2, .25, .25,
                False, False, want_prior=True, threads=thr)
            ts.append(1e3 * (time.perf_counter() - t))
        print(G, M, thr, ' '.join(f'{x:.2f}' for x in ts), 'ms', flush=True)
