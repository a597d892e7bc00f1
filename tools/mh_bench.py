#!/usr/bin/env python3
"""us per call of the native MH parameter batch (bnpc_mh_batch) by batch shape
and host threads, the draws alone, and the Python wrapper's share.  Host only."""
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


def t(fn, reps=300):
    fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps // 5):
            fn()
        best = min(best, (time.perf_counter() - t0) / (reps // 5))
    return 1e6 * best


for G, M in ((10, 1000), (3, 1000), (1, 1000), (20, 2000), (50, 5000)):
    old = np.clip(rng.uniform(size=(G, M)), TMIN, TMAX).astype(np.float32)
    old[:, :M // 4] = np.float32(TMIN)
    old[:, M // 4:M // 2] = np.float32(TMAX)
    n1 = rng.randint(0, 500, (G, M)).astype(np.int32)
    n0 = rng.randint(0, 500, (G, M)).astype(np.int32)
    draws = _lib.mh_draws(G, M, 3)
    line = [f'G={G:3d} M={M:5d}: draws {t(lambda: _lib.mh_draws(G, M, 3)):7.1f} us |']
    for thr in (1, 4, 8, 16, 32, 64):
        if thr > (os.cpu_count() or 1):
            continue
        us = t(lambda: _lib.mh_batch(kt, old, n1, n0, sd, TMIN, TMAX, .01, .2,
            .25, .25, False, False, want_prior=True, threads=thr),
            reps=100 if G * M > 50000 else 300)
        us_given = t(lambda: _lib.mh_batch(kt, old, n1, n0, sd, TMIN, TMAX,
            .01, .2, .25, .25, False, False, want_prior=True, threads=thr,
            draws=draws), reps=100 if G * M > 50000 else 300)
        line.append(f't{thr}: {us:7.1f} ({us_given:7.1f} w/o draws)')
    print(' '.join(line), flush=True)
