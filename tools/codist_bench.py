#!/usr/bin/env python3
"""Time the co-clustering kernel against the NumPy/pdist path (dev tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
from oracle import posterior_numpy as Q  # noqa: E402

rng = np.random.RandomState(0)
for S, N in ((3350, 1000), (3350, 5000), (1000, 10000)):
    a = np.tile(rng.randint(0, 10, N), (S, 1))
    flip = rng.random_sample((S, N)) < 0.05
    a[flip] = rng.randint(0, 10, flip.sum())
    _lib.codist(a[:2])
    t0 = time.perf_counter()
    d = _lib.codist(a)
    t_gpu = time.perf_counter() - t0
    ns = max(2, int(2e8 / (N * N / 2)))
    t0 = time.perf_counter()
    Q.get_dist(a[:ns])
    t_cpu = (time.perf_counter() - t0) / ns * S
    print(f'S={S} N={N}: GPU (incl. H2D/D2H) {t_gpu * 1e3:.1f} ms = '
        f'{S * N * (N - 1) / 2 / t_gpu:.3e} pair-samples/s; NumPy pdist path '
        f'~{t_cpu:.1f} s (extrapolated from {ns} samples) -> '
        f'{t_cpu / t_gpu:.0f}x')
