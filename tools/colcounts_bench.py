#!/usr/bin/env python3
"""Time bnpc_colcounts_by_label / bnpc_colcounts end to end (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bnpc_amd import _lib
import bench
for N, M, K in ((5000, 1000, 10), (10000, 2000, 20), (50000, 5000, 50), (5000, 1000, 3152)):
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    assign = np.random.RandomState(0).randint(0, K, N)
    ids = np.unique(assign)
    ref1 = np.stack([(data[assign == i] == 1).sum(axis=0) for i in ids[:3]])
    n1, n0 = ctx.colcounts_by_label(assign, ids)
    assert np.array_equal(n1[:3], ref1)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.colcounts_by_label(assign, ids)
    print(f'N={N} M={M} K={ids.size}: by_label {1e6 * (time.perf_counter() - t0) / 20:.0f} us per call (host sort + launch + D2H)')
    ctx.close()
