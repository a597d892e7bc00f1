#!/usr/bin/env python3
"""Sweep the mutation-split parameters of small ll launches (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bnpc_amd import _lib
import bench
data = bench.synth(0, 5000, 1000, 10, 0.2)
ctx = _lib.Context(data=data)
rng = np.random.RandomState(1)
for n, K in ((500, 2), (5000, 10), (5000, 18), (5000, 64)):
    view = 0
    if n != 5000:
        ctx.view_set(1, rng.permutation(5000)[:n]); view = 1
    theta = np.clip(rng.uniform(size=(K, 1000)), 1e-5, 1 - 1e-5).astype(np.float32)
    for waves, cap in ((1, 1), (1024, 16), (2048, 32), (4096, 64), (8192, 64), (8192, 128), (16384, 128)):
        os.environ['BNPC_MSPLIT_WAVES'] = str(waves); os.environ['BNPC_MSPLIT_MAX'] = str(cap)
        ctx.reload_options()
        ctx.ll_theta(view, theta, 0.01, 0.2, fetch=False); ctx.sync(); ctx.bench_ll(3)
        t = min(ctx.bench_ll(10) for _ in range(3))
        print(f'n={n:5d} K={K:3d} target_waves={waves:6d} cap={cap:4d}: {t*1e3:7.1f} us')
