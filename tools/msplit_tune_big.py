#!/usr/bin/env python3
"""Mutation split of the mid-size likelihood launches of configs 4 and 5
(converged K: too few waves per SIMD to hide the table loads): device time of
the whole evaluation against BNPC_MSPLIT_WAVES (dev tool)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

rng = np.random.RandomState(1)
for N, M, Ks in ((10000, 2000, (12, 22, 40)), (50000, 5000, (10, 30, 53, 100)),
        (5000, 1000, (14, 40))):
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    for K in Ks:
        theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
            .astype(np.float32)
        for waves in (0, 1, 4096, 8192, 16384, 32768):     # 0 = the default rule
            os.environ['BNPC_MSPLIT_WAVES'] = str(waves)
            ctx.reload_options()
            ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
            ctx.sync()
            name, K_, ms = ctx.last_launch()
            ctx.bench_ll_full(3)
            t = min(ctx.bench_ll_full(10) for _ in range(3))
            evals = N * M * K / (t * 1e-3)
            print(f'{N:6d} x {M} K={K:3d} target_waves={waves:6d}: '
                f'{t * 1e3:8.1f} us  chunks {ms:3d}  {name:40s} '
                f'{100 * evals / 19.65e12:5.1f} % of peak', flush=True)
    ctx.close()
