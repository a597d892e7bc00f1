#!/usr/bin/env python3
"""Rate of the cells x clusters x mutations kernel at first-sweep TILE shapes:
few rows (cells of one tile), tens of thousands of clusters.  (dev tool)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import _lib  # noqa: E402

N, M, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 \
    else (50000, 5000, 31608)
data = bench.synth(0, N, M, 50, 0.2)
ctx = _lib.Context(data=data)
rng = np.random.RandomState(1)
theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
ctx.theta_put(0, theta)
rows_all = np.arange(K, dtype=np.int64)
for rows in (264, 512, 1024, 1060, 2048, 4096):
    ctx.view_set(1, rng.permutation(N)[:rows])
    ctx.ll_rows_pinned(1, rows_all, 0.01, 0.2, K + 48)
    ctx.bench_ll(1)
    ms = min(ctx.bench_ll(3) for _ in range(2))
    print(f'rows {rows:5d} x K {K} x M {M}: {ms:8.3f} ms  '
        f'{rows * K * M / (ms * 1e-3) / 1e12:7.2f} T elem-evals/s '
        f'({rows * K * M / (ms * 1e-3) / 19.65e12 * 100:5.1f}% of 2-add peak)',
        flush=True)
ctx.close()
