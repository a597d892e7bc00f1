#!/usr/bin/env python3
"""cell x cluster log-likelihood evals/s of the ll kernel at the shapes of
SURVEY.md section 8(d), GPU (kernel time, HIP events) next to the CPU oracle's
_calc_ll on a bounded sample of cells (1 core)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402
import bench  # noqa: E402

SHAPES = [(1000, 200, (10, 632)), (5000, 1000, (10, 64, 3152)),
    (10000, 2000, (20, 6300)), (50000, 5000, (50, 512))]
print('| N | M | K | kernel | launch ms | cell x cluster evals/s | '
    'element-evals/s | % of FP64 2-add peak | alg. GB/s | CPU evals/s '
    '(1 core) | GPU/CPU |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for N, M, Ks in SHAPES:
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    om = O.CRP(data, [-1, -1], [1, 1], FN_error=0.2, FP_error=0.01)
    for K in Ks:
        theta = np.clip(np.random.RandomState(1).uniform(size=(K, M)), 1e-5,
            1 - 1e-5).astype(np.float32)
        ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
        ctx.sync()
        ctx.bench_ll(2)
        ms = min(ctx.bench_ll(5) for _ in range(3))
        ev = N * K / (ms * 1e-3)
        B = N * M / 4 + 4 * K * M + 8 * N * K
        # CPU: a few cells, bounded to ~1 s
        ncell = max(1, min(N, int(2e7 / (K * M))))
        t0 = time.perf_counter()
        for r in range(ncell):
            om._calc_ll(data[[r]], theta)
        cpu = ncell * K / (time.perf_counter() - t0)
        kern = ctx.last_launch()[0]
        print(f'| {N} | {M} | {K} | {kern} | {ms:.4f} | {ev:.3e} | '
            f'{ev * M:.3e} | {ev * M / 19.65e12 * 100:.1f} | '
            f'{B / ms / 1e6:.1f} | {cpu:.3e} | {ev / cpu:.0f}x |')
    ctx.close()
