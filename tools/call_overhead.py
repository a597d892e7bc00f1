#!/usr/bin/env python3
"""Wall time per C-ABI call at the converged config-3 shape (dev tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import _lib  # noqa: E402

N, M, K = 5000, 1000, 12
data = bench.synth(0, N, M, 10, 0.2)
ctx = _lib.Context(data=data)
rng = np.random.RandomState(1)
theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
labels = rng.randint(0, K, N).astype(np.int64)
ids = np.arange(K, dtype=np.int64)
cells = np.arange(500, dtype=np.int64)


def t(name, fn, reps=300):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f'{name:34s} {1e6 * (time.perf_counter() - t0) / reps:8.1f} us',
        flush=True)


t('ll_theta_pinned N x 12', lambda: ctx.ll_theta_pinned(0, theta, .01, .2, K + 16))
t('ll_theta N x 12 (pageable out)', lambda: ctx.ll_theta(0, theta, .01, .2))
t('ll_theta N x 12 (no fetch)', lambda: (ctx.ll_theta(0, theta, .01, .2, fetch=False), ctx.sync()))
t('colcounts_by_label', lambda: ctx.colcounts_by_label(labels, ids))
t('colcounts 2 segments of 250', lambda: ctx.colcounts([cells[:250], cells[250:]]))
t('ll_total 1 trial', lambda: ctx.ll_total(theta, [.01], [.2]))
t('ll_total 2 trials', lambda: ctx.ll_total(theta, [.01, .02], [.2, .2]))
t('view_set 500 cells', lambda: ctx.view_set(1, cells))
t('ll_theta view 500 x 2', lambda: ctx.ll_theta(1, theta[:2], .01, .2))
t64 = theta[:2].astype(np.float64)
L1, L0 = np.log(t64 * .8 + (1 - t64) * .01), np.log(t64 * .2 + (1 - t64) * .99)
t('ll_tables view 500 x 2 (bit-exact)', lambda: ctx.ll_tables(1, L1, L0))
t('ll_tables N x 1 (bit-exact)', lambda: ctx.ll_tables(0, L1[:1], L0[:1]))
lab = rng.randint(0, 2, 500)
t('view_counts 500 cells, 2 segments', lambda: ctx.view_counts(1, lab, 2))
def kern(name):
    ctx.bench_ll(5)
    print(f'  device time, HIP events: {name:24s} '
        f'{1e3 * min(ctx.bench_ll(20) for _ in range(5)):8.1f} us',
        flush=True)


ctx.ll_tables(1, L1, L0)
kern('k_ll_seq 500 x 2')
ctx.ll_tables(0, L1[:1], L0[:1])
kern('k_ll_seq N x 1')
ctx.ll_theta(1, theta[:2], .01, .2)
kern('ll_theta 500 x 2')
ctx.ll_theta(0, theta, .01, .2)
kern('ll_theta N x 12')
t('sync only', ctx.sync)
ctx.close()
