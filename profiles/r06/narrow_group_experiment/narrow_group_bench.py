#!/usr/bin/env python3
"""The sums launch of a converged sweep at config 5's size with and without
the narrow last cluster group (round 6): K = 49 ... 60
at 50000 x 5000, kernel time by HIP events on the library's stream (the
narrow kernel runs beside the groups of 8 on a second stream, joined before
the combine pass: the events bracket both), best of 3 x 10 launches; bits
compared.  BNPC_KW=16 forces the narrow path, BNPC_KW=8 the groups of 8.
usage: narrow_group_bench.py [N] [M] [K ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Ks = [int(a) for a in sys.argv[3:]] or [49, 50, 51, 52, 54, 56, 57, 58, 59, 60]
data = bench.synth(0, N, M, 50, 0.2)
ctx = _lib.Context(data=data)
print('| K | groups of 8: launch ms | % of the 2-add issue peak | with the '
    'narrow last group: launch ms | % | kernels | same bits |')
print('|---|---|---|---|---|---|---|')
for K in Ks:
    theta = np.clip(np.random.RandomState(1).uniform(size=(K, M)), 1e-5,
        1 - 1e-5).astype(np.float32)
    res = {}
    for mode in ('8', '16'):
        os.environ['BNPC_KW'] = mode
        ctx.reload_options()
        out = ctx.ll_theta(0, theta, 0.01, 0.2)
        ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
        ctx.sync()
        ctx.bench_ll(3)
        ms = min(ctx.bench_ll(10) for _ in range(3))
        res[mode] = (ms, ctx.last_launch()[0], out)
    pct = lambda ms: 2 * N * K * M / (ms * 1e-3) / 39.3e12 * 100
    a, b = res['8'], res['16']
    print(f'| {K} | {a[0]:.4f} | {pct(a[0]):.1f} | {b[0]:.4f} | '
        f'{pct(b[0]):.1f} | {b[1]} | {np.array_equal(a[2], b[2])} |')
ctx.close()
