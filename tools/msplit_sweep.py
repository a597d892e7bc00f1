#!/usr/bin/env python3
"""Launch time of the mutation-split sums kernel by chunk count
(BNPC_MSPLIT = N; 1 = the library's own choice) at the converged-sweep
shapes: sums alone (bench_ll) and the whole evaluation (tables + sums +
combine, bench_ll_full), HIP events on the library's stream.
usage: msplit_sweep.py [N M K ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [
    (50000, 5000, 50), (50000, 5000, 54), (10000, 2000, 22),
    (5000, 1000, 200), (5000, 1000, 14)]
chunks = (1, 4, 8, 12, 16, 20, 24, 32, 48)     # 1 = the library's choice


print('| N | M | K | chunks asked | kernel | chunks used | sums us | '
    'evaluation us | % of the 2-add issue peak (sums) |')
print('|---|---|---|---|---|---|---|---|---|')
for N, M, K in shapes:
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    theta = np.clip(np.random.RandomState(1).uniform(size=(K, M)), 1e-5,
        1 - 1e-5).astype(np.float32)
    for ms in chunks:
        os.environ['BNPC_MSPLIT'] = str(ms)
        ctx.reload_options()
        ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
        ctx.sync()
        kern, _, used = ctx.last_launch()
        ctx.bench_ll(2)
        t = min(ctx.bench_ll(10) for _ in range(3))
        ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False)
        ctx.sync()
        ctx.bench_ll_full(2)
        tf = min(ctx.bench_ll_full(10) for _ in range(3))
        print(f'| {N} | {M} | {K} | {ms} | {kern} | {used} | {t * 1e3:.1f} | '
            f'{tf * 1e3:.1f} | {N * K * M / (t * 1e-3) / 19.65e12 * 100:.1f} |',
            flush=True)
    ctx.close()
del os.environ['BNPC_MSPLIT']
