#!/usr/bin/env python3
"""L2 prefetch of the mask / table streams in k_ll8_asm (BNPC_LL_PREFETCH):
device time of the evaluation with it off and on, at shapes whose streams do
and do not fit one XCD's L2; sums must be identical (dev tool)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

rng = np.random.RandomState(1)
shapes = ((5000, 1000, None, (14, 3152)), (10000, 2000, None, (22, 6300)),
    (50000, 5000, None, (53, 100, 512)), (50000, 5000, 1024, (8000, 31608)))
for N, M, rows, Ks in shapes:
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    view = 0
    if rows:
        ctx.view_set(1, rng.permutation(N)[:rows])
        view = 1
    for K in Ks:
        theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
            .astype(np.float32)
        outs = []
        line = f'{rows or N:6d} cells x {M} x K={K:5d}:'
        for pf in ('0', '1', '-1'):
            os.environ['BNPC_LL_PREFETCH'] = pf
            ctx.reload_options()
            small = (rows or N) * K * 8 < (64 << 20)
            out = ctx.ll_theta(view, theta, 0.01, 0.2, fetch=small)
            ctx.sync()
            if small:
                outs.append(out.copy())
            name, _, ms = ctx.last_launch()
            ctx.bench_ll(2)
            t = min(ctx.bench_ll(5) for _ in range(3))
            frac = (rows or N) * M * K / (t * 1e-3) / 19.65e12
            line += f'  pf={pf:>2s} {t * 1e3:9.1f} us ({100 * frac:4.1f} %)'
        same = all(np.array_equal(outs[0], o) for o in outs[1:])
        print(line + f'  {name}  identical sums: {same if outs else "n/a"}',
            flush=True)
    ctx.close()
