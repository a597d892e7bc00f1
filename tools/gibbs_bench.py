#!/usr/bin/env python3
"""Parts of a converged Gibbs sweep at config 3: the device call, the native
loop (with its Python preparation), the rest.  (dev tool, needs the GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import _lib  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
chain = bench.new_chain(model, learned, 400)
for i in range(1, 11):
    bench.step(chain, i, 0)

acc = {}


def timed(obj, name, key):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, inner)


timed(model._dev(), 'll_theta_pinned', 'device call')
timed(model, '_gibbs_window', 'window (prep + native loop)')
lib = _lib.load()
real = lib.bnpc_gibbs_sweep


class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter()
        r = real(*a)
        acc['native loop'] = acc.get('native loop', 0.0) \
            + time.perf_counter() - t0
        return r


lib.bnpc_gibbs_sweep = Wrap()
reps = 100
model._hint_used = 0
t0 = time.perf_counter()
for _ in range(reps):
    model.update_assignments_Gibbs()
total = time.perf_counter() - t0
print(f'cells decided from the hint: {getattr(model, "_hint_used", 0) / reps:.0f} of {N} per sweep')
print(f'{cfg} K={len(model.cells_per_cluster)}: sweep '
    f'{1e6 * total / reps:7.1f} us | ' + ' | '.join(
        f'{k} {1e6 * v / reps:7.1f}' for k, v in acc.items()))
