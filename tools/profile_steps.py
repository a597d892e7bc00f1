#!/usr/bin/env python3
"""Where does a converged MCMC step spend its host time?  (dev tool)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
chain = bench.new_chain(model, learned, steps + 5)
for i in range(1, 6):
    bench.step(chain, i, 0)

names = ['update_assignments_Gibbs', 'update_assignments_split_merge',
    'update_parameters', 'update_error_rates', 'update_DP_alpha',
    'get_ll_full', 'get_lprior_full']
acc = {n: [0.0, 0] for n in names}
for n in names:
    if not hasattr(model, n):
        continue
    fn = getattr(model, n)

    def wrap(fn=fn, n=n):
        def inner(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[n][0] += time.perf_counter() - t0
            acc[n][1] += 1
            return r
        return inner
    setattr(model, n, wrap())

# wall time inside every entry point of the library (ctypes drops into C:
# cProfile books that time on the calling Python function)
from bnpc_amd import _lib  # noqa: E402
lib = _lib.load()
c_acc = {}
for name in _lib.SIGNATURES:
    if name == 'bnpc_matrix_wait':      # its address goes to the C loop
        continue
    fn = getattr(lib, name)

    def timed(*a, _fn=fn, _n=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        e = c_acc.setdefault(_n, [0.0, 0])
        e[0] += time.perf_counter() - t0
        e[1] += 1
        return r
    setattr(lib, name, timed)

pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(6, steps + 6):
    bench.step(chain, i, 0)
pr.disable()
el = time.perf_counter() - t0
print(f'{steps} steps in {el:.3f}s = {1e3 * el / steps:.2f} ms/step '
    f'(profiler on)')
for n, (t, c) in acc.items():
    if c:
        print(f'  {n:34s} calls {c:4d}  total {1e3 * t:8.1f} ms  '
            f'per call {1e3 * t / c:7.2f} ms  per step {1e3 * t / steps:6.2f}')
print('  -- inside the library, per entry point --')
tot = 0.0
for n, (t, c) in sorted(c_acc.items(), key=lambda kv: -kv[1][0]):
    tot += t
    print(f'  {n:34s} calls {c:5d}  total {1e3 * t:8.1f} ms  '
        f'per call {1e6 * t / c:7.1f} us  per step {1e3 * t / steps:6.3f}')
print(f'  library total {1e3 * tot / steps:.3f} ms/step, everything else '
    f'{1e3 * (el - tot) / steps:.3f} ms/step')
if len(sys.argv) > 3:
    pr.dump_stats(sys.argv[3])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
