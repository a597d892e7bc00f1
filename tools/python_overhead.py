#!/usr/bin/env python3
"""How much of a converged step is the interpreter?  Per move: wall time of the
model method, wall time inside the library's entry points called under it, the
difference.  No cProfile (it inflates exactly what is measured here)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402
from bnpc_amd import _lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
np.random.seed(42)
model = bench.make_model(dev_fixed, dev_learn, data, learned)
model.init()
chain = bench.new_chain(model, learned, steps + 10, cfg)
for i in range(1, 11):
    bench.step(chain, i, 0)

phase = ['other']
lib_in = {}
lib = _lib.load()
for name in _lib.SIGNATURES:
    if name == 'bnpc_matrix_wait':
        continue
    fn = getattr(lib, name)

    def timed(*a, _fn=fn):
        t0 = time.perf_counter()
        r = _fn(*a)
        lib_in[phase[0]] = lib_in.get(phase[0], 0.0) \
            + time.perf_counter() - t0
        return r
    setattr(lib, name, timed)

wall = {}
# the default path makes a whole step as ONE native call (bnpc_chain_step):
# wrapping the model's methods would switch that off - they are only wrapped
# for the method-by-method path (BNPC_NATIVE_STEP=0)
by_method = os.environ.get('BNPC_NATIVE_STEP', '1') == '0'
for name in ('update_assignments_Gibbs', 'update_assignments_split_merge',
        'update_parameters', 'update_error_rates', 'update_DP_alpha',
        'get_lprior_full', 'get_ll_full_deferred'):
    if not by_method:
        break
    fn = getattr(model, name)

    def wrap(*a, _fn=fn, _n=name, **k):
        prev = phase[0]
        phase[0] = _n
        t0 = time.perf_counter()
        r = _fn(*a, **k)
        e = wall.setdefault(_n, [0.0, 0])
        e[0] += time.perf_counter() - t0
        e[1] += 1
        phase[0] = prev
        return r
    setattr(model, name, wrap)

t0 = time.perf_counter()
for i in range(11, steps + 11):
    bench.step(chain, i, 0)
el = time.perf_counter() - t0
print(f'{steps} steps, {1e3 * el / steps:.3f} ms/step (timers on)')
tot_w = tot_l = 0.0
for n, (t, c) in sorted(wall.items(), key=lambda kv: -kv[1][0]):
    li = lib_in.get(n, 0.0)
    tot_w += t
    tot_l += li
    print(f'  {n:32s} calls {c:4d}  {1e6 * t / c:7.1f} us per call: library '
        f'{1e6 * li / c:7.1f}, interpreter {1e6 * (t - li) / c:7.1f}  '
        f'| per step: {1e3 * t / steps:.3f} = {1e3 * li / steps:.3f} + '
        f'{1e3 * (t - li) / steps:.3f}')
if not by_method:
    inside = sum(lib_in.values())
    print(f'  whole steps as one native call: {model.host_stats()["native_steps"]} '
        f'of {steps + 10} steps so far')
    print(f'  step = library {1e3 * inside / steps:.3f} + interpreter '
        f'{1e3 * (el - inside) / steps:.3f} ms (binding: state in / out, '
        'tallies, trace bookkeeping)')
    sys.exit(0)
rest = el - tot_w
print(f'  outside the moves (driver, traces, pick-up of the total): '
    f'{1e3 * rest / steps:.3f} ms/step of which library '
    f'{1e3 * lib_in.get("other", 0.0) / steps:.3f}')
print(f'  step = library {1e3 * (tot_l + lib_in.get("other", 0)) / steps:.3f}'
    f' + interpreter {1e3 * (el - tot_l - lib_in.get("other", 0)) / steps:.3f}')
