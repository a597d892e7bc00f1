#!/usr/bin/env python3
"""A/B on one box: the converged C3 chain (bench.py's timed window) with and
without the host-math shortcuts (native left-of-zero truncnorm mass, cached
prior density).  Same seed -> same trajectory in every leg; only time moves."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import fastdist as F, model as P  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

N, M, C, miss, learned = bench.CONFIGS['c3']
data = bench.synth(0, N, M, C, miss)
F.selfcheck()
print('selfcheck', F._state)


def leg(left, cache, steps=200, warm=10):
    F._state['left'] = left
    P.CRP._PRIOR_CACHE_ELEMS = cache
    np.random.seed(42)
    model = bench.make_model(dev_fixed, dev_learn, data, learned)
    model.init()
    chain = bench.new_chain(model, learned, steps + warm)
    for i in range(1, warm + 1):
        bench.step(chain, i, 0)
    t0 = time.perf_counter()
    for i in range(warm + 1, warm + steps + 1):
        bench.step(chain, i, 0)
    dt = time.perf_counter() - t0
    ml = chain.results['ML'][warm + steps]
    model.close()
    return 1e3 * dt / steps, ml


for rep in range(3):
    for left, cache in ((False, 0), (True, 0), (False, 1 << 22),
            (True, 1 << 22)):
        ms, ml = leg(left, cache)
        print(f'rep {rep} left={left!s:5} cache={cache:8d}: {ms:6.3f} ms/step '
            f'({1e3 / ms:6.1f} steps/s)  ML_end {ml:.6f}', flush=True)
