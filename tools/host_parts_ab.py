#!/usr/bin/env python3
"""A/B: elements per host thread for the MH proposal batches at converged
config 3 (K x M = 10..18 x 1000).  Same seed -> same trajectory every leg."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bnpc_amd import model as P  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

N, M, C, miss, learned = bench.CONFIGS['c3']
data = bench.synth(0, N, M, C, miss)


def leg(min_elems, per_thread, steps=200, warm=10):
    P._THREAD_MIN_ELEMS = min_elems
    P._ELEMS_PER_THREAD = per_thread
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


for rep in range(2):
    for min_elems, per_thread in ((1 << 30, 8192), (1 << 14, 8192),
            (1 << 13, 4096), (1 << 12, 4096), (1 << 12, 2048)):
        ms, ml = leg(min_elems, per_thread)
        print(f'rep {rep} min {min_elems:>10d} per-thread {per_thread:5d}: '
            f'{ms:6.3f} ms/step ({1e3 / ms:6.1f} steps/s) ML {ml:.4f}',
            flush=True)
