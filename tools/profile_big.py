#!/usr/bin/env python3
"""cProfile of converged steps at a large config (dev tool).
usage: profile_big.py N M C steps [smp sms]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402
from bnpc_amd.mcmc import Chain_steps  # noqa: E402

N, M, C, steps = (int(a) for a in sys.argv[1:5])
smp = float(sys.argv[5]) if len(sys.argv) > 5 else .33
sms = int(sys.argv[6]) if len(sys.argv) > 6 else 3
data = bench.synth(0, N, M, C, 0.2)
np.random.seed(42)
model = bench.make_model(None, dev_learn, data, True)
model.init()
params = dict(sm_prob=smp, dpa_prob=.25, error_prob=.25, sm_ratios=[.75, .25],
    sm_steps=sms, param_proposal_sd=np.array([0.1, 0.25, 0.5]))
chain = Chain_steps(model, 1, steps + 6, 0, params, 0, False)
for i in range(1, 7):
    bench.step(chain, i, 0)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(7, steps + 7):
    bench.step(chain, i, 0)
pr.disable()
print(f'{steps} steps: {1e3 * (time.perf_counter() - t0) / steps:.1f} ms/step '
    f'(profiler on), K={len(model.cells_per_cluster)}')
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
