#!/usr/bin/env python3
"""Run the first steps of a large BASELINE config on the GPU and time them
(dev tool; parity at these sizes is covered by the property tests)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402

N, M, C, miss = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]),
    float(sys.argv[4]))
steps = int(sys.argv[5])
smp = float(sys.argv[6]) if len(sys.argv) > 6 else .33
sms = int(sys.argv[7]) if len(sys.argv) > 7 else 3
t0 = time.perf_counter()
data = bench.synth(0, N, M, C, miss)
print(f'synth {N}x{M}: {time.perf_counter() - t0:.1f}s', flush=True)
np.random.seed(42)
model = bench.make_model(None, dev_learn, data, True)
t0 = time.perf_counter()
model.init()
print(f'init K0={len(model.cells_per_cluster)}: '
    f'{time.perf_counter() - t0:.1f}s', flush=True)
from bnpc_amd.mcmc import Chain_steps  # noqa: E402
params = dict(sm_prob=smp, dpa_prob=.25, error_prob=.25,
    sm_ratios=[.75, .25], sm_steps=sms,
    param_proposal_sd=np.array([0.1, 0.25, 0.5]))
t0 = time.perf_counter()
chain = Chain_steps(model, 1, steps, 0, params, 0, False)
print(f'chain init (context, ML[0]={chain.results["ML"][0]:.1f}): '
    f'{time.perf_counter() - t0:.1f}s', flush=True)
for i in range(1, steps + 1):
    t0 = time.perf_counter()
    bench.step(chain, i, steps)
    print(f'step {i}: {time.perf_counter() - t0:8.3f}s  K='
        f'{len(model.cells_per_cluster):6d}  ML={chain.results["ML"][i]:.1f} '
        f'FN={model.FN:.4f} FP={model.FP:.5f}', flush=True)
