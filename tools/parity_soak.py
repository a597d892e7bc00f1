#!/usr/bin/env python3
"""Long parity run: GPU chain vs CPU oracle chain on the same seed, reports
the first step (if any) where the assignment trajectories part.
usage: parity_soak.py <config> <steps> [seed] [smp]"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import libs.CRP as dev_fixed  # noqa: E402
import libs.CRP_learning_errors as dev_learn  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402
from bnpc_amd.mcmc import MCMC  # noqa: E402

cfg, steps = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 42
smp = float(sys.argv[4]) if len(sys.argv) > 4 else .33
N, M, C, miss, learned = bench.CONFIGS[cfg]
data = bench.synth(0, N, M, C, miss)
res = []
for name, mods in (('gpu', (dev_fixed, dev_learn)), ('oracle', (O, O))):
    model = bench.make_model(mods[0], mods[1], data, learned)
    mcmc = MCMC(model, sm_prob=smp, dpa_prob=.25,
        error_prob=.25 if learned else 0., sm_ratios=[.75, .25], sm_steps=3)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((steps, int(steps * .33)), seed, 1, 0, '', True)
    res.append(mcmc.get_results()[0])
    print(f'{name}: {steps} steps in {time.time() - t0:.1f}s', flush=True)
g, o = res
same = (g['assignments'] == o['assignments']).all(axis=1)
first = int(np.argmin(same)) if not same.all() else -1
n = steps + 1 if first < 0 else first
rel = np.max(np.abs(g['ML'][:n] / o['ML'][:n] - 1))
print(f'{cfg} seed {seed} smp {smp}: first diverging step: {first} '
    f'(of {steps}); max rel ML diff over the common prefix: {rel:.2e}; '
    f'K at end gpu/oracle: {np.unique(g["assignments"][-1]).size}/'
    f'{np.unique(o["assignments"][-1]).size}')
