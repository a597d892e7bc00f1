#!/usr/bin/env python3
"""Generate a full-size trajectory fixture with the CPU ORACLE (system python:
the stack the GPU box also runs):

    python tests/golden/make_c3_trajectory.py        # config 3, ~5 min, 1 core
    python tests/golden/make_c3_trajectory.py c4     # config 4, ~1 h, 1 core

config 3: 5000 cells x 1000 mutations, config 4: 10000 x 2000; 20 % missing,
learned errors, data seed 0, MCMC seed 42, CLI-default moves, 9 steps (step 1
= the sweep from K0 = 3152 / 6280 clusters).  Stored: assignments (int16),
ML/MAP/alpha/FN/FP traces -> <config>_trajectory.npz."""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import crp_numpy as O  # noqa: E402
from bnpc_amd.mcmc import MCMC  # noqa: E402

STEPS = 9       # the driver needs >= 9 steps (libs/MCMC.py:378)
CFG = sys.argv[1] if len(sys.argv) > 1 else 'c3'
N, M, C, miss, learned = bench.CONFIGS[CFG]
data = bench.synth(0, N, M, C, miss)
model = bench.make_model(O, O, data, learned)
mcmc = MCMC(model, error_prob=.25, **bench.MCMC_PARAMS)
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    mcmc.run((STEPS, 3), 42, 1, 0, '', True)
res = mcmc.get_results()[0]
print(f'{time.time() - t0:.0f}s; K per step',
    [len(np.unique(a)) for a in res['assignments']])
np.savez_compressed(os.path.join(HERE, f'{CFG}_trajectory.npz'),
    assignments=res['assignments'].astype(np.int16),
    ML=res['ML'], MAP=res['MAP'], DP_alpha=res['DP_alpha'], FN=res['FN'],
    FP=res['FP'], numpy=np.__version__)
