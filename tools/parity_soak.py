#!/usr/bin/env python3
"""Long parity run: GPU chain vs CPU oracle chain on the same seed, reports
the first step (if any) where the assignment trajectories part.
usage: parity_soak.py <config> <steps> [seed] [smp] [key=value ...]
keys: data=<seed of the synthetic matrix> beta=<p>,<q> (parameter prior)
learned=0|1 (error rates fixed / learned) sm_steps=<n> ratios=<split>,<merge>
alpha=<a>,<b> (Gamma prior of DP alpha; -1,-1: the reference's default)"""
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

opts = dict(a.split('=', 1) for a in sys.argv[3:] if '=' in a)
plain = [a for a in sys.argv[1:] if '=' not in a]
cfg, steps = plain[0], int(plain[1])
seed = int(plain[2]) if len(plain) > 2 else 42
smp = float(plain[3]) if len(plain) > 3 else .33
N, M, C, miss, learned = bench.CONFIGS[cfg]
learned = bool(int(opts.get('learned', learned)))
data = bench.synth(int(opts.get('data', 0)), N, M, C, miss)
beta = [float(x) for x in opts.get('beta', '.25,.25').split(',')]
alpha = [float(x) for x in opts.get('alpha', '-1,-1').split(',')]
ratios = [float(x) for x in opts.get('ratios', '.75,.25').split(',')]


def make_model(mod_fixed, mod_learn):
    # (bench.make_model with the priors of this run)
    if learned:
        return mod_learn.CRP_errors_learning(data, DP_alpha=alpha,
            param_beta=beta, FP_mean=0.01, FP_sd=0.01, FN_mean=0.2, FN_sd=0.1)
    return mod_fixed.CRP(data, DP_alpha=alpha, param_beta=beta, FN_error=0.1,
        FP_error=0.001)


res = []
for name, mods in (('gpu', (dev_fixed, dev_learn)), ('oracle', (O, O))):
    model = make_model(*mods)
    mcmc = MCMC(model, sm_prob=smp, dpa_prob=.25,
        error_prob=.25 if learned else 0., sm_ratios=ratios,
        sm_steps=int(opts.get('sm_steps', 3)))
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
extra = ' '.join(f'{k}={v}' for k, v in sorted(opts.items()))
print(f'{cfg} seed {seed} smp {smp} {extra}: first diverging step: {first} '
    f'(of {steps}); max rel ML diff over the common prefix: {rel:.2e}; '
    f'K at end gpu/oracle: {np.unique(g["assignments"][-1]).size}/'
    f'{np.unique(o["assignments"][-1]).size}')
