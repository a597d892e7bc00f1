#!/opt/conda/bin/python3.9
"""Golden vectors for the posterior estimator,
captured from the imported reference:

    /opt/conda/bin/python3.9 tests/golden/make_posterior_golden.py

Pinned functions (/root/reference/libs/utils.py): get_dist :90-97,
_get_MPEAR :100-130, _calc_MPEAR :133-145, get_mean_hierarchy_assignment
:148-192, get_latents_posterior / _get_latents_posterior_chain :195-244,
_concat_chain_results :206-223.
Inputs: two reference chains on a 60 x 40 synthetic (fixed errors)."""
import contextlib
import io
import os
import sys

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from libs.CRP import CRP  # noqa: E402
from libs.MCMC import MCMC  # noqa: E402
import libs.utils as ut  # noqa: E402
from make_golden import synth, encode  # noqa: E402

data = synth(3, 60, 40, 3, 0.1)
out = {'data': encode(data)}
results = []
for ci, seed in enumerate((7, 8)):
    model = CRP(data, DP_alpha=[-1, -1], param_beta=[.25, .25],
        FN_error=0.1, FP_error=0.001)
    mcmc = MCMC(model, sm_prob=.33, dpa_prob=.25, error_prob=0,
        sm_ratios=[.75, .25], sm_steps=3)
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((90, 30), seed, 1, 0, '', True)
    res = mcmc.get_results()[0]
    results.append(res)
    for key in ('assignments', 'params', 'DP_alpha', 'FN', 'FP', 'ML', 'MAP'):
        out[f'r{ci}_{key}'] = np.asarray(res[key])
    out[f'r{ci}_burn_in'] = np.array(res['burn_in'])

a0 = results[0]['assignments'][results[0]['burn_in']:]
out['dist0'] = ut.get_dist(a0)
out['mpear0'] = ut._get_MPEAR(a0)
sim = 1 - out['dist0']
out['mpear0_score'] = np.array(ut._calc_MPEAR(sim, out['mpear0']))

# (single_chains=True indexes params[burn_in:] of an already burn-in-free
# trace in the reference and raises IndexError: only the default path is pinned)
for tag, single in (('mean', False),):
    lat = ut.get_latents_posterior(results, data, single)
    out[f'{tag}_n'] = np.array(len(lat))
    for i, l in enumerate(lat):
        out[f'{tag}{i}_assignment'] = np.asarray(l['assignment'])
        out[f'{tag}{i}_genotypes'] = np.asarray(l['genotypes'].values,
            dtype=float)
        out[f'{tag}{i}_a'] = np.array(l['a'])
        out[f'{tag}{i}_FN'] = np.array(l['FN'])
        out[f'{tag}{i}_FP'] = np.array(l['FP'])
        out[f'{tag}{i}_FN_geno'] = np.array(l['FN_geno'])
        out[f'{tag}{i}_FP_geno'] = np.array(l['FP_geno'])
for est in ('ML', 'MAP'):
    l = ut.get_latents_point(results, est, data)[0]
    out[f'{est}_assignment'] = np.asarray(l['assignment'])
    out[f'{est}_step'] = np.array(l['step'])
    out[f'{est}_genotypes'] = np.asarray(l['genotypes'].values, dtype=float)
    out[f'{est}_FN_geno'] = np.array(l['FN_geno'])
out['psrf'] = np.array(ut.get_lugsail_batch_means_est(
    [(r['ML'], r['burn_in']) for r in results]))
np.savez_compressed(os.path.join(HERE, 'posterior.npz'), **out)
print({k: v.shape for k, v in out.items() if k.startswith(('mean', 'mpear'))})
print('psrf', out['psrf'], 'K', np.unique(out['mean0_assignment']).size)
