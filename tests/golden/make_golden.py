#!/opt/conda/bin/python3.9
"""Generate the golden vectors that pin the oracle (and through it the HIP path).

Run ONLY in the build container, with the interpreter that can import the
unmodified reference (bottleneck + seaborn present):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

It imports cbg-ethz/BnpC from /root/reference (read-only), drives the
reference's own functions on seeded inputs and stores inputs + outputs as
small .npz files next to this script.  Nothing of the reference's source is
stored: only numbers.  The reference never travels to the GPU box; these
fixtures do.

Functions pinned (reference file:line):
  CRP._calc_ll                       libs/CRP.py:197-204
  CRP.get_lpost_single_new_cluster   libs/CRP.py:230-234
  CRP._get_log_A                     libs/CRP.py:347-383
  CRP.get_ll_full / get_lprior_full  libs/CRP.py:237-251
  CRP_errors_learning.get_ll_full_error  libs/CRP_learning_errors.py:58-63
  CRP._rg_get_ll / _rg_init_split / _get_ll_ratio  libs/CRP.py:635-638, 547-567, 716-733
  CRP._normalize_log_probs / _normalize_log / init_DP_prior  libs/CRP.py:88-116, 191-194
  legacy np.random stream                (Appendix B of SURVEY.md)
  whole-chain traces via MCMC.run(debug=True)  libs/MCMC.py:79-135, 320-388
"""
import hashlib
import io
import json
import os
import sys
import contextlib

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import scipy  # noqa: E402
import bottleneck  # noqa: E402
from libs.CRP import CRP  # noqa: E402
from libs.CRP_learning_errors import CRP_errors_learning  # noqa: E402
from libs.MCMC import MCMC  # noqa: E402
import libs.dpmmIO as ref_io  # noqa: E402


def encode(data):
    """float64 {0,1,NaN} -> int8 {0,1,3} (the on-disk code of the reference)."""
    out = np.where(np.isnan(data), 3, data).astype(np.int8)
    return out


def synth(seed, N, M, C, miss, FP_true=0.001, FN_true=0.1):
    """SURVEY.md section 8(d) generator."""
    rng = np.random.RandomState(seed)
    geno = (rng.random_sample((C, M)) < 0.3)
    z = rng.randint(0, C, N)
    X = geno[z]
    u = rng.random_sample((N, M))
    obs = np.where(X == 1, u >= FN_true, u < FP_true).astype(np.float64)
    obs[rng.random_sample((N, M)) < miss] = np.nan
    return obs


def rand_data(rng, N, M, miss, p1=0.3):
    d = (rng.random_sample((N, M)) < p1).astype(np.float64)
    d[rng.random_sample((N, M)) < miss] = np.nan
    return d


def rand_theta(rng, K, M):
    t = rng.uniform(size=(K, M))
    t = np.clip(t, 1e-5, 1 - 1e-5).astype(np.float32)
    # force the clip boundaries to appear
    t[0, 0] = np.float32(1e-5)
    t[-1, -1] = np.float32(1 - 1e-5)
    return t


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} kB')


# ------------------------------------------------------------------ _calc_ll
def gold_calc_ll():
    rng = np.random.RandomState(101)
    out = {}
    cases = []
    specs = [
        # N, M, miss, K, FP, FN
        (64, 1003, 0.2, 9, 1e-3, 0.1),
        (64, 1003, 0.0, 2, 0.0123, 0.137),
        (257, 130, 0.1, 64, 1e-3, 0.1),
        (257, 130, 0.5, 1, 0.0123, 0.137),
        (33, 64, 0.2, 5, np.finfo(np.float64).resolution,
            np.finfo(np.float64).resolution),
        (5, 1, 0.2, 3, 1e-3, 0.1),
        (130, 200, 0.1, 17, 0.01, 0.2),
    ]
    for ci, (N, M, miss, K, FP, FN) in enumerate(specs):
        data = rand_data(rng, N, M, miss)
        if N > 10 and M > 10:
            data[3, :] = np.nan      # all-missing cell
            data[:, 7] = np.nan      # all-missing mutation
            data[5, :] = 1           # all ones
            data[6, :] = 0           # all zeros
        theta = rand_theta(rng, K, M)
        model = CRP(data, DP_alpha=[-1, -1], param_beta=[1, 1],
            FN_error=FN, FP_error=FP)
        ll = np.empty((N, K))
        for r in range(N):
            ll[r] = model._calc_ll(data[[r]], theta)
        assign = rng.randint(0, K, N)
        flat = model._calc_ll(data, theta[assign], True)
        pre = f'c{ci}_'
        out[pre + 'data'] = encode(data)
        out[pre + 'theta'] = theta
        out[pre + 'FPFN'] = np.array([FP, FN])
        out[pre + 'll'] = ll
        out[pre + 'assign'] = assign
        out[pre + 'flat'] = np.array(flat)
        cases.append(ci)
    out['n_cases'] = np.array(len(cases))
    save('calc_ll.npz', **out)


# ------------------------------------------- model-state dependent functions
def make_model(kind, data, pb, seed, FP=1e-3, FN=0.1):
    if kind == 'fixed':
        m = CRP(data, DP_alpha=[-1, -1], param_beta=pb, FN_error=FN,
            FP_error=FP)
    else:
        m = CRP_errors_learning(data, DP_alpha=[-1, -1], param_beta=pb,
            FP_mean=0.01, FP_sd=0.01, FN_mean=0.2, FN_sd=0.1)
    np.random.seed(seed)
    m.init()
    return m


def gold_state_functions():
    rng = np.random.RandomState(202)
    out = {}
    data = synth(5, 120, 90, 4, 0.15)
    data[2, :] = np.nan
    out['data'] = encode(data)
    pbs = [[1, 1], [.25, .25], [.75, 2]]
    out['param_betas'] = np.array(pbs, dtype=float)
    for pi, pb in enumerate(pbs):
        m = make_model('fixed', data, pb, 7 + pi)
        pre = f'p{pi}_'
        out[pre + 'assignment0'] = m.assignment.copy()
        out[pre + 'parameters0'] = m.parameters.copy()
        out[pre + 'CRP_prior0'] = m.CRP_prior.copy()
        out[pre + 'DP_a0'] = np.array(m.DP_a)
        out[pre + 'beta_mix'] = m._beta_mix_const.copy()
        out[pre + 'new_cluster'] = m.get_lpost_single_new_cluster()
        out[pre + 'll_full0'] = np.array(m.get_ll_full())
        out[pre + 'lprior0'] = np.array(m.get_lprior_full())
        # a few Gibbs sweeps + parameter updates to get a non-trivial state
        for _ in range(3):
            m.update_assignments_Gibbs()
            m.update_parameters()
        out[pre + 'assignment1'] = m.assignment.copy()
        cl = np.fromiter(m.cells_per_cluster.keys(), dtype=int)
        out[pre + 'clusters1'] = cl
        out[pre + 'sizes1'] = np.fromiter(m.cells_per_cluster.values(),
            dtype=int)
        out[pre + 'parameters1'] = m.parameters[cl].copy()
        out[pre + 'll_full1'] = np.array(m.get_ll_full())
        out[pre + 'lprior1'] = np.array(m.get_lprior_full())
        out[pre + 'lpost_single'] = np.stack([
            m.get_lpost_single(c, cl) for c in range(0, data.shape[0], 7)])

        # _get_log_A on each cluster with a recorded proposal
        for j, c in enumerate(cl[:4]):
            cells = np.argwhere(m.assignment == c).flatten()
            old = m.parameters[c].copy()
            std = rng.choice(m.param_proposal_sd, size=data.shape[1])
            new = np.clip(old + rng.normal(size=old.size) * std * 0.3,
                1e-5, 1 - 1e-5).astype(np.float32)
            a = (1e-5 - old) / std
            b = (1 - 1e-5 - old) / std
            for clip in (False, True):
                A = m._get_log_A(new, old, cells, a, b, std, clip)
                out[pre + f'logA{j}_A{int(clip)}'] = A
            out[pre + f'logA{j}_cells'] = cells
            out[pre + f'logA{j}_new'] = new
            out[pre + f'logA{j}_old'] = old
            out[pre + f'logA{j}_std'] = std
            # the likelihood part alone (independent of SciPy)
            x = data[cells]
            import bottleneck as bn
            llFN = m._Bernoulli_FN(x)
            llFP = m._Bernoulli_FP(x)
            out[pre + f'logA{j}_newll'] = bn.nansum(
                np.log(new * llFN + (1 - new) * llFP), axis=0)
            out[pre + f'logA{j}_oldll'] = bn.nansum(
                np.log(old * llFN + (1 - old) * llFP), axis=0)
        out[pre + 'n_logA'] = np.array(min(4, cl.size))

        # restricted-Gibbs primitives on the largest cluster (split-like)
        big = cl[np.argmax(out[pre + 'sizes1'])]
        cells = np.argwhere(m.assignment == big).flatten()
        rng.shuffle(cells)
        np.random.seed(99 + pi)
        m._rg_init_split(cells)
        out[pre + 'rg_cells'] = cells
        out[pre + 'rg_assignment_init'] = np.array(m.rg_assignment,
            dtype=float)
        out[pre + 'rg_params_split'] = m.rg_params_split.copy()
        m.rg_params_merge = m._init_cl_params_new(cells)
        out[pre + 'rg_params_merge'] = m.rg_params_merge.copy()
        out[pre + 'rg_ll'] = m._rg_get_ll(cells[1:-1], m.rg_params_split)
        i, j = cells[0], cells[-1]
        S = cells[1:-1]
        out[pre + 'rg_init_ll_i'] = m._calc_ll(data[S],
            np.nan_to_num(data[i], nan=m._beta_mix_const[0]))
        out[pre + 'rg_init_ll_j'] = m._calc_ll(data[S],
            np.nan_to_num(data[j], nan=m._beta_mix_const[0]))
        out[pre + 'rg_ll_ratio_split'] = np.array(
            m._get_ll_ratio(cells, 'split'))
        out[pre + 'rg_ll_ratio_merge'] = np.array(
            m._get_ll_ratio(cells, 'merge'))
        out[pre + 'rg_lprior_split'] = np.array(
            m._get_lprior_ratio_split(cells))
        out[pre + 'rg_lprior_merge'] = np.array(
            m._get_lprior_ratio_merge(cells))

    # learned-error model: get_ll_full_error
    m = make_model('learn', data, [.25, .25], 11)
    for _ in range(2):
        m.update_assignments_Gibbs()
        m.update_parameters()
    cl = np.fromiter(m.cells_per_cluster.keys(), dtype=int)
    out['le_assignment'] = m.assignment.copy()
    out['le_clusters'] = cl
    out['le_parameters'] = m.parameters[cl].copy()
    trials = np.array([[0.01, 0.2], [0.0123, 0.2], [0.01, 0.137],
        [1e-6, 0.9], [0.3, 1e-4]])
    out['le_trials'] = trials
    out['le_ll'] = np.array([m.get_ll_full_error(fp, fn)
        for fp, fn in trials])
    out['le_lprior'] = np.array(m.get_lprior_full())
    out['le_DP_a'] = np.array(m.DP_a)
    out['le_sizes'] = np.fromiter(m.cells_per_cluster.values(), dtype=int)
    save('state_functions.npz', **out)


# --------------------------------------------------------------- normalisers
def gold_normalisers():
    rng = np.random.RandomState(303)
    out = {}
    vecs = [
        rng.normal(size=7) * 3 - 50,
        np.array([-1e4, -3.0, -2.5, -9000.0, -2.5]),
        np.array([-800.0, -100.0]),
        np.array([-5.0, -5.0, -5.0]),
        rng.normal(size=200) * 30 - 1000,
        np.array([-12.5]),
    ]
    for i, v in enumerate(vecs):
        out[f'v{i}'] = v
        out[f'np{i}'] = CRP._normalize_log_probs(v)
    two = [np.array([-3.0, -4.0]), np.array([-1e4, 0.0]),
        np.array([-700.0, -701.5]), np.array([-2.0, -2.0]),
        np.array([0.0, -1e5])]
    for i, v in enumerate(two):
        out[f't{i}'] = v
        out[f'nl{i}'] = np.asarray(CRP._normalize_log(v), dtype=float)
    out['n_vecs'] = np.array(len(vecs))
    out['n_two'] = np.array(len(two))
    for N, a in ((10, 3.3), (1000, 31.6227766)):
        cl_vals = np.append(np.arange(1, N + 1), a)
        out[f'crp_prior_{N}'] = np.append(0, CRP.log_CRP_prior(cl_vals, N, a))
        out[f'crp_a_{N}'] = np.array(a)
    save('normalisers.npz', **out)


# ----------------------------------------------------------------------- RNG
def gold_rng():
    out = {}
    for seed in (1, 42, 1608637542):
        np.random.seed(seed)
        pre = f's{seed}_'
        out[pre + 'random'] = np.random.random(5)
        out[pre + 'perm'] = np.random.permutation(37)
        out[pre + 'randint'] = np.random.randint(0, 1000, size=11)
        out[pre + 'randint_big'] = np.random.randint(0, 2 ** 32 - 1, 3)
        p = np.array([.1, .2, .3, .25, .15])
        out[pre + 'choice_p'] = np.array(
            [np.random.choice(np.arange(5), p=p) for _ in range(20)])
        out[pre + 'choice_sd'] = np.random.choice(
            np.array([.1, .25, .5]), size=50)
        out[pre + 'choice2'] = np.random.choice(17, size=2, replace=False)
        out[pre + 'perm2'] = np.random.permutation(1000)
        out[pre + 'random_after'] = np.random.random(3)
        out[pre + 'beta'] = np.random.beta(
            np.array([.25, 1.25, 7.25, .25]), np.array([.25, .25, 3.25, 9.25]))
        out[pre + 'gamma'] = np.array([np.random.gamma(3.5, 0.7),
            np.random.gamma(0.4, 2.0)])
        out[pre + 'random_end'] = np.random.random(2)
    save('rng.npz', **out)


def gold_rng_beta():
    """Legacy Beta / Gamma sampler of the global stream (the draws behind
    CRP._init_cl_params_new, libs/CRP.py:183-188, _init_cl_params :155-180 and
    update_DP_alpha :386-410): array arguments over every branch - Johnk
    (both shapes <= 1), Marsaglia-Tsang for shapes > 1, the shape < 1 gamma,
    shape == 1 (exponential) - with the cached Gaussian of the polar method
    set and unset on entry, and the stream position checked afterwards."""
    out = {}
    rng = np.random.RandomState(5)
    M = 257
    # the shapes a single cell gives (p + x, q + 1 - x; NaN -> mixture)
    x = (rng.random_sample(M) < 0.3).astype(float)
    x[rng.random_sample(M) < 0.2] = np.nan
    shapes = {
        'cell_q': (np.nan_to_num(.25 + x, nan=.25),
            np.nan_to_num(.25 + (1 - x), nan=.25)),
        'cell_u': (np.nan_to_num(1 + x, nan=1.), np.nan_to_num(1 + (1 - x),
            nan=1.)),
        'cluster': (.25 + rng.randint(0, 900, M), .25 + rng.randint(0, 900, M)),
        'mixed': (np.array([.25, 1.25, 7.25, .25, 1., 1., .999, 1e-3, 400.5,
                1.0000001, .5, 3., 1e5, .25] * 8),
            np.array([.25, .25, 3.25, 9.25, 1., 2., 1., 1e-3, .75,
                .9999999, .5, 1., 2e5, 1e4] * 8)),
    }
    for name, (a, b) in shapes.items():
        out[name + '_a'], out[name + '_b'] = a, b
    for seed in (3, 42):
        for pre_gauss in (0, 1):
            np.random.seed(seed)
            if pre_gauss:       # leaves a cached Gaussian behind
                np.random.normal()
            key = f's{seed}_g{pre_gauss}_'
            for name, (a, b) in shapes.items():
                out[key + name] = np.random.beta(a, b)
            out[key + 'gamma'] = np.array([np.random.gamma(3.5, 0.7),
                np.random.gamma(0.4, 2.0), np.random.gamma(1.0, 1.5)])
            st = np.random.get_state()
            out[key + 'has_gauss'] = np.array([st[3]])
            out[key + 'gauss'] = np.array([st[4]])
            out[key + 'tail'] = np.random.random(3)
    save('rng_beta.npz', **out)


# -------------------------------------------------------------- trajectories
def run_traj(kind, data, steps, seed, pb=(.25, .25), sm_prob=.33, sm_steps=3):
    if kind == 'fixed':
        model = CRP(data, DP_alpha=[-1, -1], param_beta=list(pb),
            FN_error=0.1, FP_error=0.001)
        eup = 0
    else:
        model = CRP_errors_learning(data, DP_alpha=[-1, -1],
            param_beta=list(pb), FP_mean=0.01, FP_sd=0.01, FN_mean=0.2,
            FN_sd=0.1)
        eup = .25
    mcmc = MCMC(model, sm_prob=sm_prob, dpa_prob=.25, error_prob=eup,
        sm_ratios=[.75, .25], sm_steps=sm_steps)
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((steps, int(steps * .33)), seed, 1, 0, '', True)
    res = mcmc.get_results()[0]
    out = {k: np.asarray(res[k]) for k in
        ('ML', 'MAP', 'DP_alpha', 'FN', 'FP', 'assignments')}
    out['chain_seed'] = np.array(mcmc.get_seeds()[0], dtype=np.uint64)
    out['sha'] = np.array(hashlib.sha256(
        np.ascontiguousarray(res['assignments'], dtype=np.int64).tobytes()
    ).hexdigest()[:16])
    out['params_last'] = np.asarray(res['params'][-1])
    return out


def gold_trajectories():
    data = ref_io.load_data(os.path.join(REF, 'example_data', 'data.csv'),
        transpose=True)
    assert data.shape == (100, 100)
    out = {}
    for kind in ('fixed', 'learn'):
        r = run_traj(kind, data, 200, 42)
        for k, v in r.items():
            out[f'ex_{kind}_{k}'] = v
        print(kind, r['ML'][[0, 1, 200]], r['sha'])
    small = synth(3, 60, 40, 3, 0.1)
    out['small_data'] = encode(small)
    for kind in ('fixed', 'learn'):
        r = run_traj(kind, small, 120, 7, sm_prob=.5, sm_steps=2)
        for k, v in r.items():
            out[f'sm_{kind}_{k}'] = v
    # uniform parameter prior (beta_prior_uniform branch)
    r = run_traj('fixed', small, 80, 5, pb=(1, 1))
    for k, v in r.items():
        out[f'uni_fixed_{k}'] = v
    save('trajectories.npz', **out)


if __name__ == '__main__':
    if sys.argv[1:] == ['rng_beta']:    # added in round 3: only this file
        gold_rng_beta()
        sys.exit(0)
    gold_calc_ll()
    gold_state_functions()
    gold_normalisers()
    gold_rng()
    gold_rng_beta()
    gold_trajectories()
    meta = {
        'python': sys.version.split()[0],
        'numpy': np.__version__,
        'scipy': scipy.__version__,
        'bottleneck': bottleneck.__version__,
        'reference': 'cbg-ethz/BnpC @ 2024_08_07 (CLI 0.2.1)',
    }
    with open(os.path.join(HERE, 'versions.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(meta)
