"""TEST INFRASTRUCTURE - CPU oracle for the BnpC Bernoulli-likelihood hot path.

A NumPy restatement of the reference's model classes

    CRP                   /root/reference/libs/CRP.py:17-820
    CRP_errors_learning   /root/reference/libs/CRP_learning_errors.py:17-111

with the same public surface, the same dense N x K x M arithmetic, the same
float32/float64 mixing, the same strictly sequential NaN-skipping sums and the
same consumption ORDER of the global legacy ``np.random`` stream, so that a
chain driven by ``bnpc_amd.mcmc`` reproduces the reference's traces.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product (``bnpc_amd/``) never does.

Parity pin: tests/test_oracle_golden.py checks every function below against
the golden vectors that tests/golden/make_golden.py captured from the
imported, unmodified reference (numpy 1.26.4 / scipy 1.7.1 / bottleneck 1.3.2).
The third-party arithmetic the reference relies on and that is absent from
/root/reference: Bottleneck==1.3.5 ``nansum``/``nanargmax`` (restated in
oracle/seqsum.c and below), scipy==1.10.1 ``truncnorm``/``beta``/``gamma``
distributions and ``gammaln`` (called from the installed SciPy, as the
reference does), NumPy ufuncs and the legacy MT19937 ``np.random`` stream.
"""
import ctypes
import os

import numpy as np
from scipy.special import gamma as _gamma_fn, gammaln
from scipy.stats import beta as _beta_dist, truncnorm
from scipy.stats import gamma as _gamma_dist

# The reference traps log(0) and invalid operations and uses the resulting
# FloatingPointError as control flow (libs/CRP.py:10, 94-98, 110-114, 769-773)
np.seterr(divide='raise', over='ignore', under='ignore', invalid='raise')

EPSILON = np.finfo(np.float64).resolution      # libs/CRP.py:11
TMIN = 1e-5                                    # libs/CRP.py:12
TMAX = 1 - TMIN                                # libs/CRP.py:13
log_EPSILON = np.log(EPSILON)                  # libs/CRP.py:14

# --------------------------------------------------------------------------
# sequential NaN-skipping sums (bottleneck.nansum semantics)
# --------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_so = os.path.join(_HERE, '_build', 'liboracle_seqsum.so')
if os.path.exists(_so):
    try:
        _LIB = ctypes.CDLL(_so)
        _dp = ctypes.POINTER(ctypes.c_double)
        _LIB.bnpc_oracle_nansum.restype = ctypes.c_double
        _LIB.bnpc_oracle_nansum.argtypes = [_dp, ctypes.c_long]
        _LIB.bnpc_oracle_nansum_axis1.restype = None
        _LIB.bnpc_oracle_nansum_axis1.argtypes = [
            _dp, ctypes.c_long, ctypes.c_long, _dp]
        _LIB.bnpc_oracle_nansum_axis0.restype = None
        _LIB.bnpc_oracle_nansum_axis0.argtypes = [
            _dp, ctypes.c_long, ctypes.c_long, _dp]
    except OSError:
        _LIB = None


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _seqsum_np(v, axis=None):
    """NumPy fallback with the same order: cumsum is a sequential scan."""
    v = np.asarray(v, dtype=np.float64)
    clean = np.where(np.isnan(v), 0.0, v)
    if axis is None:
        clean = clean.ravel()
        if clean.size == 0:
            return 0.0
        return float(np.cumsum(clean)[-1])
    if clean.shape[axis] == 0:
        return np.zeros(np.delete(clean.shape, axis).astype(int))
    return np.take(np.cumsum(clean, axis=axis), -1, axis=axis)


def seqsum(v, axis=None):
    """bottleneck.nansum: sum in index order, NaN skipped, float64.

    Integer input (the reference also feeds it counts) is summed exactly.
    """
    v = np.asarray(v)
    if v.dtype.kind in 'iub':
        return v.sum(axis=axis)
    if _LIB is None or v.ndim > 2 or (axis is not None and v.ndim != 2):
        if v.ndim == 1 and axis == 0:
            axis = None
        return _seqsum_np(v, axis)
    v = np.ascontiguousarray(v, dtype=np.float64)
    if axis is None:
        return float(_LIB.bnpc_oracle_nansum(_ptr(v), v.size))
    r, c = v.shape
    if axis in (1, -1):
        out = np.empty(r)
        _LIB.bnpc_oracle_nansum_axis1(_ptr(v), r, c, _ptr(out))
    else:
        out = np.empty(c)
        _LIB.bnpc_oracle_nansum_axis0(_ptr(v), r, c, _ptr(out))
    return out


def first_nanargmax(v):
    """bottleneck.nanargmax on a 1-D vector: index of the FIRST maximum."""
    return int(np.nanargmax(v))


# --------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------
class CRP:
    """Dirichlet-process mixture of Bernoulli profiles, fixed error rates.

    data: (cells x mutations) float64 with values 0 | 1 | NaN.
    Mirrors /root/reference/libs/CRP.py:17-820.
    """

    def __init__(self, data, DP_alpha=-1, param_beta=[1, 1],
                FN_error=EPSILON, FP_error=EPSILON):
        self.data = data
        self.cells_total, self.muts_total = data.shape

        self.p, self.q = param_beta
        self.param_prior = _beta_dist(self.p, self.q)
        self.beta_prior_uniform = bool(self.p == self.q == 1)

        # expected emission weights of an unseen cluster, CRP.py:42-44
        m0 = self.beta_fct(self.p, self.q + 1)
        m1 = self.beta_fct(self.p + 1, self.q)
        self._beta_mix_const = np.array([m0, m1]) / (m0 + m1)

        self.FP = FP_error
        self.FN = FN_error

        try:
            neg = DP_alpha[0] < 0 or DP_alpha[1] < 0
        except TypeError:
            neg = True
        if neg:
            self.DP_a_gamma = (np.sqrt(self.cells_total), 1)
        else:
            self.DP_a_gamma = DP_alpha
        self.DP_a_prior = _gamma_dist(*self.DP_a_gamma)
        self.DP_a = self.DP_a_prior.mean()

        self.CRP_prior = None
        self.assignment = None
        self.parameters = None
        self.cells_per_cluster = None

        self.param_proposal_sd = np.array([0.1, 0.25, 0.5])

    def __str__(self):
        return ('\nDPMM with:\n'
            f'\t{self.cells_total} cells\n\t{self.muts_total} mutations\n'
            f'\tFixed FN rate: {self.FN}\n\tFixed FP rate: {self.FP}\n'
            '\n\tPriors:\n'
            f'\tParams.:\tBeta({self.p},{self.q})\n'
            f'\tCRP a_0:\tGamma({self.DP_a_gamma[0]:.1f},'
            f'{self.DP_a_gamma[1]})\n')

    # ------------------------------------------------------------ statics
    @staticmethod
    def beta_fct(p, q):
        return _gamma_fn(p) * _gamma_fn(q) / _gamma_fn(p + q)

    @staticmethod
    def log_CRP_prior(n_i, n, a, dtype=np.float64):
        return np.log(n_i, dtype=dtype) - np.log(n - 1 + a, dtype=dtype)

    @staticmethod
    def _normalize_log_probs(probs):
        """log-weights -> probabilities, floor 1e-15.  CRP.py:88-100"""
        top = first_nanargmax(probs)
        others = np.arange(probs.size) != top
        shifted = probs[others] - probs[top]
        try:
            tail = np.exp(shifted)
        except FloatingPointError:
            tail = np.exp(np.clip(shifted, log_EPSILON, 0))
        lnorm = probs - probs[top] - np.log1p(seqsum(tail))
        return np.exp(np.clip(lnorm, log_EPSILON, 0))

    @staticmethod
    def _normalize_log(probs):
        """log-weights -> normalised LOG-probabilities.  CRP.py:103-116"""
        top = first_nanargmax(probs)
        others = np.arange(probs.size) != top
        try:
            res = probs - probs[top] \
                - np.log1p(seqsum(np.exp(probs[others] - probs[top])))
        except FloatingPointError:
            if probs[0] > probs[1]:
                return np.array([0, log_EPSILON])
            return np.array([log_EPSILON, 0])
        return res

    # --------------------------------------------------------------- init
    def init(self, mode='random', assign=False):
        """CRP.py:119-152"""
        N = self.cells_total
        if assign:
            labels = np.array(assign)
        elif mode == 'separate':
            labels = np.arange(N, dtype=int)
        elif mode == 'together':
            labels = np.zeros(N, dtype=int)
        elif mode == 'random':
            labels = np.random.randint(0, high=N, size=N)
        else:
            raise TypeError(f'Unsupported Initialization: {mode}')

        # relabel to 0..K-1 in order of the sorted original labels
        _, inv, counts = np.unique(labels, return_inverse=True,
            return_counts=True)
        self.assignment = inv.astype(labels.dtype if labels.dtype.kind == 'i'
            else int).reshape(-1)
        self.cells_per_cluster = {i: c for i, c in enumerate(counts)}
        self.parameters = self._init_cl_params('assign' if assign else mode)
        self.init_DP_prior()

    def _beta_from_counts(self, ones, zeros):
        return np.random.beta(self.p + ones, self.q + zeros)

    def _init_cl_params(self, mode='random', fkt=1):
        """CRP.py:155-180"""
        params = np.zeros(self.data.shape)
        if mode == 'separate':
            params = np.random.beta(
                np.nan_to_num(self.p + self.data * fkt,
                    nan=self._beta_mix_const[0]),
                np.nan_to_num(self.q + (1 - self.data) * fkt,
                    nan=self._beta_mix_const[1]))
        elif mode == 'together':
            params[0] = self._beta_from_counts(
                seqsum(self.data * fkt, axis=0),
                seqsum((1 - self.data) * fkt, axis=0))
        elif mode == 'assign':
            for cl in self.cells_per_cluster:
                sub = self.data[np.where(self.assignment == cl)]
                params[cl] = self._beta_from_counts(
                    seqsum(sub * fkt, axis=0), seqsum((1 - sub) * fkt, axis=0))
        elif mode == 'random':
            k = np.unique(self.assignment)
            params[k] = np.random.uniform(size=(k.size, self.muts_total))
        return np.clip(params, TMIN, TMAX).astype(np.float32)

    def _init_cl_params_new(self, i, fkt=1):
        """Beta draw from the column counts of cells i.  CRP.py:183-188"""
        sub = self.data[i]
        params = self._beta_from_counts(
            seqsum(sub * fkt, axis=0), seqsum((1 - sub) * fkt, axis=0))
        return np.clip(params, TMIN, TMAX).astype(np.float32)

    def init_DP_prior(self):
        """CRP.py:191-194: index = cluster size, [-1] = new cluster."""
        sizes = np.append(np.arange(1, self.cells_total + 1), self.DP_a)
        self.CRP_prior = np.append(
            0, self.log_CRP_prior(sizes, self.cells_total, self.DP_a))

    # ------------------------------------------------------ likelihood core
    def _Bernoulli_FN(self, x):
        """P(x | genotype 1) = (1-FN)^x FN^(1-x).  CRP.py:207-208"""
        return (1 - self.FN) ** x * self.FN ** (1 - x)

    def _Bernoulli_FP(self, x):
        """P(x | genotype 0) = (1-FP)^(1-x) FP^x.  CRP.py:211-212"""
        return (1 - self.FP) ** (1 - x) * self.FP ** x

    def _calc_ll(self, x, theta, flat=False):
        """CRP.py:197-204.  (1 - theta) stays in theta's dtype."""
        mixed = theta * self._Bernoulli_FN(x) \
            + (1 - theta) * self._Bernoulli_FP(x)
        ll = np.log(mixed)
        if flat:
            return seqsum(ll)
        return seqsum(ll, axis=1)

    def get_lpost_single(self, cell_id, cl_ids):
        """CRP.py:223-227"""
        ll = self._calc_ll(self.data[[cell_id]], self.parameters[cl_ids])
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        return ll + self.CRP_prior[sizes]

    def get_lpost_single_new_cluster(self):
        """CRP.py:230-234"""
        wt = self._beta_mix_const[0] * self._Bernoulli_FP(self.data)
        mut = self._beta_mix_const[1] * self._Bernoulli_FN(self.data)
        return seqsum(np.log(mut + wt), axis=1) + self.CRP_prior[-1]

    def get_ll_full(self):
        """CRP.py:237-238"""
        return self._calc_ll(self.data, self.parameters[self.assignment], True)

    def get_lprior_full(self):
        """CRP.py:241-251"""
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        lprior = self.DP_a_prior.logpdf(self.DP_a) \
            + seqsum(self.CRP_prior[sizes])
        if not self.beta_prior_uniform:
            ids = np.fromiter(self.cells_per_cluster.keys(), dtype=int)
            lprior += seqsum(self.param_prior.logpdf(self.parameters[ids]))
        return lprior

    # ------------------------------------------------------------- Gibbs
    def update_assignments_Gibbs(self):
        """One sequential sweep over all cells.  CRP.py:254-288"""
        post_new_all = self.get_lpost_single_new_cluster()
        for cell in np.random.permutation(self.cells_total):
            old = self.assignment[cell]
            if self.cells_per_cluster[old] == 1:
                del self.cells_per_cluster[old]
            else:
                self.cells_per_cluster[old] -= 1

            ids = np.fromiter(self.cells_per_cluster.keys(), dtype=int)
            post = np.append(self.get_lpost_single(cell, ids),
                post_new_all[cell])
            probs = self._normalize_log_probs(post)
            pick = np.random.choice(np.append(ids, -1), p=probs)
            if pick == -1:
                pick = self.init_new_cluster(cell)
            self.assignment[cell] = pick
            if pick in self.cells_per_cluster:
                self.cells_per_cluster[pick] += 1
            else:
                self.cells_per_cluster[pick] = 1

    def init_new_cluster(self, cell_id):
        """CRP.py:291-294"""
        cl = self.get_empty_cluster()
        self.parameters[cl] = self._init_cl_params_new([cell_id])
        return cl

    def get_empty_cluster(self):
        """Lowest unused cluster id.  CRP.py:297-299"""
        i = 0
        while i in self.cells_per_cluster:
            i += 1
        return i

    # ------------------------------------------------ cluster parameters MH
    def update_parameters(self, step_no=None):
        """CRP.py:302-311"""
        declined = np.zeros(len(self.cells_per_cluster), dtype=int)
        for n, cl in enumerate(self.cells_per_cluster):
            members = np.argwhere(self.assignment == cl).flatten()
            self.parameters[cl], _, declined[n] = self.MH_cluster_params(
                self.parameters[cl], members)
        return declined.sum(), (self.muts_total - declined).sum()

    def MH_cluster_params(self, old_params, cells, trans_prob=False):
        """Per-mutation Metropolis-Hastings update.  CRP.py:314-344"""
        M = self.muts_total
        std = np.random.choice(self.param_proposal_sd, size=M)
        a = (TMIN - old_params) / std
        b = (TMAX - old_params) / std
        new_params = truncnorm.rvs(
            a, b, loc=old_params, scale=std, size=M).astype(np.float32)

        A = self._get_log_A(new_params, old_params, cells, a, b, std,
            trans_prob)
        u = np.log(np.random.random(M))
        decline = u >= A
        new_params[decline] = old_params[decline]

        if trans_prob:
            A[decline] = np.log(-1 * np.expm1(A[decline]))
            return new_params, seqsum(A), decline.sum()
        return new_params, np.nan, decline.sum()

    def _get_log_A(self, new_params, old_params, cells, a, b, std, clip=False):
        """log MH ratio per mutation.  CRP.py:347-383"""
        fwd = truncnorm.logpdf(new_params, a, b, loc=old_params, scale=std)
        a_rev = (TMIN - new_params) / std
        b_rev = (TMAX - new_params) / std
        rev = truncnorm.logpdf(old_params, a_rev, b_rev, loc=new_params,
            scale=std)

        x = self.data[cells]
        mut = self._Bernoulli_FN(x)
        wt = self._Bernoulli_FP(x)
        new_ll = seqsum(
            np.log(new_params * mut + (1 - new_params) * wt), axis=0)
        old_ll = seqsum(
            np.log(old_params * mut + (1 - old_params) * wt), axis=0)

        if self.beta_prior_uniform:
            new_prior = old_prior = 0
        else:
            new_prior = self.param_prior.logpdf(new_params)
            old_prior = self.param_prior.logpdf(old_params)

        A = new_ll + new_prior - old_ll - old_prior + rev - fwd
        if clip:
            return np.clip(A, a_min=None, a_max=0)
        return A

    # ----------------------------------------------------------- DP alpha
    def update_DP_alpha(self):
        """Escobar & West (1995) auxiliary-variable update.  CRP.py:386-410"""
        k = len(self.cells_per_cluster)
        shape0, rate0 = self.DP_a_gamma
        eta = np.random.beta(self.DP_a + 1, self.cells_total)
        w = (shape0 + k - 1) / (self.cells_total * (rate0 - np.log(eta)))
        pi_eta = w / (1 + w)
        if np.random.random() < pi_eta:
            new_alpha = np.random.gamma(shape0 + k, rate0 - np.log(eta))
        else:
            new_alpha = np.random.gamma(shape0 + k - 1, rate0 - np.log(eta))
        self.DP_a = max(1 + EPSILON, new_alpha)
        self.init_DP_prior()

    # ------------------------------------------------------- split / merge
    def update_assignments_split_merge(self, ratios=[.75, .25], step_no=5):
        """CRP.py:417-431"""
        K = len(self.cells_per_cluster)
        if K == 1:
            return (self.do_split_move(step_no), 0)
        if K == self.cells_total:
            return (self.do_merge_move(step_no), 1)
        move = np.random.choice([0, 1], p=ratios)
        if move == 0:
            return (self.do_split_move(step_no), move)
        return (self.do_merge_move(step_no), move)

    def do_split_move(self, step_no=5):
        """CRP.py:434-481"""
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=int)
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        probs = sizes / sizes.sum()
        while True:
            cl = np.random.choice(ids, p=probs)
            cells = np.argwhere(self.assignment == cl).flatten()
            if cells.size != 1:
                break
        ai, aj = np.random.choice(cells.size, size=2, replace=False)
        cells[0], cells[ai] = cells[ai], cells[0]
        cells[-1], cells[aj] = cells[aj], cells[-1]

        pos = np.argwhere(ids == cl).flatten()
        ltrans = np.log(probs[pos]) \
            - np.log(self.cells_per_cluster[cl]) \
            - np.log(self.cells_per_cluster[cl] - 1)
        size_data = (ltrans, np.delete(sizes, pos))

        accept, new_assign, new_params = self.run_rg_nc(
            'split', cells, size_data, step_no)
        if not accept:
            return [0, 1]
        new_cl = self.get_empty_cluster()
        self.parameters[cl] = new_params[0]
        self.parameters[new_cl] = new_params[1]
        moved = np.append(cells[1:-1][np.where(new_assign == 1)], cells[-1])
        self.assignment[moved] = new_cl
        self.cells_per_cluster[cl] -= moved.size
        self.cells_per_cluster[new_cl] = moved.size
        return [1, 0]

    def do_merge_move(self, step_no=5):
        """CRP.py:484-524"""
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=int)
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        inv = 1 / sizes
        probs = inv / inv.sum()
        cl_i, cl_j = np.random.choice(ids, p=probs, size=2, replace=False)

        cells_i = np.argwhere(self.assignment == cl_i).flatten()
        ai = np.random.choice(cells_i.size)
        cells_i[0], cells_i[ai] = cells_i[ai], cells_i[0]
        cells_j = np.argwhere(self.assignment == cl_j).flatten()
        aj = np.random.choice(cells_j.size)
        cells_j[-1], cells_j[aj] = cells_j[aj], cells_j[-1]
        cells = np.concatenate((cells_i, cells_j)).flatten()

        pos = np.argwhere((ids == cl_j) | (ids == cl_i)).flatten()
        size_data = seqsum(np.log(probs[pos])) - seqsum(np.log(sizes[pos]))

        accept, new_params = self.run_rg_nc('merge', cells, size_data, step_no)
        if not accept:
            return [0, 1]
        self.parameters[cl_i] = new_params
        self.assignment[cells_j] = cl_i
        self.cells_per_cluster[cl_i] += cells_j.size
        del self.cells_per_cluster[cl_j]
        return [1, 0]

    def run_rg_nc(self, move, cells, size_data, scan_no):
        """Jain & Neal (2007) restricted Gibbs, non-conjugate.  CRP.py:527-544"""
        self._rg_init_split(cells)
        self.rg_params_merge = self._init_cl_params_new(cells)
        for _ in range(scan_no):
            self._rg_scan_split(cells)
            self._rg_scan_merge(cells)
        if move == 'split':
            return self._do_rg_split_MH(cells, size_data)
        return self._do_rg_merge_MH(cells, size_data)

    def _rg_members(self, cells, which):
        """Non-anchor cells currently in launch cluster `which` + its anchor."""
        S = cells[1:-1]
        anchor = cells[0] if which == 0 else cells[-1]
        return np.append(S[np.argwhere(self.rg_assignment == which)], anchor)

    def _rg_init_split(self, cells, random=False):
        """CRP.py:547-567"""
        i, j, S = cells[0], cells[-1], cells[1:-1]
        if S.size == 0:
            self.rg_assignment = np.array([])
        elif random:
            self.rg_assignment = np.random.choice([0, 1], size=(S.size))
        else:
            fill = self._beta_mix_const[0]
            ll_i = self._calc_ll(self.data[S],
                np.nan_to_num(self.data[i], nan=fill))
            ll_j = self._calc_ll(self.data[S],
                np.nan_to_num(self.data[j], nan=fill))
            self.rg_assignment = np.where(ll_j > ll_i, 1, 0)
        par_i = self._init_cl_params_new(self._rg_members(cells, 0))
        par_j = self._init_cl_params_new(self._rg_members(cells, 1))
        self.rg_params_split = np.stack([par_i, par_j])

    def _rg_scan_split(self, cells, trans_prob=False):
        """CRP.py:570-578"""
        if cells.size == 2:
            prob_cl = 0
        else:
            prob_cl = self._rg_scan_assign(cells, trans_prob)
        prob_par = self._rg_scan_params(cells, trans_prob)
        if trans_prob:
            return prob_cl + prob_par

    def _rg_scan_merge(self, cells, trans_prob=False):
        """CRP.py:581-587"""
        self.rg_params_merge, prob, _ = self.MH_cluster_params(
            self.rg_params_merge, cells, trans_prob)
        if trans_prob:
            return prob

    def _rg_scan_params(self, cells, trans_prob=False):
        """CRP.py:590-606"""
        prob = np.zeros(2)
        for cl in range(2):
            self.rg_params_split[cl], prob[cl], _ = self.MH_cluster_params(
                self.rg_params_split[cl], self._rg_members(cells, cl),
                trans_prob)
        if trans_prob:
            return prob.sum()

    def _rg_scan_assign(self, cells, trans_prob=False):
        """Sequential 2-way restricted Gibbs scan.  CRP.py:609-632"""
        ll = self._rg_get_ll(cells[1:-1], self.rg_params_split)
        n = cells.size
        if trans_prob:
            prob = np.zeros(n - 2)
        for cell in np.random.permutation(n - 2):
            self.rg_assignment[cell] = -1
            n_j = seqsum(self.rg_assignment) + 2
            n_i = n - n_j - 1
            lpost = ll[cell] + self.log_CRP_prior([n_i, n_j], n, self.DP_a)
            lprobs = self._normalize_log(lpost)
            pick = np.random.choice([0, 1], p=np.exp(lprobs))
            self.rg_assignment[cell] = pick
            if trans_prob:
                prob[cell] = lprobs[pick]
        if trans_prob:
            return seqsum(prob)

    def _rg_get_ll(self, cells, params):
        """CRP.py:635-638"""
        x = self.data[cells]
        return np.stack([self._calc_ll(x, params[0]),
            self._calc_ll(x, params[1])], axis=1)

    def _do_rg_split_MH(self, cells, size_data):
        """CRP.py:641-653"""
        A = self._get_trans_prob_ratio_split(cells) \
            + self._get_lprior_ratio_split(cells) \
            + self._get_ll_ratio(cells, 'split') \
            + self._get_ltrans_prob_size_ratio_split(*size_data)
        if np.unique(self.rg_assignment).size == 1:
            return (False, [], [])
        if np.log(np.random.random()) < A:
            return (True, self.rg_assignment, self.rg_params_split)
        return (False, [], [])

    def _do_rg_merge_MH(self, cells, size_data):
        """CRP.py:656-665"""
        A = self._get_trans_prob_ratio_merge(cells) \
            + self._get_lprior_ratio_merge(cells) \
            + self._get_ll_ratio(cells, 'merge') \
            + self._get_ltrans_prob_size_ratio_merge(size_data)
        if np.log(np.random.random()) < A:
            return (True, self.rg_params_merge)
        return (False, [])

    def _get_trans_prob_ratio_split(self, cells):
        """Jain & Neal eq. 15.  CRP.py:668-682"""
        gs_split = self._rg_scan_split(cells, trans_prob=True)
        std = np.random.choice(self.param_proposal_sd, size=self.muts_total)
        a = (TMIN - self.rg_params_merge) / std
        b = (TMAX - self.rg_params_merge) / std
        gs_merge = seqsum(self._get_log_A(
            self.parameters[self.assignment[cells[0]]], self.rg_params_merge,
            cells, a, b, std, True))
        return gs_merge - gs_split

    def _get_trans_prob_ratio_merge(self, cells):
        """Jain & Neal eq. 16.  CRP.py:685-692"""
        gs_merge = self._rg_scan_merge(cells, trans_prob=True)
        gs_split = self._rg_get_split_prob(cells)
        return gs_split - gs_merge

    def _get_lprior_ratio_split(self, cells):
        """Jain & Neal eq. 7.  CRP.py:695-713"""
        n = self.rg_assignment.size + 2
        n_j = seqsum(self.rg_assignment) + 1
        n_i = n - n_j
        ratio = np.log(self.DP_a) - gammaln(n)
        if n_i > 0:
            ratio += gammaln(n_j)
        if n_j > 0:
            ratio += gammaln(n_i)
        if not self.beta_prior_uniform:
            cl = self.assignment[cells[0]]
            ratio += seqsum(self.param_prior.logpdf(self.rg_params_split)) \
                - seqsum(self.param_prior.logpdf(self.parameters[cl]))
        return ratio

    def _get_ll_ratio(self, cells, move):
        """Jain & Neal eq. 11/12.  CRP.py:716-733"""
        S = cells[1:-1]
        i_ids = np.append(S[np.argwhere(self.rg_assignment == 0)], cells[0])
        j_ids = np.append(S[np.nonzero(self.rg_assignment)], cells[-1])
        ll_i = self._calc_ll(self.data[i_ids], self.rg_params_split[0], True)
        ll_j = self._calc_ll(self.data[j_ids], self.rg_params_split[1], True)
        ll_all = self._calc_ll(self.data[cells], self.rg_params_merge, True)
        if move == 'split':
            return ll_i + ll_j - ll_all
        return ll_all - ll_i - ll_j

    def _get_lprior_ratio_merge(self, cells):
        """Jain & Neal eq. 8.  CRP.py:736-754"""
        n = cells.size
        n_j = seqsum(self.rg_assignment) + 1
        n_i = n - n_j
        ratio = gammaln(n) - np.log(self.DP_a)
        if n_i > 0:
            ratio -= gammaln(n_i)
        if n_j > 0:
            ratio -= gammaln(n_j)
        if not self.beta_prior_uniform:
            cls = self.assignment[[cells[0], cells[-1]]]
            ratio += seqsum(self.param_prior.logpdf(self.rg_params_merge)) \
                - seqsum(self.param_prior.logpdf(self.parameters[cls]))
        return ratio

    def _get_ltrans_prob_size_ratio_split(self, ltrans_prob_size, cluster_size):
        """CRP.py:757-764"""
        n_j = seqsum(self.rg_assignment) + 1
        n_i = self.rg_assignment.size + 2 - n_j
        norm = seqsum(1 / np.append(cluster_size, [n_i, n_j]))
        rev = np.log(1 / n_i / norm) + np.log(1 / n_j / norm)
        return rev - ltrans_prob_size[0]

    def _get_ltrans_prob_size_ratio_merge(self, trans_prob_size):
        """CRP.py:767-774 (log(0) / log(-1) trapped as FloatingPointError)"""
        try:
            rev = -np.log(self.cells_total) \
                - np.log(self.rg_assignment.size - 1)
        except FloatingPointError:
            rev = -np.log(self.cells_total)
        return rev - trans_prob_size

    def _rg_get_split_prob(self, cells):
        """Probability of reaching the ORIGINAL split from the launch state.
        CRP.py:777-820 (bounds (0,1), rg_assignment overwritten in place)."""
        std = np.random.choice(self.param_proposal_sd,
            size=(2, self.muts_total))
        a = (0 - self.rg_params_split) / std
        b = (1 - self.rg_params_split) / std

        i, j, S = cells[0], cells[-1], cells[1:-1]
        cl_i = self.assignment[i]
        cl_j = self.assignment[j]
        prob_i = seqsum(self._get_log_A(
            self.parameters[cl_i], self.rg_params_split[0],
            self._rg_members(cells, 0), a[0], b[0], std[0], True))
        prob_j = seqsum(self._get_log_A(
            self.parameters[cl_j], self.rg_params_split[1],
            self._rg_members(cells, 1), a[1], b[1], std[1], True))

        ll = self._rg_get_ll(S, (self.parameters[cl_i], self.parameters[cl_j]))
        n = cells.size
        prob_assign = np.zeros(S.size)
        target = np.where(self.assignment[S] == cl_i, 0, 1)
        for obs in range(S.size):
            self.rg_assignment[obs] = -1
            n_j = seqsum(self.rg_assignment) + 2
            n_i = n - n_j - 1
            lpost = ll[obs] + self.log_CRP_prior([n_i, n_j], n, self.DP_a)
            lprobs = self._normalize_log(lpost)
            self.rg_assignment[obs] = target[obs]
            prob_assign[obs] = lprobs[target[obs]]
        return prob_i + prob_j + seqsum(prob_assign)


class CRP_errors_learning(CRP):
    """CRP with truncated-normal priors on FP / FN and an MH update of both.
    Mirrors /root/reference/libs/CRP_learning_errors.py:17-111."""

    def __init__(self, data, DP_alpha=1, param_beta=[1, 1],
                FP_mean=0.001, FP_sd=0.0005, FN_mean=0.25, FN_sd=0.05):
        super().__init__(data, DP_alpha, param_beta, FN_mean, FP_mean)
        self.FP_prior = truncnorm(
            (0 - FP_mean) / FP_sd, (1 - FP_mean) / FP_sd, FP_mean, FP_sd)
        self.FP_sd = np.array([FP_sd * 0.5, FP_sd, FP_sd * 1.5])
        self.FN_prior = truncnorm(
            (0 - FN_mean) / FN_sd, (1 - FN_mean) / FN_sd, FN_mean, FN_sd)
        self.FN_sd = np.array([FN_sd * 0.5, FN_sd, FN_sd * 1.5])

    def __str__(self):
        return ('\nDPMM with:\n'
            f'\t{self.cells_total} cells\n\t{self.muts_total} mutations\n'
            '\tlearning errors\n'
            '\n\tPriors:\n'
            f'\tparams.:\tBeta({self.p},{self.q})\n'
            f'\tCRP a_0:\tGamma({self.DP_a_gamma[0]:.2f},'
            f'{self.DP_a_gamma[1]})\n'
            f'\tFP:\t\ttrunc norm({self.FP_prior.args[2]},'
            f'{self.FP_prior.args[3]})\n'
            f'\tFN:\t\ttrunc norm({self.FN_prior.args[2]},'
            f'{self.FN_prior.args[3]})\n')

    def get_lprior_full(self):
        """CRP_learning_errors.py:47-49"""
        return super().get_lprior_full() \
            + self.FP_prior.logpdf(self.FP) + self.FN_prior.logpdf(self.FN)

    def update_error_rates(self):
        """CRP_learning_errors.py:52-55"""
        self.FP, FP_count = self.MH_error_rates('FP')
        self.FN, FN_count = self.MH_error_rates('FN')
        return FP_count, FN_count

    def get_ll_full_error(self, FP, FN):
        """Total log-likelihood under trial error rates.
        CRP_learning_errors.py:58-63"""
        par = self.parameters[self.assignment]
        mut = par * (1 - FN) ** self.data * FN ** (1 - self.data)
        wt = (1 - par) * (1 - FP) ** (1 - self.data) * FP ** self.data
        return seqsum(np.log(mut + wt))

    def MH_error_rates(self, error_type):
        """CRP_learning_errors.py:66-111"""
        if error_type == 'FP':
            old, prior, sds = self.FP, self.FP_prior, self.FP_sd
        else:
            old, prior, sds = self.FN, self.FN_prior, self.FN_sd

        std = np.random.choice(sds)
        a = (0 - old) / std
        b = (1 - old) / std
        try:
            new = truncnorm.rvs(a, b, loc=old, scale=std)
        except FloatingPointError:
            new = truncnorm.rvs(a, np.inf, loc=old, scale=std)

        fwd = truncnorm.logpdf(new, a, b, loc=old, scale=std)
        rev = truncnorm.logpdf(old, (0 - new) / std, (1 - new) / std,
            loc=new, scale=std)

        if error_type == 'FP':
            new_ll = self.get_ll_full_error(new, self.FN)
            old_ll = self.get_ll_full_error(old, self.FN)
        else:
            new_ll = self.get_ll_full_error(self.FP, new)
            old_ll = self.get_ll_full_error(self.FP, old)

        A = new_ll + prior.logpdf(new) - old_ll - prior.logpdf(old) \
            + rev - fwd
        if np.log(np.random.random()) < A:
            return new, [1, 0]
        return old, [0, 1]
