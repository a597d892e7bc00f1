"""TEST INFRASTRUCTURE - CPU oracle for the BnpC Bernoulli-likelihood hot path.

A NumPy restatement of the reference's model classes

    CRP                   /root/reference/libs/CRP.py:17-820
    CRP_errors_learning   /root/reference/libs/CRP_learning_errors.py:17-111

with the same public surface, the same dense N x K x M arithmetic, the same
float32/float64 mixing, the same strictly sequential NaN-skipping sums and the
same consumption ORDER of the global legacy ``np.random`` stream, so that a
chain driven by ``bnpc_amd.mcmc`` reproduces the reference's traces.  The
oracle is organised by concern, one module each:

    constants.py     EPSILON / TMIN / TMAX and the floating-point traps
    seqsum.py        bottleneck-style sequential sums (+ oracle/seqsum.c)
    likelihood.py    emissions, `_calc_ll`, posteriors of one cell, normalisers
    gibbs.py         the assignment sweep and the concentration update
    param_moves.py   per-mutation MH of the cluster profiles
    split_merge.py   restricted-Gibbs split-merge moves
    crp_numpy.py     (this file) construction, initialisation, error learning

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``bnpc_amd/``) never does.

Parity pin: tests/test_oracle_golden.py checks the functions of this package
against the golden vectors that tests/golden/make_golden.py captured from the
imported, unmodified reference (numpy 1.26.4 / scipy 1.7.1 / bottleneck 1.3.2);
oracle/check_same_stack.py shows bit identity on that same stack.  The
third-party arithmetic the reference relies on and that is absent from
/root/reference: Bottleneck==1.3.5 ``nansum``/``nanargmax`` (restated in
seqsum.py / seqsum.c), scipy==1.10.1 ``truncnorm``/``beta``/``gamma``
distributions and ``gammaln`` (called from the installed SciPy, as the
reference does), NumPy ufuncs and the legacy MT19937 ``np.random`` stream.
"""
import numpy as np
from scipy.special import gamma as _gamma_fn
from scipy.stats import beta as _beta_dist, truncnorm
from scipy.stats import gamma as _gamma_dist

from .constants import EPSILON, TMAX, TMIN, log_EPSILON  # noqa: F401
from .gibbs import GibbsMoves
from .likelihood import Likelihood
from .param_moves import ParameterMoves
from .seqsum import _seqsum_np, first_nanargmax, seqsum  # noqa: F401
from .split_merge import SplitMergeMoves


def _euler_beta(p, q):
    return _gamma_fn(p) * _gamma_fn(q) / _gamma_fn(p + q)


class CRP(Likelihood, GibbsMoves, ParameterMoves, SplitMergeMoves):
    """Dirichlet-process mixture of Bernoulli profiles with fixed error
    rates; data is (cells x mutations) float64 holding 0 | 1 | NaN."""

    beta_fct = staticmethod(_euler_beta)

    def __init__(self, data, DP_alpha=-1, param_beta=[1, 1],
                FN_error=EPSILON, FP_error=EPSILON):
        """libs/CRP.py:27-80"""
        self.data = data
        self.cells_total, self.muts_total = data.shape
        self.FP, self.FN = FP_error, FN_error

        # Beta(p, q) prior of every profile entry; its mean gives the
        # emission weights of a cluster nobody sits in yet (CRP.py:42-44)
        self.p, self.q = param_beta
        self.param_prior = _beta_dist(self.p, self.q)
        self.beta_prior_uniform = bool(self.p == self.q == 1)
        w0 = _euler_beta(self.p, self.q + 1)
        w1 = _euler_beta(self.p + 1, self.q)
        self._beta_mix_const = np.array([w0, w1]) / (w0 + w1)

        # Gamma prior of the concentration; default (sqrt(N), 1)
        try:
            given = not (DP_alpha[0] < 0 or DP_alpha[1] < 0)
        except TypeError:
            given = False
        self.DP_a_gamma = DP_alpha if given \
            else (np.sqrt(self.cells_total), 1)
        self.DP_a_prior = _gamma_dist(*self.DP_a_gamma)
        self.DP_a = self.DP_a_prior.mean()

        self.param_proposal_sd = np.array([0.1, 0.25, 0.5])
        self.CRP_prior = self.assignment = self.parameters = None
        self.cells_per_cluster = None

    def _describe(self, error_lines, tail_lines=(), label='Params.',
                digits=1):
        shape, rate = self.DP_a_gamma
        head = ['', 'DPMM with:', f'\t{self.cells_total} cells',
            f'\t{self.muts_total} mutations', *error_lines, '', '\tPriors:',
            f'\t{label}:\tBeta({self.p},{self.q})',
            f'\tCRP a_0:\tGamma({shape:.{digits}f},{rate})']
        return '\n'.join([*head, *tail_lines]) + '\n'

    def __str__(self):
        return self._describe([f'\tFixed FN rate: {self.FN}',
            f'\tFixed FP rate: {self.FP}'])

    # ------------------------------------------------------- initial state
    def init(self, mode='random', assign=False):
        """libs/CRP.py:119-152.  Labels are compacted to 0..K-1 in sorted
        order; the dict is filled in that order."""
        n = self.cells_total
        if assign:
            labels, mode = np.array(assign), 'assign'
        elif mode == 'random':
            labels = np.random.randint(0, high=n, size=n)
        elif mode == 'separate':
            labels = np.arange(n, dtype=int)
        elif mode == 'together':
            labels = np.zeros(n, dtype=int)
        else:
            raise TypeError(f'Unsupported Initialization: {mode}')

        _, compact, sizes = np.unique(labels, return_inverse=True,
            return_counts=True)
        kind = labels.dtype if labels.dtype.kind == 'i' else int
        self.assignment = compact.astype(kind).reshape(-1)
        self.cells_per_cluster = dict(enumerate(sizes))
        self.parameters = self._init_cl_params(mode)
        self.init_DP_prior()

    def init_DP_prior(self):
        """Table of log CRP weights indexed by cluster size; the last entry
        is the weight of a new cluster (libs/CRP.py:191-194)."""
        counts = np.append(np.arange(1, self.cells_total + 1), self.DP_a)
        self.CRP_prior = np.append(0,
            self.log_CRP_prior(counts, self.cells_total, self.DP_a))

    def _beta_given(self, rows, fkt):
        """Beta(p + #ones, q + #zeros) per mutation for a block of cells."""
        ones = seqsum(rows * fkt, axis=0)
        zeros = seqsum((1 - rows) * fkt, axis=0)
        return np.random.beta(self.p + ones, self.q + zeros)

    def _init_cl_params(self, mode='random', fkt=1):
        """libs/CRP.py:155-180 -> (N, M) float32, row = cluster id."""
        theta = np.zeros(self.data.shape)
        if mode == 'random':
            used = np.unique(self.assignment)
            theta[used] = np.random.uniform(size=(used.size, self.muts_total))
        elif mode == 'separate':
            fill0, fill1 = self._beta_mix_const
            theta = np.random.beta(
                np.nan_to_num(self.p + self.data * fkt, nan=fill0),
                np.nan_to_num(self.q + (1 - self.data) * fkt, nan=fill1))
        elif mode == 'together':
            theta[0] = self._beta_given(self.data, fkt)
        elif mode == 'assign':
            for cl in self.cells_per_cluster:
                members = np.where(self.assignment == cl)
                theta[cl] = self._beta_given(self.data[members], fkt)
        return np.clip(theta, TMIN, TMAX).astype(np.float32)

    def _init_cl_params_new(self, i, fkt=1):
        """A fresh profile for the cells `i` (libs/CRP.py:183-188)."""
        return np.clip(self._beta_given(self.data[i], fkt), TMIN, TMAX) \
            .astype(np.float32)


class CRP_errors_learning(CRP):
    """The same mixture with truncated-normal priors on FP / FN and an MH
    update of both (libs/CRP_learning_errors.py:17-111)."""

    def __init__(self, data, DP_alpha=1, param_beta=[1, 1],
                FP_mean=0.001, FP_sd=0.0005, FN_mean=0.25, FN_sd=0.05):
        super().__init__(data, DP_alpha, param_beta, FN_mean, FP_mean)
        self.FP_prior, self.FP_sd = self._rate_prior(FP_mean, FP_sd)
        self.FN_prior, self.FN_sd = self._rate_prior(FN_mean, FN_sd)

    @staticmethod
    def _rate_prior(mean, sd):
        """N(mean, sd) truncated to (0, 1) + the three proposal widths."""
        prior = truncnorm((0 - mean) / sd, (1 - mean) / sd, mean, sd)
        return prior, np.array([sd * 0.5, sd, sd * 1.5])

    def __str__(self):
        fp, fn = self.FP_prior.args, self.FN_prior.args
        return self._describe(['\tlearning errors'],
            [f'\tFP:\t\ttrunc norm({fp[2]},{fp[3]})',
             f'\tFN:\t\ttrunc norm({fn[2]},{fn[3]})'],
            label='params.', digits=2)

    def get_lprior_full(self):
        """libs/CRP_learning_errors.py:47-49"""
        return super().get_lprior_full() \
            + self.FP_prior.logpdf(self.FP) + self.FN_prior.logpdf(self.FN)

    def get_ll_full_error(self, FP, FN):
        """Total log-likelihood under trial rates
        (libs/CRP_learning_errors.py:58-63; note the product order
        `theta * (1-FN)**x * FN**(1-x)`, which differs from `_calc_ll`)."""
        theta = self.parameters[self.assignment]
        x = self.data
        mut = theta * (1 - FN) ** x * FN ** (1 - x)
        wt = (1 - theta) * (1 - FP) ** (1 - x) * FP ** x
        return seqsum(np.log(mut + wt))

    def update_error_rates(self):
        """FP first, then FN with the new FP
        (libs/CRP_learning_errors.py:52-55)."""
        self.FP, FP_outcome = self.MH_error_rates('FP')
        self.FN, FN_outcome = self.MH_error_rates('FN')
        return FP_outcome, FN_outcome

    def MH_error_rates(self, error_type):
        """One MH step of a scalar rate (libs/CRP_learning_errors.py:66-111).
        Stream: choice(3 widths) -> truncnorm.rvs -> random()."""
        is_fp = error_type == 'FP'
        now = self.FP if is_fp else self.FN
        prior = self.FP_prior if is_fp else self.FN_prior
        std = np.random.choice(self.FP_sd if is_fp else self.FN_sd)

        a, b = (0 - now) / std, (1 - now) / std
        try:
            trial = truncnorm.rvs(a, b, loc=now, scale=std)
        except FloatingPointError:
            trial = truncnorm.rvs(a, np.inf, loc=now, scale=std)
        forward = truncnorm.logpdf(trial, a, b, loc=now, scale=std)
        backward = truncnorm.logpdf(now, (0 - trial) / std, (1 - trial) / std,
            loc=trial, scale=std)

        def total_ll(rate):
            if is_fp:
                return self.get_ll_full_error(rate, self.FN)
            return self.get_ll_full_error(self.FP, rate)

        ll_trial = total_ll(trial)
        ll_now = total_ll(now)
        A = ll_trial + prior.logpdf(trial) - ll_now - prior.logpdf(now) \
            + backward - forward
        if np.log(np.random.random()) < A:
            return trial, [1, 0]
        return now, [0, 1]
