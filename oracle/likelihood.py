"""TEST INFRASTRUCTURE - the dense likelihood arithmetic of the CPU oracle.

This is the array op the HIP kernels replace (SURVEY.md section 8(a) rows
a1-a4, a8): cells x clusters x mutations float64 temporaries, `(1 - theta)`
in theta's own dtype, a log per element and a strictly sequential
NaN-skipping sum over mutations.  Nothing here is optimised - the point of
the oracle is to do what the reference does, the way it does it.
"""
import numpy as np

from .constants import log_EPSILON
from .seqsum import first_nanargmax, seqsum


def crp_log_weight(n_i, n, alpha, dtype=np.float64):
    """log(n_i / (n - 1 + alpha)), libs/CRP.py:83-85."""
    return np.log(n_i, dtype=dtype) - np.log(n - 1 + alpha, dtype=dtype)


def weights_to_probs(log_w):
    """Log-weights -> probabilities with a 1e-15 floor (libs/CRP.py:88-100):
    shift by the first maximum, normalise with log1p over the others."""
    top = first_nanargmax(log_w)
    rest = np.arange(log_w.size) != top
    gap = log_w[rest] - log_w[top]
    try:
        tail = np.exp(gap)
    except FloatingPointError:
        tail = np.exp(np.clip(gap, log_EPSILON, 0))
    log_p = log_w - log_w[top] - np.log1p(seqsum(tail))
    return np.exp(np.clip(log_p, log_EPSILON, 0))


def weights_to_log_probs(log_w):
    """Log-weights -> normalised log-probabilities (libs/CRP.py:103-116); a
    trapped underflow collapses a PAIR to (0, log 1e-15)."""
    top = first_nanargmax(log_w)
    rest = np.arange(log_w.size) != top
    try:
        return log_w - log_w[top] \
            - np.log1p(seqsum(np.exp(log_w[rest] - log_w[top])))
    except FloatingPointError:
        if log_w[0] > log_w[1]:
            return np.array([0, log_EPSILON])
        return np.array([log_EPSILON, 0])


class Likelihood:
    """Needs: data, parameters, assignment, cells_per_cluster, CRP_prior,
    FP, FN, _beta_mix_const, DP_a, DP_a_prior, param_prior,
    beta_prior_uniform."""

    log_CRP_prior = staticmethod(crp_log_weight)
    _normalize_log_probs = staticmethod(weights_to_probs)
    _normalize_log = staticmethod(weights_to_log_probs)

    # emission of an observation given the true genotype, libs/CRP.py:207-212
    def _Bernoulli_FN(self, x):
        """genotype 1: observed 1 w.p. 1-FN, observed 0 w.p. FN"""
        return (1 - self.FN) ** x * self.FN ** (1 - x)

    def _Bernoulli_FP(self, x):
        """genotype 0: observed 1 w.p. FP, observed 0 w.p. 1-FP"""
        return (1 - self.FP) ** (1 - x) * self.FP ** x

    def _calc_ll(self, x, theta, flat=False):
        """libs/CRP.py:197-204.  x (r, M) against theta (K, M) or (M,);
        NaN observations drop out of the sum."""
        per_element = np.log(theta * self._Bernoulli_FN(x)
            + (1 - theta) * self._Bernoulli_FP(x))
        return seqsum(per_element) if flat else seqsum(per_element, axis=1)

    def _cluster_sizes(self):
        return np.fromiter(self.cells_per_cluster.values(), dtype=int)

    def _cluster_ids(self):
        return np.fromiter(self.cells_per_cluster.keys(), dtype=int)

    def get_lpost_single(self, cell_id, cl_ids):
        """One cell against the given clusters, + their CRP weights
        (libs/CRP.py:223-227; sizes in dict order)."""
        row = self.data[[cell_id]]
        return self._calc_ll(row, self.parameters[cl_ids]) \
            + self.CRP_prior[self._cluster_sizes()]

    def get_lpost_single_new_cluster(self):
        """Every cell against a not-yet-existing cluster whose profile is
        integrated out under the Beta prior (libs/CRP.py:230-234)."""
        mix0, mix1 = self._beta_mix_const
        wt = mix0 * self._Bernoulli_FP(self.data)
        mut = mix1 * self._Bernoulli_FN(self.data)
        return seqsum(np.log(mut + wt), axis=1) + self.CRP_prior[-1]

    def get_ll_full(self):
        """libs/CRP.py:237-238"""
        return self._calc_ll(self.data, self.parameters[self.assignment],
            flat=True)

    def get_lprior_full(self):
        """libs/CRP.py:241-251"""
        total = self.DP_a_prior.logpdf(self.DP_a) \
            + seqsum(self.CRP_prior[self._cluster_sizes()])
        if not self.beta_prior_uniform:
            live = self.parameters[self._cluster_ids()]
            total += seqsum(self.param_prior.logpdf(live))
        return total
