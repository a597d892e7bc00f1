"""TEST INFRASTRUCTURE - Metropolis-Hastings updates of the cluster profiles
in the CPU oracle."""
import numpy as np
from scipy.stats import truncnorm

from .constants import TMAX, TMIN
from .seqsum import seqsum


def _support_bounds(centre, std, lo=TMIN, hi=TMAX):
    """Standardised truncation bounds of a proposal centred at `centre`."""
    return (lo - centre) / std, (hi - centre) / std


class ParameterMoves:

    def update_parameters(self, step_no=None):
        """Every populated cluster in dict order (libs/CRP.py:302-311).
        Returns (#declined, #accepted) over all clusters x mutations."""
        rejected = []
        for cl in self.cells_per_cluster:
            members = np.argwhere(self.assignment == cl).flatten()
            self.parameters[cl], _, n_rej = self.MH_cluster_params(
                self.parameters[cl], members)
            rejected.append(n_rej)
        rejected = np.array(rejected, dtype=int)
        return rejected.sum(), (self.muts_total - rejected).sum()

    def MH_cluster_params(self, old_params, cells, trans_prob=False):
        """One independent MH proposal per mutation (libs/CRP.py:314-344).
        Stream: choice(sd, M) -> truncnorm.rvs (M uniforms) -> random(M)."""
        M = self.muts_total
        std = np.random.choice(self.param_proposal_sd, size=M)
        a, b = _support_bounds(old_params, std)
        proposal = truncnorm.rvs(a, b, loc=old_params, scale=std, size=M) \
            .astype(np.float32)

        A = self._get_log_A(proposal, old_params, cells, a, b, std,
            trans_prob)
        rejected = np.log(np.random.random(M)) >= A
        proposal[rejected] = old_params[rejected]

        if not trans_prob:
            return proposal, np.nan, rejected.sum()
        # log-probability of the move that was actually made, per mutation
        A[rejected] = np.log(-1 * np.expm1(A[rejected]))
        return proposal, seqsum(A), rejected.sum()

    def _get_log_A(self, new_params, old_params, cells, a, b, std, clip=False):
        """log acceptance ratio per mutation (libs/CRP.py:347-383):
        likelihood of the subset's columns + Beta prior + proposal densities
        (reverse - forward)."""
        forward = truncnorm.logpdf(new_params, a, b, loc=old_params, scale=std)
        a_back, b_back = _support_bounds(new_params, std)
        backward = truncnorm.logpdf(old_params, a_back, b_back,
            loc=new_params, scale=std)

        x = self.data[cells]
        mut = self._Bernoulli_FN(x)
        wt = self._Bernoulli_FP(x)
        ll_new = seqsum(np.log(new_params * mut + (1 - new_params) * wt),
            axis=0)
        ll_old = seqsum(np.log(old_params * mut + (1 - old_params) * wt),
            axis=0)

        if self.beta_prior_uniform:
            prior_new = prior_old = 0
        else:
            prior_new = self.param_prior.logpdf(new_params)
            prior_old = self.param_prior.logpdf(old_params)

        A = ll_new + prior_new - ll_old - prior_old + backward - forward
        return np.clip(A, a_min=None, a_max=0) if clip else A
