"""TEST INFRASTRUCTURE - Jain & Neal (2007) non-conjugate split-merge with
restricted Gibbs scans, as the reference runs it (libs/CRP.py:417-820;
SURVEY.md section 8(a) row a10).

Launch state of a move over `cells` (anchor i first, anchor j last, the
others S = cells[1:-1] between them):

    rg_assignment      (|S|,) 0/1: side of every non-anchor cell
    rg_params_split    (2, M) float32: profiles of the two sides
    rg_params_merge    (M,)   float32: profile of the merged cluster
"""
import numpy as np
from scipy.special import gammaln

from .constants import TMAX, TMIN
from .seqsum import seqsum

REJECTED = [0, 1]
ACCEPTED = [1, 0]


def _swap(v, i, j):
    v[i], v[j] = v[j], v[i]


class SplitMergeMoves:

    # ----------------------------------------------------------- the move
    def update_assignments_split_merge(self, ratios=[.75, .25], step_no=5):
        """libs/CRP.py:417-431 -> ([accepted, declined], 0=split | 1=merge).
        The kind is forced when only one is possible (no draw then)."""
        K = len(self.cells_per_cluster)
        if K == 1:
            kind = 0
        elif K == self.cells_total:
            kind = 1
        else:
            kind = np.random.choice([0, 1], p=ratios)
        mover = self.do_split_move if kind == 0 else self.do_merge_move
        return (mover(step_no), kind)

    def do_split_move(self, step_no=5):
        """libs/CRP.py:434-481: cluster ~ size (redrawn while a singleton),
        two distinct anchors, restricted scans, MH."""
        ids = self._cluster_ids()
        sizes = self._cluster_sizes()
        pick_p = sizes / sizes.sum()
        while True:
            cl = np.random.choice(ids, p=pick_p)
            cells = np.argwhere(self.assignment == cl).flatten()
            if cells.size != 1:
                break
        first, last = np.random.choice(cells.size, size=2, replace=False)
        _swap(cells, 0, first)
        _swap(cells, -1, last)

        where = np.argwhere(ids == cl).flatten()
        n_cl = self.cells_per_cluster[cl]
        log_pick = np.log(pick_p[where]) - np.log(n_cl) - np.log(n_cl - 1)
        size_data = (log_pick, np.delete(sizes, where))

        ok, sides, profiles = self.run_rg_nc('split', cells, size_data,
            step_no)
        if not ok:
            return REJECTED
        fresh = self.get_empty_cluster()
        self.parameters[cl] = profiles[0]
        self.parameters[fresh] = profiles[1]
        leaving = np.append(cells[1:-1][np.where(sides == 1)], cells[-1])
        self.assignment[leaving] = fresh
        self.cells_per_cluster[cl] -= leaving.size
        self.cells_per_cluster[fresh] = leaving.size
        return ACCEPTED

    def do_merge_move(self, step_no=5):
        """libs/CRP.py:484-524: two clusters ~ 1/size without replacement,
        one anchor from each, restricted scans, MH."""
        ids = self._cluster_ids()
        sizes = self._cluster_sizes()
        inverse = 1 / sizes
        pick_p = inverse / inverse.sum()
        keep, gone = np.random.choice(ids, p=pick_p, size=2, replace=False)

        cells_keep = np.argwhere(self.assignment == keep).flatten()
        _swap(cells_keep, 0, np.random.choice(cells_keep.size))
        cells_gone = np.argwhere(self.assignment == gone).flatten()
        _swap(cells_gone, -1, np.random.choice(cells_gone.size))
        cells = np.concatenate((cells_keep, cells_gone)).flatten()

        where = np.argwhere((ids == gone) | (ids == keep)).flatten()
        size_data = seqsum(np.log(pick_p[where])) \
            - seqsum(np.log(sizes[where]))

        ok, profile = self.run_rg_nc('merge', cells, size_data, step_no)
        if not ok:
            return REJECTED
        self.parameters[keep] = profile
        self.assignment[cells_gone] = keep
        self.cells_per_cluster[keep] += cells_gone.size
        del self.cells_per_cluster[gone]
        return ACCEPTED

    def run_rg_nc(self, move, cells, size_data, scan_no):
        """Launch state + `scan_no` intermediate scans of both the split and
        the merged configuration, then the final MH (libs/CRP.py:527-544)."""
        self._rg_init_split(cells)
        self.rg_params_merge = self._init_cl_params_new(cells)
        for _ in range(scan_no):
            self._rg_scan_split(cells)
            self._rg_scan_merge(cells)
        if move == 'split':
            return self._do_rg_split_MH(cells, size_data)
        return self._do_rg_merge_MH(cells, size_data)

    # -------------------------------------------------------- launch state
    def _rg_members(self, cells, side):
        """Cells on `side` (0: with anchor i, 1: with anchor j), anchor last."""
        anchor = cells[0] if side == 0 else cells[-1]
        inner = cells[1:-1]
        return np.append(inner[np.argwhere(self.rg_assignment == side)],
            anchor)

    def _rg_init_split(self, cells, random=False):
        """libs/CRP.py:547-567: each non-anchor cell joins the anchor whose
        own data row (NaN -> prior weight) explains it better; ties -> i."""
        inner = cells[1:-1]
        if inner.size == 0:
            self.rg_assignment = np.array([])
        elif random:
            self.rg_assignment = np.random.choice([0, 1], size=(inner.size))
        else:
            fill = self._beta_mix_const[0]
            as_profile = [np.nan_to_num(self.data[anchor], nan=fill)
                for anchor in (cells[0], cells[-1])]
            ll_i = self._calc_ll(self.data[inner], as_profile[0])
            ll_j = self._calc_ll(self.data[inner], as_profile[1])
            self.rg_assignment = np.where(ll_j > ll_i, 1, 0)
        self.rg_params_split = np.stack([
            self._init_cl_params_new(self._rg_members(cells, side))
            for side in (0, 1)])

    # --------------------------------------------------------------- scans
    def _rg_scan_split(self, cells, trans_prob=False):
        """libs/CRP.py:570-578"""
        lp_sides = self._rg_scan_assign(cells, trans_prob) \
            if cells.size != 2 else 0
        lp_profiles = self._rg_scan_params(cells, trans_prob)
        if trans_prob:
            return lp_sides + lp_profiles

    def _rg_scan_merge(self, cells, trans_prob=False):
        """libs/CRP.py:581-587"""
        self.rg_params_merge, lp, _ = self.MH_cluster_params(
            self.rg_params_merge, cells, trans_prob)
        if trans_prob:
            return lp

    def _rg_scan_params(self, cells, trans_prob=False):
        """libs/CRP.py:590-606"""
        lp = np.zeros(2)
        for side in (0, 1):
            self.rg_params_split[side], lp[side], _ = self.MH_cluster_params(
                self.rg_params_split[side], self._rg_members(cells, side),
                trans_prob)
        if trans_prob:
            return lp.sum()

    def _rg_two_way(self, ll_pair, n):
        """Normalised log-probabilities of the two sides for the cell whose
        entry of rg_assignment is currently -1."""
        n_j = seqsum(self.rg_assignment) + 2
        n_i = n - n_j - 1
        return self._normalize_log(
            ll_pair + self.log_CRP_prior([n_i, n_j], n, self.DP_a))

    def _rg_scan_assign(self, cells, trans_prob=False):
        """Sequential restricted Gibbs scan over the non-anchor cells in
        permuted order (libs/CRP.py:609-632)."""
        n = cells.size
        ll = self._rg_get_ll(cells[1:-1], self.rg_params_split)
        taken = np.zeros(n - 2) if trans_prob else None
        for pos in np.random.permutation(n - 2):
            self.rg_assignment[pos] = -1
            log_p = self._rg_two_way(ll[pos], n)
            side = np.random.choice([0, 1], p=np.exp(log_p))
            self.rg_assignment[pos] = side
            if trans_prob:
                taken[pos] = log_p[side]
        if trans_prob:
            return seqsum(taken)

    def _rg_get_ll(self, cells, params):
        """(|cells|, 2): every cell under the two profiles, libs/CRP.py:635-638"""
        x = self.data[cells]
        return np.stack([self._calc_ll(x, params[0]),
            self._calc_ll(x, params[1])], axis=1)

    # ------------------------------------------------------------ final MH
    def _do_rg_split_MH(self, cells, size_data):
        """libs/CRP.py:641-653 (the ratio is evaluated - and consumes the
        stream - even when the launch state is degenerate)."""
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
        """libs/CRP.py:656-665"""
        A = self._get_trans_prob_ratio_merge(cells) \
            + self._get_lprior_ratio_merge(cells) \
            + self._get_ll_ratio(cells, 'merge') \
            + self._get_ltrans_prob_size_ratio_merge(size_data)
        if np.log(np.random.random()) < A:
            return (True, self.rg_params_merge)
        return (False, [])

    def _side_counts(self):
        """(n_i, n_j) of the launch state, anchors included."""
        n_j = seqsum(self.rg_assignment) + 1
        return self.rg_assignment.size + 2 - n_j, n_j

    def _get_trans_prob_ratio_split(self, cells):
        """Jain & Neal eq. 15, libs/CRP.py:668-682"""
        lp_split = self._rg_scan_split(cells, trans_prob=True)
        std = np.random.choice(self.param_proposal_sd, size=self.muts_total)
        a = (TMIN - self.rg_params_merge) / std
        b = (TMAX - self.rg_params_merge) / std
        original = self.parameters[self.assignment[cells[0]]]
        lp_merge = seqsum(self._get_log_A(original, self.rg_params_merge,
            cells, a, b, std, True))
        return lp_merge - lp_split

    def _get_trans_prob_ratio_merge(self, cells):
        """Jain & Neal eq. 16, libs/CRP.py:685-692"""
        lp_merge = self._rg_scan_merge(cells, trans_prob=True)
        lp_split = self._rg_get_split_prob(cells)
        return lp_split - lp_merge

    def _get_lprior_ratio_split(self, cells):
        """Jain & Neal eq. 7, libs/CRP.py:695-713 (the reference pairs the
        n_i test with gammaln(n_j) and vice versa - kept)."""
        n_i, n_j = self._side_counts()
        ratio = np.log(self.DP_a) - gammaln(self.rg_assignment.size + 2)
        if n_i > 0:
            ratio += gammaln(n_j)
        if n_j > 0:
            ratio += gammaln(n_i)
        if not self.beta_prior_uniform:
            before = self.parameters[self.assignment[cells[0]]]
            ratio += seqsum(self.param_prior.logpdf(self.rg_params_split)) \
                - seqsum(self.param_prior.logpdf(before))
        return ratio

    def _get_lprior_ratio_merge(self, cells):
        """Jain & Neal eq. 8, libs/CRP.py:736-754"""
        n = cells.size
        n_j = seqsum(self.rg_assignment) + 1
        n_i = n - n_j
        ratio = gammaln(n) - np.log(self.DP_a)
        if n_i > 0:
            ratio -= gammaln(n_i)
        if n_j > 0:
            ratio -= gammaln(n_j)
        if not self.beta_prior_uniform:
            before = self.parameters[self.assignment[[cells[0], cells[-1]]]]
            ratio += seqsum(self.param_prior.logpdf(self.rg_params_merge)) \
                - seqsum(self.param_prior.logpdf(before))
        return ratio

    def _get_ll_ratio(self, cells, move):
        """Jain & Neal eq. 11/12, libs/CRP.py:716-733"""
        inner = cells[1:-1]
        side_i = np.append(inner[np.argwhere(self.rg_assignment == 0)],
            cells[0])
        side_j = np.append(inner[np.nonzero(self.rg_assignment)], cells[-1])
        ll_i = self._calc_ll(self.data[side_i], self.rg_params_split[0], True)
        ll_j = self._calc_ll(self.data[side_j], self.rg_params_split[1], True)
        ll_one = self._calc_ll(self.data[cells], self.rg_params_merge, True)
        if move == 'split':
            return ll_i + ll_j - ll_one
        return ll_one - ll_i - ll_j

    def _get_ltrans_prob_size_ratio_split(self, ltrans_prob_size, cluster_size):
        """Probability of choosing this pair for the reverse merge over the
        probability of having chosen this cluster (libs/CRP.py:757-764)."""
        n_i, n_j = self._side_counts()
        norm = seqsum(1 / np.append(cluster_size, [n_i, n_j]))
        back = np.log(1 / n_i / norm) + np.log(1 / n_j / norm)
        return back - ltrans_prob_size[0]

    def _get_ltrans_prob_size_ratio_merge(self, trans_prob_size):
        """libs/CRP.py:767-774; log(0) and log(-1) are trapped."""
        back = -np.log(self.cells_total)
        try:
            back = back - np.log(self.rg_assignment.size - 1)
        except FloatingPointError:
            pass
        return back - trans_prob_size

    def _rg_get_split_prob(self, cells):
        """log-probability of moving from the launch state to the ORIGINAL
        two clusters (libs/CRP.py:777-820).  Bounds of the proposal density
        are (0, 1) here, and rg_assignment is overwritten with the original
        sides as the scan proceeds."""
        std = np.random.choice(self.param_proposal_sd,
            size=(2, self.muts_total))
        a = (0 - self.rg_params_split) / std
        b = (1 - self.rg_params_split) / std

        inner = cells[1:-1]
        homes = (self.assignment[cells[0]], self.assignment[cells[-1]])
        lp_profiles = [
            seqsum(self._get_log_A(self.parameters[homes[side]],
                self.rg_params_split[side], self._rg_members(cells, side),
                a[side], b[side], std[side], True))
            for side in (0, 1)]

        ll = self._rg_get_ll(inner,
            (self.parameters[homes[0]], self.parameters[homes[1]]))
        original = np.where(self.assignment[inner] == homes[0], 0, 1)
        lp_sides = np.zeros(inner.size)
        for pos in range(inner.size):
            self.rg_assignment[pos] = -1
            log_p = self._rg_two_way(ll[pos], cells.size)
            self.rg_assignment[pos] = original[pos]
            lp_sides[pos] = log_p[original[pos]]
        return lp_profiles[0] + lp_profiles[1] + seqsum(lp_sides)
