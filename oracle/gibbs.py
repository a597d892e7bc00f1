"""TEST INFRASTRUCTURE - assignment sweep and concentration update of the CPU
oracle."""
import numpy as np

from .constants import EPSILON


class GibbsMoves:

    def _take_out(self, cell):
        home = self.assignment[cell]
        if self.cells_per_cluster[home] == 1:
            del self.cells_per_cluster[home]
        else:
            self.cells_per_cluster[home] -= 1

    def _put_in(self, cell, cluster):
        self.assignment[cell] = cluster
        self.cells_per_cluster[cluster] = \
            self.cells_per_cluster.get(cluster, 0) + 1

    def update_assignments_Gibbs(self):
        """libs/CRP.py:254-288: cells in permuted order; each is removed,
        scored against the populated clusters (dict order) and a new one,
        and re-drawn with one `choice`."""
        lpost_new = self.get_lpost_single_new_cluster()
        for cell in np.random.permutation(self.cells_total):
            self._take_out(cell)
            ids = self._cluster_ids()
            lpost = np.append(self.get_lpost_single(cell, ids),
                lpost_new[cell])
            drawn = np.random.choice(np.append(ids, -1),
                p=self._normalize_log_probs(lpost))
            if drawn == -1:
                drawn = self.init_new_cluster(cell)
            self._put_in(cell, drawn)

    def init_new_cluster(self, cell_id):
        """libs/CRP.py:291-294"""
        slot = self.get_empty_cluster()
        self.parameters[slot] = self._init_cl_params_new([cell_id])
        return slot

    def get_empty_cluster(self):
        """Lowest id without cells (libs/CRP.py:297-299)."""
        slot = 0
        while slot in self.cells_per_cluster:
            slot += 1
        return slot

    def update_DP_alpha(self):
        """Escobar & West (1995) auxiliary-variable draw, libs/CRP.py:386-410."""
        k = len(self.cells_per_cluster)
        n = self.cells_total
        shape0, rate0 = self.DP_a_gamma
        eta = np.random.beta(self.DP_a + 1, n)
        odds = (shape0 + k - 1) / (n * (rate0 - np.log(eta)))
        if np.random.random() < odds / (1 + odds):
            draw = np.random.gamma(shape0 + k, rate0 - np.log(eta))
        else:
            draw = np.random.gamma(shape0 + k - 1, rate0 - np.log(eta))
        self.DP_a = max(1 + EPSILON, draw)
        self.init_DP_prior()
