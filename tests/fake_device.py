"""A NumPy stand-in for bnpc_amd._lib.Context, for CPU-only tests of the HOST
logic of the product (model surface, RNG order, native sequential sweeps).

Test infrastructure: it computes the four device primitives with the oracle's
arithmetic.  The product never imports this; GPU parity proper is in
tests/test_gpu_parity.py (-m gpu)."""
import numpy as np

from oracle import crp_numpy as O
from bnpc_amd import _lib


class FakeContext:
    def __init__(self, data=None, codes=None, device=0):
        self.data = np.asarray(data, dtype=np.float64)
        self.N, self.M = self.data.shape
        self.views = {0: np.arange(self.N)}
        self.lab = None
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def close(self):
        pass

    def cell_counts(self):
        return (np.nansum(self.data == 1, axis=1).astype(np.int32),
            np.nansum(self.data == 0, axis=1).astype(np.int32))

    def view_set(self, view, cells):
        self._count('view_set')
        self.views[view] = np.asarray(cells, dtype=np.int64).copy()
        return self.views[view].size

    def view_set_slot(self, view, cells, slot):
        assert slot not in getattr(self, 'tiles', {})
        return self.view_set(view, cells)

    def view_size(self, view):
        return self.views[view].size

    def _sum_tables(self, view, L1, L0):
        x = self.data[self.views[view]]
        out = np.empty((x.shape[0], L1.shape[0]))
        for k in range(L1.shape[0]):
            el = np.where(x == 1, L1[k], np.where(x == 0, L0[k], np.nan))
            out[:, k] = O.seqsum(el, axis=1) if x.shape[0] else 0
        return out

    def ll_theta(self, view, theta, FP, FN, out=None, fetch=True):
        self._count('ll_theta')
        theta = np.asarray(theta, dtype=np.float32)
        if theta.ndim == 1:
            theta = theta[None, :]
        t64 = theta.astype(np.float64)
        om64 = (1 - theta).astype(np.float64)
        L1 = np.log(t64 * (1 - FN) + om64 * FP)
        L0 = np.log(t64 * FN + om64 * (1 - FP))
        res = self._sum_tables(view, L1, L0)
        if out is not None:
            out[:, :res.shape[1]] = res
            return out
        return res

    def ll_theta_pinned(self, view, theta, FP, FN, ld):
        n = self.views[view].size
        out = np.empty((n, ld))
        return self.ll_theta(view, theta, FP, FN, out=out)

    def ll_theta_pinned_top2(self, view, theta, FP, FN, ld, col_prior,
                wait=True):
        mat = self.ll_theta_pinned(view, theta, FP, FN, ld)
        K = np.asarray(theta).shape[0]
        from bnpc_amd._lib import hints_from_matrix, HINT_COLS_MAX
        if K > HINT_COLS_MAX:
            return mat, None
        hint = hints_from_matrix(mat[:, :K], col_prior)
        return mat, hint

    def ll_theta_pinned_sums_issue(self, view, theta, FP, FN, ld, col_prior):
        mat = self.ll_theta_pinned(view, theta, FP, FN, ld)
        self._sums = (mat, np.asarray(theta).shape[0],
            np.array(col_prior, dtype=np.float64))
        return mat

    def hints_in_order_issue(self, order):
        from bnpc_amd._lib import hints_from_matrix, HINT_COLS_MAX
        mat, K, col_prior = self._sums
        self._sums = None
        if K > HINT_COLS_MAX:
            return None
        order = np.asarray(order, dtype=np.int64)
        return np.ascontiguousarray(
            hints_from_matrix(mat[:, :K], col_prior)[order])

    def matrix_wait(self):
        pass

    def ll_total_issue(self, theta, FP, FN):
        self._pending_total = self.ll_total(theta, FP, FN)

    def ll_total_wait(self):
        out, self._pending_total = self._pending_total, None
        return out

    def theta_put(self, row0, theta):
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float32))
        self._poison_in_flight(row0, theta.shape[0])
        store = getattr(self, 'store', np.zeros((0, self.M), np.float32))
        need = row0 + theta.shape[0]
        if need > store.shape[0]:
            store = np.concatenate([store,
                np.zeros((need - store.shape[0], self.M), np.float32)])
        store[row0:need] = theta
        self.store = store

    def ll_rows_issue(self, view, rows, FP, FN, ld, slot):
        self._count('ll_rows_issue')
        tiles = self.__dict__.setdefault('tiles', {})
        assert slot in (0, 1, 2, 3) and slot not in tiles
        tiles[slot] = (np.asarray(rows).copy(),
            self.ll_rows_pinned(view, rows, FP, FN, ld))

    def ll_rows_wait(self, slot, n_rows, ld):
        _, out = self.tiles.pop(slot)
        assert out.shape == (n_rows, ld)
        self.__dict__.setdefault('tile_hints', {}).pop(slot, None)
        return out

    def ll_rows_issue_hint(self, view, rows, FP, FN, ld, slot, col_prior):
        """The tile with its hints, computed from the matrix AS ISSUED (a
        column poisoned later, because its id was re-used, keeps the hint it
        had - as on the device)."""
        self.ll_rows_issue(view, rows, FP, FN, ld, slot)
        self._count('ll_rows_issue_hint')
        hints = self.__dict__.setdefault('tile_hints', {})
        hints[slot] = _lib.wide_hints_from_matrix(self.tiles[slot][1],
            col_prior)

    def ll_rows_wait_hint(self, slot, n_rows, ld):
        hint = self.__dict__.setdefault('tile_hints', {}).get(slot)
        return self.ll_rows_wait(slot, n_rows, ld), hint

    def _poison_in_flight(self, row0, n):
        """A parameter row rewritten while a tile is in flight makes that
        tile's column for the id meaningless: NaN it, so that a model reading
        it cannot go unnoticed."""
        for rows, out in getattr(self, 'tiles', {}).values():
            hit = np.flatnonzero((rows >= row0) & (rows < row0 + n))
            out[:, hit] = np.nan
            self.poisoned = getattr(self, 'poisoned', 0) + hit.size

    def ll_rows_pinned(self, view, rows, FP, FN, ld):
        return self.ll_theta_pinned(view, self.store[np.asarray(rows)], FP,
            FN, ld)

    def ll_tables(self, view, L1, L0, out=None):
        self._count('ll_tables')
        L1 = np.atleast_2d(np.asarray(L1, dtype=np.float64))
        L0 = np.atleast_2d(np.asarray(L0, dtype=np.float64))
        return self._sum_tables(view, L1, L0)

    def colcounts(self, segments):
        self._count('colcounts')
        G = len(segments)
        n1 = np.zeros((G, self.M), dtype=np.int32)
        n0 = np.zeros((G, self.M), dtype=np.int32)
        for g, seg in enumerate(segments):
            sub = self.data[np.asarray(seg, dtype=np.int64)]
            n1[g] = (sub == 1).sum(axis=0)
            n0[g] = (sub == 0).sum(axis=0)
        return n1, n0

    def view_counts(self, view, labels, G):
        self._count('view_counts')
        labels = np.asarray(labels)
        cells = self.views[view]
        assert labels.size == cells.size
        n1, n0 = self.colcounts([cells[labels == g] for g in range(G)])
        self.calls['colcounts'] -= 1
        return n1, n0

    def reload_options(self):
        pass

    def colcounts_by_label(self, assignment, ids, fetch=True):
        self._count('colcounts_by_label')
        assignment = np.asarray(assignment)
        segs = [np.flatnonzero(assignment == i) for i in ids]
        n1, n0 = self.colcounts(segs)
        self.lab = (n1, n0)
        return n1, n0

    def ll_total(self, theta, FP, FN):
        self._count('ll_total')
        theta = np.asarray(theta, dtype=np.float32)
        n1, n0 = self.lab
        assert theta.shape == n1.shape
        t64 = theta.astype(np.float64)
        om64 = (1 - theta).astype(np.float64)
        out = []
        for fp, fn in zip(np.atleast_1d(FP), np.atleast_1d(FN)):
            L1 = np.log(t64 * (1 - fn) + om64 * fp)
            L0 = np.log(t64 * fn + om64 * (1 - fp))
            out.append((n1 * L1 + n0 * L0).sum())
        return np.array(out)

    def sync(self):
        pass

    def hints_wait(self):
        pass


def attach(model):
    """Give a bnpc_amd.model instance a FakeContext instead of a GPU."""
    model._ctx = FakeContext(data=model.data)
    return model


class FakePosterior:
    """NumPy stand-in for _lib.Posterior (bnpc_post) on the CPU."""

    def __init__(self, assignments, device=None):
        from scipy.spatial.distance import pdist
        a = np.asarray(assignments)
        self.S, self.N = a.shape
        self._differ = np.zeros(self.N * (self.N - 1) // 2, dtype=np.int32)
        for row in a:
            self._differ += pdist(np.stack([row, row]).T, 'hamming') \
                .astype(np.int32)
        self.differ_sum = int(self._differ.astype(np.int64).sum())

    def differ(self):
        return self._differ.copy()

    def dist(self):
        return self._differ / self.S

    def ward(self):
        from scipy.cluster.hierarchy import linkage
        return linkage(self.dist(), method='ward')

    def mpear_sums(self, labels):
        from scipy.spatial.distance import pdist
        out = np.empty(len(labels), dtype=np.int64)
        for c, lab in enumerate(np.asarray(labels)):
            same = pdist(np.stack([lab, lab]).T, 'hamming') == 0
            out[c] = self._differ[same].astype(np.int64).sum()
        return out

    def close(self):
        pass
