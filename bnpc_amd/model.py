"""GPU-backed mirror of the reference's model classes.

Same public surface, attributes, return values and consumption order of the
global legacy ``np.random`` stream as

    CRP                   /root/reference/libs/CRP.py:17-820
    CRP_errors_learning   /root/reference/libs/CRP_learning_errors.py:17-111

so that the sampler driver (libs/MCMC.py, or bnpc_amd.mcmc) runs unchanged on
top.  Underneath, every cells x mutations (x clusters) array expression of the
reference is one of four device primitives of libbnpc_hip.so
(include/bnpc_hip.h):

    ll_theta / ll_tables   per-cell log-likelihood sums  (_calc_ll, axis=1)
    colcounts(_by_label)   per-subset counts of 1s / 0s per mutation; every
                           sum over CELLS of log(theta*P1 + (1-theta)*P0) is
                           then n1*L1(theta) + n0*L0(theta), O(M) on the host
    ll_total               the flat total (get_ll_full, get_ll_full_error)
    gibbs_sweep / rg_scan  the sequential per-cell loops, native, on an exact
                           replica of NumPy's legacy MT19937 stream

Host NumPy arrays stay the source of truth for ``assignment``, ``parameters``
and ``cells_per_cluster`` (the driver reads them, MCMC.py:255-282); the
device context is created lazily in the worker process (after fork) and is
dropped on pickle/deepcopy (MCMC.py:115-128).

There is no CPU fallback: without the HIP library / a GPU the first device
call raises RuntimeError.
"""
import ctypes as C
import os

import numpy as np
from scipy.special import gamma as _gamma_fn, gammaln
from scipy.stats import beta as _beta_dist, truncnorm
from scipy.stats import gamma as _gamma_dist

from bnpc_amd import _lib, fastdist, hostkernels

# the reference traps these and uses FloatingPointError as control flow
# (libs/CRP.py:10)
np.seterr(divide='raise', over='ignore', under='ignore', invalid='raise')

EPSILON = np.finfo(np.float64).resolution
TMIN = 1e-5
TMAX = 1 - TMIN
log_EPSILON = np.log(EPSILON)

_THREAD_MIN_ELEMS = 1 << 14     # below this NumPy call overhead dominates
_ELEMS_PER_THREAD = 8192
_POOL = {}


def _host_parts(elems, rows):
    """Row blocks to evaluate an elementwise host batch in.  Measured on the
    GPU box's host (tools/host_threads_ab.py): ~8k elements per thread is the
    break-even, more than 4 threads only pay for >= 128k elements (the GIL is
    held between ufunc calls)."""
    if elems < _THREAD_MIN_ELEMS or rows < 2:
        return 1
    parts = 8 if elems >= (1 << 17) \
        else min(4, max(1, elems // _ELEMS_PER_THREAD))
    return min(parts, rows)


def _host_pool():
    """Thread pool for large elementwise host batches (BNPC_HOST_THREADS,
    default min(8, cores); 1 disables).  Created per process, after fork."""
    pid = os.getpid()
    if _POOL.get('pid') != pid:
        n = int(os.environ.get('BNPC_HOST_THREADS',
            min(8, os.cpu_count() or 1)))
        from concurrent.futures import ThreadPoolExecutor
        _POOL.clear()
        _POOL.update(pid=pid,
            pool=ThreadPoolExecutor(max_workers=n) if n > 1 else None)
    return _POOL['pool']


def _rowwise(fn, x):
    """fn(x) for an elementwise `fn` over a 2-D array, row blocks on the host
    threads when the array is large (same values: elementwise)."""
    parts = _host_parts(x.size, x.shape[0])
    pool = _host_pool() if parts > 1 else None
    if pool is None:
        return fn(x)
    parts = min(parts, pool._max_workers)
    edges = np.linspace(0, x.shape[0], parts + 1, dtype=int)
    return np.concatenate(list(pool.map(lambda lo_hi: fn(x[lo_hi[0]:lo_hi[1]]),
        zip(edges[:-1], edges[1:]))))


# spare columns of an issued tile: clusters (re)born before / while it is walked
# (running out means copying the tile - 256 MiB - into a wider matrix; with two
# tiles in flight a tile also receives the births of the two walked before it:
# 50 at config 5's start)
_TILE_SPARE = 160
_TILES_AHEAD = 2        # tiles of a tiled sweep in flight while one is walked

VIEW_ALL = 0      # identity view: all cells
VIEW_MOVE = 1     # non-anchor cells of the current split/merge move
VIEW_ONE = 2      # a single cell (get_lpost_single)
VIEW_SWEEP = 3    # a permutation-ordered tile of a tiled Gibbs sweep


# -- native batch of the parameter moves (csrc/bnpc_hostmath.cpp) -------------
_NATIVE = {}


def _native_kernels():
    """The host kernel table if the native batch may be used in this process:
    BNPC_NATIVE_MH != 0, the table could be assembled, and a batch covering
    every branch (central / upper-clip / lower-clip intervals, both priors,
    plain and transition-probability mode, the prior cache, 1 and several
    threads) came out bit-identical to the SciPy-level evaluation
    (CRP._mh_math).  Otherwise None: the SciPy-level path is used."""
    pid = os.getpid()
    if _NATIVE.get('pid') != pid:
        _NATIVE.clear()
        _NATIVE['pid'] = pid
        _NATIVE['table'] = None
        if _lib.env('BNPC_NATIVE_MH', '1') != '0':
            table = hostkernels.table()
            if table is not None and _native_batch_is_exact(table):
                _NATIVE['table'] = table
    return _NATIVE['table']


def _native_batch_is_exact(table):
    try:
        rng = np.random.RandomState(20240)
        G, M = 5, 389
        probe = CRP.__new__(CRP)
        probe.param_proposal_sd = np.array([0.1, 0.25, 0.5])
        ok = True
        for (p, q), (FP, FN) in (((.25, .25), (0.01, 0.2)),
                ((1, 1), (0.001, 0.1)), ((.75, 2.0), (1e-15, 1e-15))):
            probe.p, probe.q, probe.FP, probe.FN = p, q, FP, FN
            probe.beta_prior_uniform = bool(p == q == 1)
            old = np.clip(rng.uniform(size=(G, M)), TMIN, TMAX) \
                .astype(np.float32)
            old[:, :40] = np.float32(TMIN)      # a == 0: mirrored ppf form
            old[:, 40:80] = np.float32(TMAX)    # b == 0: mass left of zero
            old[:, 80:90] = np.float32(0.5)
            n1 = rng.randint(0, 900, size=(G, M)).astype(np.int32)
            n0 = rng.randint(0, 900, size=(G, M)).astype(np.int32)
            n1[0] = n0[0] = 0                   # an empty subset
            sd_idx = rng.randint(0, 3, size=(G, M)).astype(np.int32)
            U = rng.uniform(size=(G, M))
            u = rng.uniform(size=(G, M))
            U[1, :6] = [1e-300, 1e-17, 1 - 1e-16, .5, 1e-8, 1 - 1e-9]
            std = probe.param_proposal_sd[sd_idx]
            for trans in (False, True):
                want = probe._mh_math(old, std, U, u, n1, n0, trans, None)
                known = None
                if want[3] is not None:
                    # a cache that is right for half of the entries
                    known = (old.copy(), fastdist.beta_logpdf(old, p, q))
                    known[0][:, ::2] = np.float32(0.123)
                for threads in (1, 3):
                    got = _lib.mh_batch(table, old, n1, n0,
                        probe.param_proposal_sd, TMIN, TMAX, FP, FN, p, q,
                        probe.beta_prior_uniform, trans, known=known,
                        want_prior=True, draws=(sd_idx, U, u),
                        threads=threads)
                    ok &= got[0] == 0 and np.array_equal(got[1], want[0]) \
                        and np.array_equal(got[3], want[2].sum(axis=1))
                    if trans:
                        ok &= np.array_equal(got[2],
                            np.cumsum(want[1], axis=1)[:, -1])
                    if want[3] is not None:
                        ok &= np.array_equal(got[4], want[3])
            # log acceptance ratios of given rows (split / merge ratios)
            new_rows = np.clip(rng.uniform(size=(2, M)), TMIN, TMAX) \
                .astype(np.float32)
            new_rows[:, :30] = np.float32(TMIN)
            new_rows[:, 30:60] = np.float32(TMAX)
            for fmin, fmax in ((TMIN, TMAX), (0, 1)):
                for clip in (False, True):
                    a_ = (fmin - old[:2]) / std[:2]
                    b_ = (fmax - old[:2]) / std[:2]
                    ref = probe._get_log_A(new_rows, old[:2], None, a_, b_,
                        std[:2], clip, counts=(n1[:2], n0[:2]))
                    got = _lib.log_A(table, new_rows, old[:2], std[:2],
                        n1[:2], n0[:2], fmin, fmax, TMIN, TMAX, FP, FN, p, q,
                        probe.beta_prior_uniform, clip, threads=2)
                    ok &= got is not None and np.array_equal(got[0], ref) \
                        and np.array_equal(got[1],
                            np.cumsum(ref, axis=1)[:, -1])
            # scalar truncated-normal densities (error-rate moves / priors)
            for x, lo, sc in ((0.013, 0.01, 0.005), (0.19, 0.2, 0.15),
                    (0.0004, 0.001, 0.0005), (0.2, 0.2, 0.1), (1.5, 0.2, 0.1),
                    (0.9999, 0.2, 0.05)):
                for a, b in (((0 - lo) / sc, (1 - lo) / sc),
                        ((0 - x) / sc, (1 - x) / sc)):
                    with np.errstate(all='ignore'):
                        ref = fastdist.tn_logpdf(x, a, b, lo, sc)
                    got = _lib.tn_logpdf_scalar(table, x, a, b, lo, sc)
                    ok &= got is not None and (got == ref or
                        (np.isnan(got) and np.isnan(ref)))
                for q_ in (1e-12, 0.03, 0.5, 0.97, 1 - 1e-12):
                    a, b = (0 - lo) / sc, (1 - lo) / sc
                    ref = fastdist.tn_rvs_from_uniform(q_, a, b, lo, sc)
                    got = _lib.tn_ppf_scalar(table, q_, a, b, lo, sc)
                    ok &= got is not None and bool(got == ref)
            if not probe.beta_prior_uniform:
                dens, total = _lib.beta_logpdf_f32(table, old, p, q,
                    threads=2)
                ref = fastdist.beta_logpdf(old, p, q)
                ok &= np.array_equal(dens, ref) \
                    and total == np.cumsum(ref.ravel())[-1]
        return bool(ok)
    except Exception:
        return False


_BETA = {}


def _native_beta():
    """Whether np.random.beta may be drawn natively (bnpc_mt_beta: NumPy's
    legacy sampler restated on NumPy's own stream): BNPC_NATIVE_BETA != 0 and
    a draw over every branch of the sampler - Johnk, the shape < 1 and
    shape > 1 gammas, shape == 1, with and without a cached Gaussian - came
    out bit-identical to NumPy's from the same state and left the stream and
    the cached Gaussian where NumPy leaves them.  Checked once per process;
    the stream is put back as it was."""
    pid = os.getpid()
    if _BETA.get('pid') != pid:
        _BETA.clear()
        _BETA.update(pid=pid, ok=False)
        if _lib.env('BNPC_NATIVE_BETA', '1') != '0':
            saved = np.random.get_state()
            try:
                a = np.array([.25, 1.25, 7.25, .25, 1., 1., .999, 400.5, .5,
                    3., 1e5, .25, 1.25] * 5)
                b = np.array([.25, .25, 3.25, 9.25, 1., 2., 1., .75, .5, 1.,
                    2e5, 1e4, 1.25] * 5)
                ok = True
                for pre in (0, 1):
                    np.random.seed(917 + pre)
                    if pre:
                        np.random.normal()
                    start = np.random.get_state()
                    want = np.random.beta(a, b)
                    end = np.random.get_state()
                    np.random.set_state(start)
                    got = _lib.beta(a, b)
                    now = np.random.get_state()
                    ok &= np.array_equal(want, got) and now[2:] == end[2:] \
                        and np.array_equal(now[1], end[1])
                _BETA['ok'] = bool(ok)
            except Exception:       # noqa: BLE001
                _BETA['ok'] = False
            finally:
                np.random.set_state(saved)
    return _BETA['ok']


def _tn_rvs_scalar(a, b, loc, scale):
    """truncnorm.rvs(a, b, loc=loc, scale=scale), one draw: one uniform of the
    global stream through the ppf - natively when allowed, else SciPy."""
    table = _native_kernels()
    if table is not None:
        q = np.random.uniform()
        val = _lib.tn_ppf_scalar(table, q, float(a), float(b), float(loc),
            float(scale))
        if val is None:     # same uniform, SciPy's kernel
            val = np.float64(fastdist.tn_rvs_from_uniform(q, a, b, loc, scale))
        return val
    return fastdist.tn_rvs_scalar(a, b, loc, scale)


def _tn_logpdf_scalar(x, a, b, loc, scale):
    """truncnorm.logpdf for scalar arguments: natively on SciPy's own kernels
    when the start-up comparison allows, else through SciPy."""
    table = _native_kernels()
    if table is not None:
        val = _lib.tn_logpdf_scalar(table, float(x), float(a), float(b),
            float(loc), float(scale))
        if val is not None:
            return val
    return fastdist.tn_logpdf(x, a, b, loc, scale)


# the methods a native move (bnpc_sm_move) stands for: an override of any of
# them - by a subclass or on the instance - keeps the step-by-step path
_MOVE_STEPS = ('run_rg_nc', '_propose_split', '_propose_merge', '_rg_open',
    '_rg_init_split', '_rg_scan_both', '_rg_scan_split', '_rg_scan_merge',
    '_rg_scan_fused', '_rg_scan_assign', '_rg_scan_params', '_rg_get_ll',
    '_do_rg_split_MH', '_do_rg_merge_MH', '_get_trans_prob_ratio_split',
    '_get_trans_prob_ratio_merge', '_get_lprior_ratio_split',
    '_get_lprior_ratio_merge', '_get_ll_ratio',
    '_get_ltrans_prob_size_ratio_split', '_get_ltrans_prob_size_ratio_merge',
    '_rg_get_split_prob', '_beta_draw', '_mh_batch', '_log_A_sum',
    'MH_cluster_params', '_get_log_A', 'get_empty_cluster', '_tables',
    '_subset_ll')


_MOVE_STEP_SET = frozenset(_MOVE_STEPS)
_plain_classes = {}


def _overrides_a_move_step(model):
    cls = type(model)
    plain = _plain_classes.get(cls)
    if plain is None:
        plain = _plain_classes[cls] = all(
            getattr(cls, name) is getattr(CRP, name) for name in _MOVE_STEPS)
    return not plain or not _MOVE_STEP_SET.isdisjoint(model.__dict__)


_MOVES_OK = {}


def _native_moves_allowed():
    """Whether split / merge moves may be made natively in this process
    (bnpc_sm_move restates NumPy behaviour that is not part of any interface:
    np.sum's pairwise order inside the iterator's 8192-element buffer, the
    legacy choice with p / without replacement, permutation, randint):
    np_sum equals np.sum around every change of regime and the proposals of
    both moves equal the binding's own (_propose_split / _propose_merge) from
    the same stream - compared once per process, the stream put back."""
    pid = os.getpid()
    if _MOVES_OK.get('pid') != pid:
        _MOVES_OK.clear()
        _MOVES_OK.update(pid=pid, ok=False)
        table = _native_kernels()
        if table is not None and table.gammaln:
            saved = np.random.get_state()
            try:
                rng = np.random.RandomState(77)
                ok = True
                for n in (1, 7, 8, 9, 127, 128, 129, 1000, 8192, 8193, 20001):
                    a = rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)
                    ok &= _lib.np_sum(a) == float(np.sum(a))
                probe = CRP.__new__(CRP)
                for trial in range(6):
                    K = (2, 3, 5, 9, 14, 40)[trial]
                    N = 60 + 25 * trial
                    labels = np.concatenate([np.arange(K),
                        rng.randint(0, K, N - K)])
                    rng.shuffle(labels)
                    ids = rng.permutation(K)
                    probe.assignment = np.ascontiguousarray(
                        ids[labels], dtype=np.int64)
                    order = rng.permutation(K)
                    probe.cells_per_cluster = {int(ids[k]):
                        int((labels == k).sum()) for k in order}
                    for move in (0, 1):
                        np.random.seed(900 + trial)
                        if move == 0:
                            _, cells, (lt, others) = probe._propose_split()
                            want = (cells, float(lt[0]), list(others))
                        else:
                            _, _, _, cells, sd = probe._propose_merge()
                            want = (cells, float(sd), None)
                        end = np.random.get_state()
                        np.random.seed(900 + trial)
                        got = _lib.move_propose(table, move,
                            np.fromiter(probe.cells_per_cluster.keys(),
                                np.int64),
                            np.fromiter(probe.cells_per_cluster.values(),
                                np.int64), probe.assignment)
                        now = np.random.get_state()
                        ok &= got is not None \
                            and np.array_equal(got[0], want[0]) \
                            and got[3] == want[1] \
                            and (want[2] is None
                                or list(got[4]) == want[2]) \
                            and now[2] == end[2] \
                            and np.array_equal(now[1], end[1])
                _MOVES_OK['ok'] = bool(ok)
            except Exception:       # noqa: BLE001
                _MOVES_OK['ok'] = False
            finally:
                np.random.set_state(saved)
    return _MOVES_OK['ok']


# the methods a native STEP (bnpc_chain_step) stands for, on top of the moves'
_STEP_METHODS = _MOVE_STEPS + ('update_assignments_Gibbs', '_gibbs_window',
    '_sweep_order', '_new_cluster_ll', 'get_lpost_single_new_cluster',
    'update_assignments_split_merge', 'do_split_move', 'do_merge_move',
    '_native_move', 'update_DP_alpha', 'init_DP_prior', 'log_CRP_prior',
    'update_parameters', '_label_counts', '_known_prior', '_remember_prior',
    '_prior_density_sum', 'get_ll_full', 'get_ll_full_deferred', '_ll_total',
    'get_lprior_full', '_memo', 'update_error_rates', 'MH_error_rates',
    'get_ll_full_error', '_error_prior_logpdf')
_STEP_SET = frozenset(_STEP_METHODS)
_plain_step_classes = {}


def _overrides_a_step_method(model):
    """Whether the model's class (or the instance itself) replaces one of the
    methods a native step is made of: only the two classes of this module as
    they are take the one-call step."""
    cls = type(model)
    plain = _plain_step_classes.get(cls)
    if plain is None:
        base = CRP_errors_learning if issubclass(cls, CRP_errors_learning) \
            else CRP
        plain = _plain_step_classes[cls] = all(
            getattr(cls, name, None) is getattr(base, name, None)
            for name in _STEP_METHODS)
    return not plain or not _STEP_SET.isdisjoint(model.__dict__)


_STEP = {}


def _native_step_allowed():
    """Whether whole steps may be made natively in this process
    (bnpc_chain_step): BNPC_NATIVE_STEP != 0, the kernel table with SciPy's
    gammaln, the native Beta sampler, and the scalar pieces the step restates
    on its own - np.random.gamma on the stream, the Gamma log-density of the
    concentration parameter - bit-identical to NumPy's / SciPy's here."""
    pid = os.getpid()
    if _STEP.get('pid') != pid:
        _STEP.clear()
        _STEP.update(pid=pid, ok=False)
        table = _native_kernels()
        if table is not None and table.gammaln and _native_beta() \
                and _lib.rng_live() is not None:
            saved = np.random.get_state()
            try:
                ok = True
                for shape, scale in ((71.7, 3.2), (.4, 1.), (1., 2.5),
                        (5000.2, .11), (2.5, 7.)):
                    for pre in (0, 1):
                        np.random.seed(4242 + pre)
                        if pre:
                            np.random.normal()
                        start = np.random.get_state()
                        want = np.random.gamma(shape, scale)
                        end = np.random.get_state()
                        np.random.set_state(start)
                        got = _lib.gamma(shape, scale)
                        now = np.random.get_state()
                        ok &= want == got and now[2:] == end[2:] \
                            and np.array_equal(now[1], end[1])
                for x, a, loc in ((70.71, 70.71, 1), (1 + 1e-15, 31.6, 1),
                        (3.7, 2.0, 0.5), (224.0, 223.6, 1), (12.5, 100.0, 1)):
                    ok &= _lib.gamma_logpdf_scalar(table, x, a, loc) \
                        == float(fastdist.gamma_logpdf(x, a, loc))
                _STEP['ok'] = bool(ok)
            except Exception:       # noqa: BLE001
                _STEP['ok'] = False
            finally:
                np.random.set_state(saved)
    return _STEP['ok']


def fast_paths():
    """The private interfaces and start-up comparisons the fast paths of this
    process rest on, each True (in use) or False (its documented fallback
    runs: same chain, slower) - what bench.py prints as `host.fast_paths`:

        rng_live      NumPy's MT19937 state addressed in place
                      (bit_generator.ctypes.state_address), else copies
                      through get_state / set_state per native call
        gauss_live    the stream's cached Gaussian located inside the
                      RandomState object, else copies likewise
        scipy_capi    scipy.special.cython_special's C entry points
        ufunc_loops   NumPy's float64 inner loops of log / exp / log1p / expm1
        native_mh     the parameter batch natively, bit-compared with the
                      SciPy-level evaluation (else array expressions)
        native_beta   NumPy's legacy Beta sampler restated natively
        native_moves  a split / merge move as one native call
        native_step   a whole step as one native call
    """
    out = {'rng_live': _lib.rng_live() is not None,
        'gauss_live': _lib.gauss_live() is not None}
    out.update(hostkernels.probe())
    out['native_mh'] = _native_kernels() is not None
    out['native_beta'] = bool(_native_beta())
    out['native_moves'] = bool(_native_moves_allowed()) \
        and _lib.env('BNPC_NATIVE_MOVES', '1') != '0'
    out['native_step'] = out['native_moves'] and out['native_beta'] \
        and bool(_native_step_allowed()) \
        and _lib.env('BNPC_NATIVE_STEP', '1') != '0'
    return out


class CRP:
    """DPMM of Bernoulli profiles with fixed error rates (libs/CRP.py:17)."""

    def __init__(self, data, DP_alpha=-1, param_beta=[1, 1],
                FN_error=EPSILON, FP_error=EPSILON):
        self.data = data
        self.cells_total, self.muts_total = data.shape

        self.p, self.q = param_beta
        self.param_prior = _beta_dist(self.p, self.q)
        self.beta_prior_uniform = bool(self.p == self.q == 1)
        m0 = self.beta_fct(self.p, self.q + 1)
        m1 = self.beta_fct(self.p + 1, self.q)
        self._beta_mix_const = np.array([m0, m1]) / (m0 + m1)

        self.FP = FP_error
        self.FN = FN_error

        try:
            neg = DP_alpha[0] < 0 or DP_alpha[1] < 0
        except TypeError:
            neg = True
        self.DP_a_gamma = (np.sqrt(self.cells_total), 1) if neg else DP_alpha
        self.DP_a_prior = _gamma_dist(*self.DP_a_gamma)
        self.DP_a = self.DP_a_prior.mean()

        self.CRP_prior = None
        self.assignment = None
        self.parameters = None
        self.cells_per_cluster = None
        self.param_proposal_sd = np.array([0.1, 0.25, 0.5])

        self._reset_device_state()

    # ---------------------------------------------------------- life cycle
    def _reset_device_state(self):
        self._ctx = None
        self._newcl = None          # ((FP, FN), per-cell sums)
        self._lab = None            # per-cluster column counts cache
        self._nat = None            # the native chain (bnpc_chain_step)

    def __getstate__(self):
        """Pickle / deepcopy (libs/MCMC.py:115-128): the device context and
        its caches are dropped; the float64 data matrix travels as int8 codes
        0 | 1 | 3 and the (N, M) parameter table as the rows of the populated
        clusters only (rows of unpopulated ids are never read before they are
        overwritten, libs/CRP.py:291-294, 455-457).  The chain hand-off
        through the pool's pipes is 3 GB per direction at config 5 otherwise."""
        state = self.__dict__.copy()
        state['_ctx'] = None
        state['_newcl'] = None
        state['_lab'] = None
        state['_nat'] = None
        state['_prior_rows'] = None
        state['_rg_view'] = None
        data = state.pop('data')
        if hasattr(data, 'planes'):     # bit planes travel as they are
            state['_data_planes'] = (np.asarray(data.planes), data.shape[1])
        else:
            state['_data_codes'] = np.where(np.isnan(data), 3, data) \
                .astype(np.int8)
        theta = state.get('parameters')
        if isinstance(theta, np.ndarray) and theta.ndim == 2 \
                and self.cells_per_cluster is not None \
                and 4 * len(self.cells_per_cluster) < theta.shape[0]:
            live = np.sort(np.fromiter(self.cells_per_cluster.keys(),
                dtype=np.int64))
            del state['parameters']
            state['_theta_rows'] = (theta.shape, theta.dtype.str, live,
                theta[live])
        return state

    def __setstate__(self, state):
        codes = state.pop('_data_codes', None)
        planes = state.pop('_data_planes', None)
        rows = state.pop('_theta_rows', None)
        self.__dict__.update(state)
        if planes is not None:
            from bnpc_amd.bitplanes import BitPlanes
            self.data = BitPlanes(*planes)
        if codes is not None:
            data = codes.astype(np.float64)
            data[codes == 3] = np.nan
            self.data = data
        if rows is not None:
            shape, dtype, live, values = rows
            self.parameters = np.zeros(shape, dtype=np.dtype(dtype))
            self.parameters[live] = values

    def _dev(self):
        """The device context of this chain, created on first use."""
        if self._ctx is None:
            device = int(os.environ.get('BNPC_DEVICE', '0'))
            self._ctx = _lib.Context(data=self.data, device=device)
        return self._ctx

    def _memo(self, key, compute):
        """Scalar prior densities by argument value (a pure function of the
        key; the same few arguments recur step after step)."""
        memo = self.__dict__.setdefault('_scalar_memo', {})
        hit = memo.get(key)
        if hit is None:
            if len(memo) > 64:
                memo.clear()
            hit = memo[key] = compute()
        return hit

    def close(self):
        if getattr(self, '_nat', None) is not None:
            self._nat.close()
        if self._ctx is not None:
            self._ctx.close()
        self._reset_device_state()

    def __str__(self):
        return ('\nDPMM with:\n'
            f'\t{self.cells_total} cells\n\t{self.muts_total} mutations\n'
            f'\tFixed FN rate: {self.FN}\n\tFixed FP rate: {self.FP}\n'
            '\n\tPriors:\n'
            f'\tParams.:\tBeta({self.p},{self.q})\n'
            f'\tCRP a_0:\tGamma({self.DP_a_gamma[0]:.1f},'
            f'{self.DP_a_gamma[1]})\n')

    # -------------------------------------------------------------- statics
    @staticmethod
    def beta_fct(p, q):
        return _gamma_fn(p) * _gamma_fn(q) / _gamma_fn(p + q)

    @staticmethod
    def log_CRP_prior(n_i, n, a, dtype=np.float64):
        return np.log(n_i, dtype=dtype) - np.log(n - 1 + a, dtype=dtype)

    @staticmethod
    def _normalize_log_probs(probs):
        """libs/CRP.py:88-100 (host; the sweep uses the native replica)."""
        top = int(np.nanargmax(probs))
        others = np.arange(probs.size) != top
        shifted = probs[others] - probs[top]
        try:
            tail = np.exp(shifted)
        except FloatingPointError:
            tail = np.exp(np.clip(shifted, log_EPSILON, 0))
        lnorm = probs - probs[top] - np.log1p(np.cumsum(tail)[-1]
            if tail.size else 0.0)
        return np.exp(np.clip(lnorm, log_EPSILON, 0))

    @staticmethod
    def _normalize_log(probs):
        """libs/CRP.py:103-116"""
        top = int(np.nanargmax(probs))
        others = np.arange(probs.size) != top
        try:
            tail = np.exp(probs[others] - probs[top])
            res = probs - probs[top] - np.log1p(np.cumsum(tail)[-1])
        except FloatingPointError:
            if probs[0] > probs[1]:
                return np.array([0, log_EPSILON])
            return np.array([log_EPSILON, 0])
        return res

    # ----------------------------------------------------- element tables
    def _tables(self, theta, FP=None, FN=None):
        """The two values log(theta*P(x|1) + (1-theta)*P(x|0)) can take:
        L1 for an observed 1, L0 for an observed 0 (libs/CRP.py:198-200 with
        :207-212 at x = 1 / x = 0).  (1 - theta) is evaluated in theta's own
        dtype (float32 for cluster parameters), everything else in float64."""
        FP = self.FP if FP is None else FP
        FN = self.FN if FN is None else FN
        theta = np.asarray(theta)
        t64 = theta.astype(np.float64)
        om64 = (1 - theta).astype(np.float64)
        L1 = np.log(t64 * (1 - FN) + om64 * FP)
        L0 = np.log(t64 * FN + om64 * (1 - FP))
        return L1, L0

    def _subset_ll(self, theta, counts, FP=None, FN=None):
        """sum over the cells of a subset of the per-element log-likelihood,
        per mutation: n1*L1 + n0*L0 (the bn.nansum(axis=0) of
        libs/CRP.py:363-368)."""
        L1, L0 = self._tables(theta, FP, FN)
        return counts[0] * L1 + counts[1] * L0

    def _counts_of(self, cells):
        """(n1, n0) float64 M-vectors: observed 1s / 0s per mutation."""
        cells = np.asarray(cells, dtype=np.int64).reshape(-1)
        if cells.size <= 2:
            sub = self.data[cells]
            return (np.nansum(sub, axis=0),
                np.nansum(1 - sub, axis=0))
        n1, n0 = self._dev().colcounts([cells])
        return n1[0].astype(np.float64), n0[0].astype(np.float64)

    # ----------------------------------------------------------------- init
    def init(self, mode='random', assign=False):
        """libs/CRP.py:119-152"""
        self._newcl = None
        self._lab = None
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
        _, inv, counts = np.unique(labels, return_inverse=True,
            return_counts=True)
        self.assignment = np.ascontiguousarray(inv.reshape(-1), dtype=np.int64)
        self.cells_per_cluster = {i: c for i, c in enumerate(counts)}
        self.parameters = self._init_cl_params('assign' if assign else mode)
        self.init_DP_prior()

    def _init_cl_params(self, mode='random', fkt=1):
        """libs/CRP.py:155-180.  Rows of unpopulated ids hold TMIN (a zero
        clipped), as in the reference."""
        N, M = self.data.shape
        params = np.full((N, M), np.float32(TMIN), dtype=np.float32)
        if mode == 'separate':
            data = np.asarray(self.data)        # (bit planes: materialised)
            draw = np.random.beta(
                np.nan_to_num(self.p + data * fkt,
                    nan=self._beta_mix_const[0]),
                np.nan_to_num(self.q + (1 - data) * fkt,
                    nan=self._beta_mix_const[1]))
            params = np.clip(draw, TMIN, TMAX).astype(np.float32)
        elif mode == 'together':
            n1, n0 = self._counts_of(np.arange(N))
            params[0] = self._beta_draw(n1 * fkt, n0 * fkt)
        elif mode == 'assign':
            ids = list(self.cells_per_cluster)
            n1, n0 = self._dev().colcounts_by_label(self.assignment, ids)
            for row, cl in enumerate(ids):
                params[cl] = self._beta_draw(n1[row] * fkt, n0[row] * fkt)
        elif mode == 'random':
            k = np.unique(self.assignment)
            draw = np.random.uniform(size=(k.size, M))
            params[k] = np.clip(draw, TMIN, TMAX).astype(np.float32)
        return params

    def _beta_draw(self, n1, n0):
        """clip(Beta(p + n1, q + n0)) as float32, one draw per mutation in
        order (libs/CRP.py:183-188): NumPy's legacy sampler, natively on the
        same stream when the start-up comparison allows."""
        if _native_beta():
            draw = _lib.beta(self.p + n1, self.q + n0)
        else:
            draw = np.random.beta(self.p + n1, self.q + n0)
        return np.clip(draw, TMIN, TMAX).astype(np.float32)

    def _init_cl_params_new(self, i, fkt=1):
        """Beta draw from the column counts of cells i (libs/CRP.py:183-188)."""
        n1, n0 = self._counts_of(i)
        return self._beta_draw(n1 * fkt, n0 * fkt)

    def init_DP_prior(self):
        """libs/CRP.py:191-194"""
        sizes = np.append(np.arange(1, self.cells_total + 1), self.DP_a)
        self.CRP_prior = np.append(
            0, self.log_CRP_prior(sizes, self.cells_total, self.DP_a))

    # ------------------------------------------------- likelihood surface
    # The reference's private helpers _calc_ll / _Bernoulli_FN / _Bernoulli_FP
    # (libs/CRP.py:197-212) take raw data ROWS; here every caller hands cell
    # INDICES to the device primitives instead, so there is deliberately no
    # host-side likelihood routine in this class.
    def get_lpost_single(self, cell_id, cl_ids):
        """libs/CRP.py:223-227 (one cell; the sweep evaluates all at once)."""
        ctx = self._dev()
        ctx.view_set(VIEW_ONE, [cell_id])
        ll = ctx.ll_theta(VIEW_ONE, self.parameters[cl_ids], self.FP,
            self.FN)[0]
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        return ll + self.CRP_prior[sizes]

    def _new_cluster_ll(self):
        """Per-cell sum over mutations of log(mix1*P(x|1) + mix0*P(x|0)):
        an m-sequential device sum over constant tables, cached per (FP, FN)."""
        key = (self.FP, self.FN)
        if self._newcl is None or self._newcl[0] != key:
            mix0, mix1 = self._beta_mix_const
            c1 = np.log(np.array([mix1 * (1 - self.FN) + mix0 * self.FP]))[0]
            c0 = np.log(np.array([mix1 * self.FN + mix0 * (1 - self.FP)]))[0]
            M = self.muts_total
            ll = self._dev().ll_tables(VIEW_ALL, np.full((1, M), c1),
                np.full((1, M), c0))[:, 0]
            self._newcl = (key, ll)
        return self._newcl[1]

    def get_lpost_single_new_cluster(self):
        """libs/CRP.py:230-234"""
        return self._new_cluster_ll() + self.CRP_prior[-1]

    # per-cluster column counts, valid while the assignment is unchanged
    def _label_counts(self):
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=np.int64)
        lab = self._lab
        if lab is not None and np.array_equal(lab['ids'], ids) \
                and np.array_equal(lab['assignment'], self.assignment):
            return lab
        n1, n0 = self._dev().colcounts_by_label(self.assignment, ids)
        self._lab = {'ids': ids, 'assignment': self.assignment.copy(),
            'n1': n1, 'n0': n0}
        return self._lab

    def _ll_total(self, FP, FN):
        lab = self._label_counts()
        return self._dev().ll_total(self.parameters[lab['ids']], FP, FN)

    def get_ll_full(self):
        """libs/CRP.py:237-238"""
        return float(self._ll_total([self.FP], [self.FN])[0])

    def get_ll_full_deferred(self):
        """get_ll_full in two halves: the launch now, the value when the
        returned function is called - host work (get_lprior_full) runs under
        the launch in between."""
        lab = self._label_counts()
        ctx = self._dev()
        ctx.ll_total_issue(self.parameters[lab['ids']], [self.FP], [self.FN])
        return lambda: float(ctx.ll_total_wait()[0])

    def get_lprior_full(self):
        """libs/CRP.py:241-251"""
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=int)
        # Gamma(*DP_a_gamma): scipy reads the pair as (shape, loc), as in the
        # reference (libs/CRP.py:55)
        shape, loc = self.DP_a_gamma[0], self.DP_a_gamma[1]
        lprior = self._memo(('DP_a', self.DP_a), lambda:
                fastdist.gamma_logpdf(self.DP_a, shape, loc)) \
            + np.cumsum(self.CRP_prior[sizes])[-1]
        if not self.beta_prior_uniform:
            ids = np.fromiter(self.cells_per_cluster.keys(), dtype=np.int64)
            lprior += self._prior_density_sum(ids, self.parameters[ids])
        return lprior

    # ---------------------------------------------------------------- Gibbs
    def update_assignments_Gibbs(self):
        """libs/CRP.py:254-288: the N x K log-likelihood matrix comes from the
        device, the sequential per-cell loop runs natively on the replica of
        NumPy's stream; Python only opens new clusters (rare).

        If the matrix fits the host budget (BNPC_SWEEP_BYTES, default 256 MiB)
        it is computed in ONE launch for all cells (rows = cell ids).  A
        larger sweep - the first one, from K0 ~ 0.63 N clusters - is tiled
        over permutation-ordered chunks of cells: each tile is evaluated only
        for the clusters still alive when it starts (clusters die quickly
        during that sweep), so the work and the memory shrink as the sweep
        proceeds.  Same decisions, same draws, same trajectory either way.
        """
        N = self.cells_total
        # (checker hook: a sweep that ends after its first `_sweep_stop`
        # cells - the state the CPU oracle reaches walking that many cells of
        # config 5's first sweep, tests/golden/make_c5_first_cells.py)
        stop = min(N, getattr(self, '_sweep_stop', None) or N)
        timing = _lib.env('BNPC_TIMING') in ('1', '2')
        if timing:
            import time
            t_start = time.perf_counter()
        post_new = np.ascontiguousarray(self.get_lpost_single_new_cluster())
        crp_prior = np.ascontiguousarray(self.CRP_prior, dtype=np.float64)
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=np.int64)
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=np.int64)
        K_start = ids.size

        budget = int(_lib.env('BNPC_SWEEP_BYTES', 256 << 20))
        ctx = self._dev()
        # a few spare columns for clusters opened during the sweep (the
        # matrix is re-allocated with more if they run out); the budget is
        # tested on the matrix as allocated (bnpc_step.cpp: gibbs_phase has
        # the same expression)
        spare = max(4, min(16, ids.size // 4)) if ids.size <= 64 \
            else min(512, max(16, ids.size // 8))
        if budget // (8 * (ids.size + spare)) >= N:
            # the whole matrix in one launch, rows = cell ids
            # the device also reports, per cell, the four largest entries
            # under the priors at launch with the columns of three: the
            # native loop then decides most cells without scanning them
            hint = None
            if ids.size + spare <= _lib.HINT_COLS_MAX \
                    and _lib.env('BNPC_SWEEP_HINT', '1') != '0':
                # the sums are queued, not waited for: the permutation - the
                # sweep's first draw, which needs nothing from the device - is
                # drawn under them, then the hint kernel is queued with that
                # order (record r = the cell visited r-th: the loop reads its
                # records front to back) and the sweep's private state is
                # copied under it
                col_prior = np.ascontiguousarray(crp_prior[sizes])
                in_order = hasattr(ctx, 'hints_in_order_issue')
                if in_order:
                    ll = ctx.ll_theta_pinned_sums_issue(VIEW_ALL,
                        self.parameters[ids], self.FP, self.FN,
                        ids.size + spare, col_prior)
                    perm, assignment = self._sweep_order(N)
                    top2 = ctx.hints_in_order_issue(perm)
                else:
                    ll, top2 = ctx.ll_theta_pinned_top2(VIEW_ALL,
                        self.parameters[ids], self.FP, self.FN,
                        ids.size + spare, col_prior, wait=False)
                    perm, assignment = self._sweep_order(N)
                ctx.hints_wait()
                if top2 is not None:
                    hint = (top2, col_prior, in_order)
            else:
                perm, assignment = self._sweep_order(N)
                ll = ctx.ll_theta_pinned(VIEW_ALL, self.parameters[ids],
                    self.FP, self.FN, ids.size + spare)
            ids, sizes, born = self._gibbs_window(perm, 0, stop, VIEW_ALL, ll,
                ids, ids, sizes, (), assignment, post_new, crp_prior,
                hint=hint)
            opened, tiles = len(born), 1
        else:
            perm, assignment = self._sweep_order(N)
            if timing:
                print(f'[bnpc]   new-cluster term + order: '
                    f'{time.perf_counter() - t_start:.3f}s', flush=True)
            # Tiled sweep.  The parameter rows stay resident on the device
            # (row = cluster id) and tiles select clusters by index.  Tiles
            # t+1 and t+2 are in flight while the host walks tile t - the
            # device sums t+2 while the copy engine moves t+1 - each issued
            # for the clusters alive at that moment: a superset of what it
            # will need, minus the clusters (re)born before it is picked up,
            # whose columns are evaluated then.
            ctx.theta_put(0, self.parameters[:int(ids.max()) + 1])
            if timing:
                print(f'[bnpc]   parameter rows on the device: '
                    f'{time.perf_counter() - t_start:.3f}s', flush=True)

            tile_bytes = min(budget,
                int(_lib.env('BNPC_TILE_BYTES', 256 << 20)))
            # (two tiles in flight while the host walks one: the device sums
            # t + 2 while the copy engine moves t + 1; one or three measured
            # slower in round 4 - a constant since round 6)
            ahead_max = max(1, min(_lib.TILE_SLOTS - 1, _TILES_AHEAD))
            born_log = []       # ids born during this sweep, in order
            tile_hints = _lib.env('BNPC_SWEEP_HINT', '1') != '0' \
                and hasattr(ctx, 'll_rows_issue_hint')

            def issue(start, number):
                rows = max(64, tile_bytes // (8 * (ids.size + _TILE_SPARE)))
                # whole workgroups of the likelihood kernel: 8 blocks of 64
                # cells (measured at 31608 x 5000: 264 rows 6.0e12
                # element-evals/s, 512 rows 10.6e12, 1024 rows 10.9e12)
                if rows >= 512:
                    rows -= rows % 512
                slot = number % _lib.TILE_SLOTS
                tile = dict(pos=start, end=min(stop, start + rows), slot=slot,
                    number=number, view=VIEW_SWEEP + slot, cols=ids.copy(),
                    ld=ids.size + _TILE_SPARE, born_mark=len(born_log))
                if timing:
                    t_v = time.perf_counter()
                ctx.view_set_slot(tile['view'], perm[start:tile['end']], slot)
                if timing:
                    spent['view'] = spent.get('view', 0.0) \
                        + time.perf_counter() - t_v
                if tile_hints:
                    # with the priors of the clusters as they are now: the
                    # device also says, per cell, which column is largest and
                    # by how much - once the true clusters exist the loop
                    # decides nearly every cell from that one record
                    tile['prior'] = np.ascontiguousarray(crp_prior[sizes])
                    ctx.ll_rows_issue_hint(tile['view'], tile['cols'],
                        self.FP, self.FN, tile['ld'], slot, tile['prior'])
                else:
                    ctx.ll_rows_issue(tile['view'], tile['cols'], self.FP,
                        self.FN, tile['ld'], slot)
                return tile

            opened, tiles = 0, 0
            in_flight = []
            upcoming = [0, 0]           # start position, tile number

            spent = {'issue': 0.0, 'wait': 0.0}

            def fill():
                while len(in_flight) < ahead_max and upcoming[0] < stop:
                    if timing:
                        t_0 = time.perf_counter()
                    tile = issue(upcoming[0], upcoming[1])
                    if timing and upcoming[1] < 3:
                        print(f'[bnpc]   tile {upcoming[1]} issued: '
                            f'{time.perf_counter() - t_start:.3f}s',
                            flush=True)
                    in_flight.append(tile)
                    upcoming[0] = tile['end']
                    upcoming[1] += 1
                    if timing:
                        spent['issue'] += time.perf_counter() - t_0

            try:
                fill()
                while in_flight:
                    tile = in_flight.pop(0)
                    if timing:
                        t_0 = time.perf_counter()
                    tile_hint = None
                    if 'prior' in tile:
                        ll, records = ctx.ll_rows_wait_hint(tile['slot'],
                            tile['end'] - tile['pos'], tile['ld'])
                        if records is not None:
                            tile_hint = (records, tile['prior'])
                    else:
                        ll = ctx.ll_rows_wait(tile['slot'],
                            tile['end'] - tile['pos'], tile['ld'])
                    if timing:
                        spent['wait'] += time.perf_counter() - t_0
                    # the slot of `tile` stays ours until it has been walked:
                    # ahead_max tiles in flight + this one <= TILE_SLOTS
                    fill()
                    ids, sizes, born = self._gibbs_window(perm, tile['pos'],
                        tile['end'], tile['view'], ll, tile['cols'], ids,
                        sizes, born_log[tile['born_mark']:], assignment,
                        post_new, crp_prior, hint=tile_hint)
                    born_log.extend(born)
                    opened += len(born)
                    tiles += 1
            finally:
                # an exception must not leave an issued tile behind (the
                # context refuses to re-use its slot)
                for left in in_flight:
                    try:
                        (ctx.ll_rows_wait_hint if 'prior' in left
                            else ctx.ll_rows_wait)(left['slot'],
                            left['end'] - left['pos'], left['ld'])
                    except RuntimeError:
                        pass
        if timing:
            extra = ''
            if tiles > 1:
                extra = (f"; issuing {spent['issue']:.3f}s (views "
                    f"{spent.get('view', 0.0):.3f}s), waiting for the "
                    f"device {spent['wait']:.3f}s")
            print(f'[bnpc] gibbs N={N} K={K_start}->{ids.size}: '
                f'{time.perf_counter() - t_start:.3f}s in {tiles} tile(s), '
                f'{opened} clusters opened{extra}', flush=True)

        self.assignment = assignment
        self.cells_per_cluster = {
            int(i): int(n) for i, n in zip(ids, sizes)}

    def _sweep_order(self, N):
        """(visiting order, private copy of the assignment): the sweep's
        result is committed at the end, so an exception half-way leaves
        assignment and cells_per_cluster consistent with each other
        (parameter rows of ids re-used by the aborted sweep are not restored
        - the run is over at that point anyway)."""
        if _lib.rng_live() is not None:
            perm = _lib.permutation(N)
        else:
            perm = _lib.as_i64(np.random.permutation(N))
        return perm, np.array(self.assignment, dtype=np.int64, order='C')

    def _gibbs_window(self, perm, pos, pos_end, view, ll, cols, ids, sizes,
                stale, assignment, post_new, crp_prior, hint=None):
        """Positions [pos, pos_end) of the sweep.

        ll: (rows, ld) matrix whose first cols.size columns were evaluated for
        the cluster ids `cols` (rows = cell ids for VIEW_ALL, tile positions
        otherwise); `ids`/`sizes`: the clusters alive now, dict order - a
        subset of `cols` plus the ids in `stale` (clusters (re)born after the
        matrix was issued: their columns are evaluated here).  Returns the
        live clusters (dict order) after the window and the ids born in it."""
        lib = _lib.load()
        ctx = self._dev()
        N = self.cells_total
        whole = view == VIEW_ALL
        n_rows, ld = ll.shape
        tile_timing = _lib.env('BNPC_TIMING') == '2'
        if tile_timing:
            import time
            t0 = time.perf_counter()

        # column of every live cluster: its slot in the issued matrix, or a
        # fresh one behind them
        col_in_ll = np.full(N, -1, dtype=np.int64)
        col_in_ll[cols] = np.arange(cols.size)
        if len(stale):
            col_in_ll[np.asarray(stale, dtype=np.int64)] = -1
        live_col = col_in_ll[ids]
        late = np.flatnonzero(live_col < 0)
        n_cols = cols.size + late.size
        if n_cols + 1 > ld:
            if hint is not None and whole:
                ctx.matrix_wait()
            ll = np.concatenate([ll, np.empty((n_rows, n_cols + 16 - ld))],
                axis=1)
            ld = ll.shape[1]
        if late.size:
            live_col[late] = cols.size + np.arange(late.size)
            ll[:, cols.size:n_cols] = ctx.ll_theta(view,
                self.parameters[ids[late]], self.FP, self.FN)

        K = ids.size
        col_of_id = np.full(N, -1, dtype=np.int64)
        col_of_id[ids] = live_col
        col_id = np.full(ld, -1, dtype=np.int64)
        col_id[live_col] = ids
        col_size = np.zeros(ld, dtype=np.int64)
        col_size[live_col] = sizes
        order = np.zeros(ld, dtype=np.int64)
        order[:K] = live_col
        scratch = np.empty(2 * (ld + 1), dtype=np.float64)

        st = _lib.GibbsState(N, ld, n_cols, K, pos, -1, pos_end,
            -1 if whole else pos, _lib.host_threads())
        if hint is not None and (not whole or not late.size):
            # columns 0..cols.size-1 of ll are the hint's columns (a tile's
            # hint also covers columns of clusters that have died or whose id
            # was re-used since: the loop never picks a dead column, and the
            # hint's largest entry bounds every issued column from above)
            st.hint = _lib.ptr(hint[0])
            st.hint_prior = _lib.ptr(hint[1])
            st.hint_cols = cols.size
            # (the records of a whole-matrix sweep may be in visiting order)
            st.hint_in_order = int(whole and len(hint) > 2 and bool(hint[2]))
            hook = getattr(ctx, 'matrix_wait_hook', None) if whole else None
            if hook is not None:
                # the matrix is copied behind the loop, which waits for it
                # itself before the first row it has to read
                st.matrix_wait, st.matrix_wait_arg = hook()
        i64, f64 = C.c_int64, C.c_double
        born = []
        # clusters are opened INSIDE the native loop (lowest free id, Beta
        # draws of the profile on the same stream, the new column from the
        # device) when the context is a real one and the native sampler has
        # passed its comparison; the Python branch below remains for the
        # births the call cannot make itself (no spare column left)
        theta = self.parameters
        native_births = bool(getattr(ctx, '_h', None)) and _native_beta() \
            and _lib.env('BNPC_NATIVE_BIRTHS', '1') != '0' \
            and theta.dtype == np.float32 and theta.flags['C_CONTIGUOUS'] \
            and theta.shape[1] == self.muts_total
        if native_births:
            born_buf = np.empty(256, dtype=np.int64)
            st.birth_ctx = ctx._h
            st.birth_view = view
            st.birth_put = 0 if whole else 1
            st.birth_rows = n_rows
            st.theta_host = _lib.ptr(theta)
            st.beta_p, st.beta_q = float(self.p), float(self.q)
            st.tmin, st.tmax = TMIN, TMAX
            st.FP, st.FN = float(self.FP), float(self.FN)
            st.born, st.born_cap = _lib.ptr(born_buf), born_buf.size
        while True:
            st.n_born = 0
            with _lib.NumpyGaussStream() as (rng, gauss):
                st.gauss = gauss if native_births else None
                _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), rng,
                    _lib.ptr(perm, i64), _lib.ptr(ll, f64),
                    _lib.ptr(post_new, f64), _lib.ptr(crp_prior, f64),
                    _lib.ptr(assignment, i64), _lib.ptr(col_of_id, i64),
                    _lib.ptr(col_id, i64), _lib.ptr(col_size, i64),
                    _lib.ptr(order, i64), _lib.ptr(scratch, f64)),
                    'gibbs_sweep')
            if native_births and st.n_born:
                born.extend(int(i) for i in born_buf[:st.n_born])
            if st.new_cell < 0:
                break
            if hint is not None and whole:
                # the matrix may still be on its way: it is about to be
                # copied / written to from here
                ctx.matrix_wait()
            # open a new cluster for this cell (libs/CRP.py:281-282, 291-299);
            # its Beta draws continue the same stream
            cell = int(st.new_cell)
            new_id = int(np.flatnonzero(col_of_id < 0)[0])
            self.parameters[new_id] = self._init_cl_params_new([cell])
            if not whole:
                ctx.theta_put(new_id, self.parameters[new_id])
            if st.n_cols == ld:
                grow = max(16, ld // 4)
                ll = np.concatenate(
                    [ll, np.empty((n_rows, grow), dtype=np.float64)], axis=1)
                col_id = np.concatenate([col_id, np.full(grow, -1, np.int64)])
                col_size = np.concatenate([col_size, np.zeros(grow, np.int64)])
                order = np.concatenate([order, np.zeros(grow, np.int64)])
                ld += grow
                scratch = np.empty(2 * (ld + 1), dtype=np.float64)
                st.ld = ld
            col = int(st.n_cols)
            ll[:, col] = ctx.ll_theta(view, self.parameters[[new_id]],
                self.FP, self.FN)[:, 0]
            col_id[col] = new_id
            col_size[col] = 1
            col_of_id[new_id] = col
            order[st.n_active] = col
            st.n_active += 1
            st.n_cols += 1
            assignment[cell] = new_id
            born.append(new_id)
        live = order[:st.n_active]
        self._hint_used = getattr(self, '_hint_used', 0) + int(st.hint_used)
        self._pair_used = getattr(self, '_pair_used', 0) + int(st.pair_used)
        self._triple_used = getattr(self, '_triple_used', 0) \
            + int(st.triple_used)
        self._lane_used = getattr(self, '_lane_used', 0) + int(st.lane_used)
        self._stride_used = getattr(self, '_stride_used', 0) \
            + int(st.stride_used)
        self._swept = getattr(self, '_swept', 0) + (pos_end - pos)
        if tile_timing:
            print(f'[bnpc]   tile [{pos},{pos_end}) cols={cols.size} '
                f'late={late.size} born={len(born)}: host '
                f'{time.perf_counter() - t0:.3f}s', flush=True)
        return col_id[live].copy(), col_size[live].copy(), born

    def init_new_cluster(self, cell_id):
        """libs/CRP.py:291-294"""
        cl = self.get_empty_cluster()
        self.parameters[cl] = self._init_cl_params_new([cell_id])
        return cl

    def get_empty_cluster(self):
        """libs/CRP.py:297-299"""
        i = 0
        while i in self.cells_per_cluster:
            i += 1
        return i

    # ------------------------------------------------------- a whole step
    def native_step(self, knobs, fix_assign, learning, record=None):
        """One step of the sampler - the move schedule of libs/MCMC.py:320-342
        and the recording of :242-258 - as ONE native call (bnpc_chain_step),
        when nothing stands in the way: a real device context, the kernel
        table, the native samplers, no overridden method.  Returns None when
        the step is not made here (nothing was drawn or changed: the caller
        walks the schedule itself), else a dict:

            move        None | 'split' | 'merge' | 'gibbs'
            sm          [accepted, declined] of a split / merge move
            parameters  (declined, accepted)
            errors      None | (FP [acc, dec], FN [acc, dec])
            recorded    whether `record` was served (ML, MAP, ... written)

        record = (addresses of the ML / MAP / DP_alpha / FN / FP slots,
        address of the assignment row, address of the parameter block or 0,
        its capacity in clusters), or None: no recording (do_step alone).
        Phases the library hands back (first steps with thousands of clusters,
        an element left to SciPy) run through the methods of this class in
        between, on the same stream."""
        table = _native_kernels()
        if table is None or _lib.env('BNPC_NATIVE_STEP', '1') == '0' \
                or _lib.env('BNPC_TIMING') in ('1', '2') \
                or not _native_step_allowed() \
                or not _native_beta() or not _native_moves_allowed() \
                or _lib.env('BNPC_NATIVE_MOVES', '1') == '0' \
                or _overrides_a_step_method(self):
            return None
        ratios = knobs['sm_ratios']
        if len(ratios) != 2 or not abs(float(ratios[0]) + float(ratios[1])
                - 1.) <= np.sqrt(np.finfo(np.float64).eps):
            # np.random.choice(p=) raises for these (libs/CRP.py:427): the
            # step-by-step path lets NumPy do so
            return None
        ctx = self._dev()
        theta = self.parameters
        if not getattr(ctx, '_h', None) or theta is None \
                or theta.dtype != np.float32 or not theta.flags['C_CONTIGUOUS'] \
                or theta.shape != (self.cells_total, self.muts_total) \
                or self.CRP_prior is None \
                or self.CRP_prior.dtype != np.float64 \
                or self.CRP_prior.size != self.cells_total + 2:
            return None
        nat = self._nat
        if nat is None:
            nat = self._nat = self._native_chain(learning)
        st = nat.st
        lib = nat._lib
        # ---- the state as it is now ----
        assignment = self.assignment
        if assignment.dtype != np.int64 \
                or not assignment.flags['C_CONTIGUOUS']:
            assignment = self.assignment = np.ascontiguousarray(
                assignment, dtype=np.int64)
        self._state_to_chain(nat)
        st.sm_prob = knobs['sm_prob']
        st.dpa_prob = knobs['dpa_prob']
        st.error_prob = knobs['error_prob'] if learning else 0.0
        st.sm_ratios[0], st.sm_ratios[1] = knobs['sm_ratios']
        st.sm_steps = knobs['sm_steps']
        st.fix_assign = 1 if fix_assign else 0
        st.learning = 1 if learning else 0
        # (the team sizes follow two environment variables: looked up again
        # only when those have changed - 4 us a step otherwise)
        team_key = (_lib.env('BNPC_HOST_THREADS'), _lib.env('BNPC_HOST_SHARE'))
        if getattr(nat, 'team_key', None) != team_key:
            nat.team_key = team_key
            nat.teams = (_lib.host_threads(), _lib.threads_for(1 << 40))
        st.threads, st.threads_wide = nat.teams
        if record is None:
            for i in range(5):
                st.rec_scalars[i] = None
            st.rec_assignment = st.rec_params = None
            st.rec_params_cap = 0
        else:
            scalars, labels, params, cap = record
            for i in range(5):
                st.rec_scalars[i] = scalars[i]
            st.rec_assignment = labels
            st.rec_params = params or None
            st.rec_params_cap = cap
        st.phase = _lib.PHASE_ASSIGN
        out = {'move': None, 'sm': None, 'parameters': None, 'errors': None,
            'recorded': False}
        addr = C.addressof(table)
        caller_records = False
        while True:
            with _lib.NumpyGaussStream() as (rng, gauss):
                st.gauss = gauss
                _lib.check(lib.bnpc_chain_step(ctx._h, addr, rng,
                    C.byref(st)), 'chain_step')
            need = st.need
            if st.phase == _lib.PHASE_ASSIGN and st.move >= 0:
                out['move'] = ('split', 'merge', 'gibbs')[st.move]
                if st.move < 2 and need != _lib.NEED_MOVE:
                    out['sm'] = [1, 0] if st.sm_accepted else [0, 1]
                    self._rg_view = None    # the move's view was overwritten
            if need == _lib.NEED_NONE:
                break
            # a phase for the methods of this class: the model as the
            # library left it, the phase, the model back
            self._chain_to_state(nat)
            if need == _lib.NEED_MOVE:
                out['sm'] = self.do_split_move(st.sm_steps) if st.move == 0 \
                    else self.do_merge_move(st.sm_steps)
                st.phase = _lib.PHASE_ALPHA
            elif need == _lib.NEED_GIBBS:
                self.update_assignments_Gibbs()
                st.phase = _lib.PHASE_ALPHA
            elif need == _lib.NEED_PARAMS:
                out['parameters'] = self.update_parameters()
                st.phase = _lib.PHASE_ERRORS
            elif need == _lib.NEED_ERRORS:
                out['errors'] = self.update_error_rates()
                st.phase = _lib.PHASE_RECORD
            else:
                # NEED_RECORD (a scalar density left to SciPy - e.g. alpha at
                # or below the Gamma prior's location, where the reference
                # records -inf): the step itself is complete, the caller
                # records through put_state
                caller_records = True
                break
            self._state_to_chain(nat)
        self._chain_to_state(nat)
        if out['parameters'] is None:
            out['parameters'] = (st.par_declined, st.par_accepted)
        if st.errors_updated and out['errors'] is None:
            out['errors'] = (
                [1, 0] if st.FP_accepted else [0, 1],
                [1, 0] if st.FN_accepted else [0, 1])
        if st.move in (0, 1) and st.sm_cells:
            self._note_move('merge' if st.move else 'split', st.sm_cells,
                st.sm_accepted)
        out['recorded'] = record is not None and not caller_records
        out['params_recorded'] = out['recorded'] and bool(st.rec_params_done)
        if not caller_records and record is not None:
            out['ML'], out['lprior'] = st.ML, st.lprior
        self._native_steps = getattr(self, '_native_steps', 0) + 1
        return out

    def _native_chain(self, learning):
        """The bnpc_chain of this model: everything that does not change from
        step to step."""
        nat = _lib.NativeChain(self.cells_total, self.muts_total)
        st = nat.st
        st.param_stride = self.muts_total
        st.dpa_shape = float(self.DP_a_gamma[0])
        st.dpa_rate = float(self.DP_a_gamma[1])
        st.p, st.q = float(self.p), float(self.q)
        st.tmin, st.tmax = TMIN, TMAX
        st.mix0 = float(self._beta_mix_const[0])
        st.mix1 = float(self._beta_mix_const[1])
        st.uniform_prior = int(self.beta_prior_uniform)
        sd = np.ascontiguousarray(self.param_proposal_sd, dtype=np.float64)
        nat.keep.append(sd)
        st.sd, st.n_sd = _lib.ptr(sd), sd.size
        if learning:
            for i in range(4):
                st.FP_prior[i] = float(self.FP_prior.args[i])
                st.FN_prior[i] = float(self.FN_prior.args[i])
            for i in range(3):
                st.FP_sd[i] = float(self.FP_sd[i])
                st.FN_sd[i] = float(self.FN_sd[i])
        st.wide_from = _lib.MH_WIDE_FROM \
            if _lib.env('BNPC_HOST_THREADS') is None else 1 << 62
        st.sweep_bytes = int(_lib.env('BNPC_SWEEP_BYTES', 256 << 20))
        st.view_move = VIEW_MOVE
        st.sweep_hint = int(_lib.env('BNPC_SWEEP_HINT', '1') != '0')
        return nat

    def _state_to_chain(self, nat):
        """Model -> bnpc_chain: what a step reads and the binding may have
        changed since the last one."""
        st = nat.st
        clusters = self.cells_per_cluster
        K = len(clusters)
        nat.ids[:K] = np.fromiter(clusters.keys(), np.int64, K)
        nat.sizes[:K] = np.fromiter(clusters.values(), np.int64, K)
        st.K = K
        st.assignment = _lib.ptr(self.assignment)
        st.parameters = _lib.ptr(self.parameters)
        st.crp_prior = _lib.ptr(self.CRP_prior)
        st.DP_a = self.DP_a
        st.FP, st.FN = self.FP, self.FN

    def _chain_to_state(self, nat):
        """bnpc_chain -> model (the arrays were updated in place)."""
        st = nat.st
        # the library counts per cluster on its own (bnpc_chain_step): the
        # device's resident counts are no longer the ones this cache is for
        self._lab = None
        K = st.K
        self.cells_per_cluster = dict(zip(nat.ids[:K].tolist(),
            nat.sizes[:K].tolist()))
        if st.alpha_updated:
            self.DP_a = st.DP_a
        if st.errors_updated:
            self.FP, self.FN = st.FP, st.FN

    def host_stats(self):
        """Counters of the sweeps and moves made so far (bench.py): cells
        swept / decided from the device's hint / between two / among three
        columns / in the loop's lane (bnpc_gibbs_state.lane_used), moves made
        natively, steps made natively."""
        out = {'swept': getattr(self, '_swept', 0),
            'hint_used': getattr(self, '_hint_used', 0),
            'pair_used': getattr(self, '_pair_used', 0),
            'triple_used': getattr(self, '_triple_used', 0),
            'lane_used': getattr(self, '_lane_used', 0),
            'stride_used': getattr(self, '_stride_used', 0),
            'native_moves': getattr(self, '_native_moves', 0),
            'native_steps': getattr(self, '_native_steps', 0)}
        nat = getattr(self, '_nat', None)
        if nat is not None:
            for key in ('swept', 'hint_used', 'pair_used', 'triple_used',
                    'lane_used', 'stride_used', 'native_moves'):
                out[key] += getattr(nat.st, key)
        # parameter batches whose draws a walker took ahead of them
        ctx = getattr(self, '_ctx', None)
        ahead = getattr(ctx, 'mh_ahead_stats', None)
        out['ahead_begun'], out['ahead_taken'], out['ahead_rows'] = \
            ahead() if ahead is not None else (0, 0, 0)
        return out

    # ------------------------------------------------- cluster parameters
    def update_parameters(self, step_no=None):
        """libs/CRP.py:302-311; the per-cluster sums over cells come from one
        column-count launch that is reused by get_ll_full / the error update,
        and the K proposals are evaluated as one batch (the draws stay in the
        reference's per-cluster order)."""
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=np.int64)
        lab = self._lab
        stale = lab is None or not np.array_equal(lab['ids'], ids) \
            or not np.array_equal(lab['assignment'], self.assignment)
        ctx = self._dev()
        table = _native_kernels()
        if stale and table is not None and getattr(ctx, '_h', None):
            # counts and batch in ONE call: the device counts, screens the
            # proposals against those counts, and hands both back together
            old = self.parameters[ids]
            n1 = np.empty(old.shape, dtype=np.int32)
            n0 = np.empty(old.shape, dtype=np.int32)
            status, new, _, declined, prior, draws = _lib.mh_batch(
                table, old, n1, n0, self.param_proposal_sd, TMIN, TMAX,
                self.FP, self.FN, self.p, self.q, self.beta_prior_uniform,
                False, known=self._known_prior(ids), want_prior=True,
                ctx=ctx, label=(self.assignment, ids))
            self._lab = {'ids': ids, 'assignment': self.assignment.copy(),
                'n1': n1, 'n0': n0}
            if status == 0:
                if prior is not None:
                    self._remember_prior(ids, new, prior)
            else:       # an element the library leaves to SciPy
                new, _, declined = self._mh_batch(old, (n1, n0), False,
                    known=self._known_prior(ids), keep_prior=ids,
                    draws=draws)
            self.parameters[ids] = new
            return declined.sum(), (self.muts_total - declined).sum()
        lab = self._label_counts()
        ids = lab['ids']
        old = self.parameters[ids]
        # (the device holds these very counts: it screens the batch and the
        # host evaluates only the proposals it cannot rule out)
        new, _, declined = self._mh_batch(old, (lab['n1'], lab['n0']), False,
            known=self._known_prior(ids), keep_prior=ids,
            device=(self._dev(), 0))
        self.parameters[ids] = new
        return declined.sum(), (self.muts_total - declined).sum()

    # The Beta prior log-density of a profile entry that has not changed
    # since the last update is the value computed then (an elementwise
    # function of theta): every entry of the result is either the proposal's
    # density (accepted) or the old one (declined).  The cache keeps a copy
    # of theta beside the densities and an entry is only used where the bits
    # of theta match, so writes to `parameters` from anywhere (new clusters,
    # split/merge, callers) simply miss.
    _PRIOR_CACHE_ELEMS = 1 << 22

    def _known_prior(self, ids):
        """(theta copy, density) rows aligned with `ids`, or None (uniform
        prior / nothing cached).  Rows of ids that are not in the cache hold
        NaN parameters, which match nothing."""
        cache = getattr(self, '_prior_rows', None)
        if self.beta_prior_uniform or not cache:
            return None
        c_ids, c_theta, c_prior = cache
        if c_ids.size == ids.size and c_ids.dtype == ids.dtype \
                and c_ids.tobytes() == ids.tobytes():
            return c_theta, c_prior
        pos = {int(cl): g for g, cl in enumerate(c_ids)}
        take = np.array([pos.get(int(cl), -1) for cl in ids], dtype=np.int64)
        theta = c_theta[take]
        theta[take < 0] = np.nan
        return theta, c_prior[take]

    def _remember_prior(self, ids, theta, prior):
        if theta.size > self._PRIOR_CACHE_ELEMS:
            self._prior_rows = None
            return
        self._prior_rows = (np.array(ids, dtype=np.int64), theta.copy(), prior)

    def _prior_density_sum(self, ids, theta):
        """Sum in index order of the Beta log-density of theta =
        parameters[ids] (np.cumsum(...)[-1] of libs/CRP.py:249)."""
        known = self._known_prior(ids)
        table = _native_kernels()
        if table is not None:
            return _lib.beta_logpdf_f32(table, theta, self.p, self.q,
                known=known)[1]
        if known is not None:
            density = np.where(known[0].view(np.int32) == theta.view(np.int32),
                known[1], np.nan)
            miss = np.isnan(density)
            if miss.any():
                density[miss] = fastdist.beta_logpdf(theta[miss], self.p,
                    self.q)
        else:
            density = _rowwise(
                lambda t: fastdist.beta_logpdf(t, self.p, self.q), theta)
        return np.cumsum(density.ravel())[-1]

    def _mh_batch(self, old, counts, trans_prob, known=None, keep_prior=None,
                draws=None, device=None):
        """MH_cluster_params (libs/CRP.py:314-344) for G clusters at once.

        old: (G, M) float32; counts: (n1, n0) each (G, M).  RNG order per
        cluster, as in the reference: choice(sd, M) -> truncnorm.rvs (its M
        uniforms) -> random(M); the arithmetic in between does not draw, so
        it is hoisted out of the loop and batched - natively on the host
        thread team (bnpc_mh_batch) when the self-check allows, else as NumPy
        / SciPy array expressions.  `known`: (theta, density) rows of the
        prior cache; `keep_prior`: cluster ids under which the log-density of
        the result is remembered."""
        G, M = old.shape
        n1, n0 = counts
        table = _native_kernels() if draws is None else None
        if table is not None:
            status, new, prob, declined, prior, draws = _lib.mh_batch(
                table, old, n1, n0, self.param_proposal_sd, TMIN, TMAX,
                self.FP, self.FN, self.p, self.q, self.beta_prior_uniform,
                trans_prob, known=known, want_prior=keep_prior is not None,
                ctx=device[0] if device else None,
                counts_src=device[1] if device else 0)
            if status == 0:
                if keep_prior is not None and prior is not None:
                    self._remember_prior(keep_prior, new, prior)
                return new, prob, declined
            # an element needs a branch the native batch leaves to SciPy:
            # evaluate the whole batch here from the draws already taken
        sd_idx, U, lu = draws if draws is not None \
            else _lib.mh_draws(G, M, self.param_proposal_sd.size)
        std = self.param_proposal_sd[sd_idx]
        old_prior = None
        if known is not None:
            old_prior = np.where(
                known[0].view(np.int32) == old.view(np.int32), known[1],
                np.nan)
            miss = np.isnan(old_prior)
            if miss.any():
                old_prior[miss] = fastdist.beta_logpdf(old[miss], self.p,
                    self.q)
        pool, n_parts = None, _host_parts(G * M, G)
        if n_parts > 1:
            pool = _host_pool()
            n_parts = min(n_parts, pool._max_workers) if pool else 1
        if pool is None or n_parts < 2:
            new, A, decline, prior = self._mh_math(old, std, U, lu, n1, n0,
                trans_prob, old_prior)
        else:
            # large batches (first steps, config 4/5): the per-element SciPy
            # math of row blocks runs on host threads (ufuncs release the
            # GIL); elementwise, so the values do not depend on the split
            edges = np.linspace(0, G, n_parts + 1, dtype=int)

            def block(lo_hi):
                lo, hi = lo_hi
                return self._mh_math(*(x[lo:hi] for x in
                    (old, std, U, lu, n1, n0)), trans_prob,
                    None if old_prior is None else old_prior[lo:hi])
            parts = list(pool.map(block, zip(edges[:-1], edges[1:])))
            new, A, decline = (np.concatenate([p[i] for p in parts])
                for i in range(3))
            prior = None if parts[0][3] is None \
                else np.concatenate([p[3] for p in parts])
        if keep_prior is not None and prior is not None:
            self._remember_prior(keep_prior, new, prior)
        if trans_prob:
            prob = np.cumsum(A, axis=1)[:, -1]
        else:
            prob = np.full(G, np.nan)
        return new, prob, decline.sum(axis=1)

    def _mh_math(self, old, std, U, lu, n1, n0, trans_prob, old_prior=None):
        """Proposal, log acceptance ratio and accept/decline of a block of
        clusters (no random draws in here).  Also returns the Beta prior
        log-density of the resulting profiles (None under a uniform prior)."""
        with np.errstate(divide='raise', over='ignore', under='ignore',
                invalid='raise'):       # error state is per thread
            a = (TMIN - old) / std
            b = (TMAX - old) / std
            draw, fwd_logpdf = fastdist.tn_propose(U, a, b, old, std)
            new = draw.astype(np.float32)
            priors = None
            if not self.beta_prior_uniform:
                if old_prior is None:
                    old_prior = fastdist.beta_logpdf(old, self.p, self.q)
                priors = (fastdist.beta_logpdf(new, self.p, self.q), old_prior)
            A = self._get_log_A(new, old, None, a, b, std, trans_prob,
                counts=(n1, n0), fwd=fwd_logpdf(new), priors=priors)
            decline = np.log(lu) >= A
            new[decline] = old[decline]
            if priors is not None:
                priors = np.where(decline, priors[1], priors[0])
            if trans_prob:
                A[decline] = np.log(-1 * np.expm1(A[decline]))
        return new, A, decline, priors

    def MH_cluster_params(self, old_params, cells, trans_prob=False,
                counts=None):
        """libs/CRP.py:314-344 (draw order: choice(sd) -> truncnorm.rvs ->
        random(M))."""
        if counts is None:
            counts = self._counts_of(cells)
        new, prob, declined = self._mh_batch(
            np.asarray(old_params)[None, :],
            (counts[0][None, :], counts[1][None, :]), trans_prob)
        return new[0], prob[0], declined[0]

    def _get_log_A(self, new_params, old_params, cells, a, b, std, clip=False,
                counts=None, fwd=None, priors=None):
        """libs/CRP.py:347-383 (any leading batch dimension); `fwd` may carry
        the forward proposal log-density and `priors` the Beta log-densities
        of (new, old) if the caller already has them."""
        if counts is None:
            counts = self._counts_of(cells)
        if fwd is None:
            fwd = fastdist.tn_logpdf(new_params, a, b, old_params, std)
        a_rev = (TMIN - new_params) / std
        b_rev = (TMAX - new_params) / std
        rev = fastdist.tn_logpdf(old_params, a_rev, b_rev, new_params, std)

        new_ll = self._subset_ll(new_params, counts)
        old_ll = self._subset_ll(old_params, counts)

        if self.beta_prior_uniform:
            new_prior = old_prior = 0
        elif priors is not None:
            new_prior, old_prior = priors
        else:
            new_prior = fastdist.beta_logpdf(new_params, self.p, self.q)
            old_prior = fastdist.beta_logpdf(old_params, self.p, self.q)

        A = new_ll + new_prior - old_ll - old_prior + rev - fwd
        if clip:
            return np.clip(A, a_min=None, a_max=0)
        return A

    def _log_A_sum(self, new_params, old_params, std, counts, fmin, fmax):
        """np.cumsum(_get_log_A(new, old, None, (fmin - old) / std,
        (fmax - old) / std, std, True, counts))[-1] per row: natively when
        the start-up comparison allows, else through the array path."""
        table = _native_kernels()
        if table is not None:
            res = _lib.log_A(table, new_params, old_params, std, counts[0],
                counts[1], fmin, fmax, TMIN, TMAX, self.FP, self.FN, self.p,
                self.q, self.beta_prior_uniform, True)
            if res is not None:
                return res[1]
        new_params = np.atleast_2d(new_params)
        old_params = np.atleast_2d(old_params)
        std = np.atleast_2d(std)
        a = (fmin - old_params) / std
        b = (fmax - old_params) / std
        A = self._get_log_A(new_params, old_params, None, a, b, std, True,
            counts=(np.atleast_2d(counts[0]), np.atleast_2d(counts[1])))
        return np.cumsum(A, axis=1)[:, -1]

    # ------------------------------------------------------------- DP alpha
    def update_DP_alpha(self):
        """libs/CRP.py:386-410"""
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

    # ----------------------------------------------------------- split/merge
    def update_assignments_split_merge(self, ratios=[.75, .25], step_no=5):
        """libs/CRP.py:417-431"""
        K = len(self.cells_per_cluster)
        if K == 1:
            return (self.do_split_move(step_no), 0)
        if K == self.cells_total:
            return (self.do_merge_move(step_no), 1)
        move = np.random.choice([0, 1], p=ratios)
        if move == 0:
            return (self.do_split_move(step_no), move)
        return (self.do_merge_move(step_no), move)

    def _native_move(self, move, step_no):
        """A whole split (0) / merge (1) move as one native call
        (bnpc_sm_move): the proposal, the restricted scans, the acceptance
        test and the writes of an accepted move.  None when the move is not
        done there - no device context / kernel table / native Beta sampler,
        a subclass or an instance that overrides a step of the move, a move
        of at most 2 cells, an element left to SciPy - with the stream and
        the model untouched, and the caller walks the steps itself."""
        ctx = self._dev()
        table = _native_kernels()
        if table is None or not table.gammaln \
                or not getattr(ctx, '_h', None) or not _native_beta() \
                or _lib.env('BNPC_NATIVE_MOVES', '1') == '0' \
                or not _native_moves_allowed() \
                or _overrides_a_move_step(self):
            return None
        ids = np.fromiter(self.cells_per_cluster.keys(), dtype=np.int64)
        sizes = np.fromiter(self.cells_per_cluster.values(), dtype=np.int64)
        done = _lib.sm_move(ctx, table, move, step_no, ids, sizes,
            self.assignment, self.parameters, self.DP_a,
            self.param_proposal_sd, self.FP, self.FN, self.p, self.q,
            self.beta_prior_uniform, TMIN, TMAX, self._beta_mix_const[0],
            VIEW_MOVE)
        if done is None:
            return None
        self._native_moves = getattr(self, '_native_moves', 0) + 1
        accepted, cl_i, cl_j, moved, n_cells, _ = done
        self._rg_view = None            # the move's view was overwritten
        self._note_move('merge' if move else 'split', n_cells, accepted)
        if not accepted:
            return [0, 1]
        if move == 0:
            self.cells_per_cluster[cl_i] -= moved
            self.cells_per_cluster[cl_j] = moved
        else:
            self.cells_per_cluster[cl_i] += moved
            del self.cells_per_cluster[cl_j]
        return [1, 0]

    def _note_move(self, move, n_cells, accepted):
        """Observer of the restricted-Gibbs runs (tests, tools): called as
        _move_hook(move, cells of the move, accepted) after every one."""
        hook = getattr(self, '_move_hook', None)
        if hook is not None:
            hook(move, int(n_cells), bool(accepted))

    def _propose_split(self):
        """libs/CRP.py:434-457: (cluster, its cells with the two anchors
        first and last, (log transition term, sizes of the others))"""
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
        return cl, cells, (ltrans, np.delete(sizes, pos))

    def do_split_move(self, step_no=5):
        """libs/CRP.py:434-481"""
        done = self._native_move(0, step_no)
        if done is not None:
            return done
        cl, cells, size_data = self._propose_split()
        accept, new_assign, new_params = self.run_rg_nc(
            'split', cells, size_data, step_no)
        self._note_move('split', cells.size, accept)
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

    def _propose_merge(self):
        """libs/CRP.py:484-510: (the two clusters, the cells of the second,
        all cells with the anchors first and last, log size term)"""
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
        size_data = np.cumsum(np.log(probs[pos]))[-1] \
            - np.cumsum(np.log(sizes[pos]))[-1]
        return cl_i, cl_j, cells_j, cells, size_data

    def do_merge_move(self, step_no=5):
        """libs/CRP.py:484-524"""
        done = self._native_move(1, step_no)
        if done is not None:
            return done
        cl_i, cl_j, cells_j, cells, size_data = self._propose_merge()
        accept, new_params = self.run_rg_nc('merge', cells, size_data, step_no)
        self._note_move('merge', cells.size, accept)
        if not accept:
            return [0, 1]
        self.parameters[cl_i] = new_params
        self.assignment[cells_j] = cl_i
        self.cells_per_cluster[cl_i] += cells_j.size
        del self.cells_per_cluster[cl_j]
        return [1, 0]

    # -- restricted Gibbs (Jain & Neal 2007), libs/CRP.py:527-820 ----------
    def run_rg_nc(self, move, cells, size_data, scan_no):
        """libs/CRP.py:527-544.  The non-anchor cells of the move are gathered
        ONCE into a device slot view and re-used by every scan; the counts of
        the merged cluster are those of the two halves added."""
        self._rg_open(cells)
        self._rg_init_split(cells)
        self.rg_params_merge = self._beta_draw(*self._rg_all_counts(cells))
        for _ in range(scan_no):
            self._rg_scan_both(cells)
        if move == 'split':
            return self._do_rg_split_MH(cells, size_data)
        return self._do_rg_merge_MH(cells, size_data)

    def _rg_open(self, cells):
        """Device view of a move's cells [i, S..., j] (anchors first and
        last): the scans use rows 1..n-2, the column counts of the two launch
        clusters all of them."""
        self._rg_S = cells[1:-1]
        self._rg_counts = None
        self._rg_view = None
        if self._rg_S.size:
            self._dev().view_set(VIEW_MOVE, cells)
            self._rg_view = np.array(cells, dtype=np.int64)

    def _rg_members(self, cells, which):
        S = cells[1:-1]
        anchor = cells[0] if which == 0 else cells[-1]
        return np.append(S[np.argwhere(self.rg_assignment == which)], anchor)

    def _rg_split_counts(self, cells):
        """Column counts of the two launch clusters for the CURRENT
        rg_assignment: one launch, cached until the assignment changes."""
        key = np.asarray(self.rg_assignment, dtype=np.int64).tobytes()
        if self._rg_counts is None or self._rg_counts[0] != key:
            view = getattr(self, '_rg_view', None)
            if cells.size <= 4:
                cnt = [self._counts_of(self._rg_members(cells, g))
                    for g in range(2)]
            else:
                if view is not None and view.size == cells.size \
                        and np.array_equal(view, cells):
                    # the two launch clusters are two segments of the move's
                    # own view: anchors fixed, the rest by rg_assignment
                    labels = np.empty(cells.size, dtype=np.int64)
                    labels[0], labels[-1] = 0, 1
                    labels[1:-1] = self.rg_assignment
                    n1, n0 = self._dev().view_counts(VIEW_MOVE, labels, 2)
                else:
                    n1, n0 = self._dev().colcounts(
                        [self._rg_members(cells, g) for g in range(2)])
                cnt = [(n1[g].astype(np.float64), n0[g].astype(np.float64))
                    for g in range(2)]
            self._rg_counts = (key, cnt)
        return self._rg_counts[1]

    def _rg_all_counts(self, cells):
        ci, cj = self._rg_split_counts(cells)
        return ci[0] + cj[0], ci[1] + cj[1]

    def _rg_init_split(self, cells, random=False):
        """libs/CRP.py:547-567.  `ll_j > ll_i` is the one DISCRETE decision on
        the path, so these two sums are reproduced bit for bit: tables built
        by the same NumPy expression as the reference's elements, summed in
        mutation order by the device."""
        i, j, S = cells[0], cells[-1], cells[1:-1]
        if S.size == 0:
            self.rg_assignment = np.array([], dtype=np.int64)
        elif random:
            self.rg_assignment = np.random.choice([0, 1], size=(S.size)) \
                .astype(np.int64)
        else:
            fill = self._beta_mix_const[0]
            anchors = np.nan_to_num(self.data[[i, j]], nan=fill)
            L1, L0 = self._tables(anchors)
            ll = self._dev().ll_tables(VIEW_MOVE, L1, L0)[1:-1]
            self.rg_assignment = np.where(ll[:, 1] > ll[:, 0], 1, 0) \
                .astype(np.int64)
        self._rg_counts = None
        ci, cj = self._rg_split_counts(cells)
        par_i = self._beta_draw(*ci)
        par_j = self._beta_draw(*cj)
        self.rg_params_split = np.stack([par_i, par_j])

    def _rg_scan_split(self, cells, trans_prob=False):
        """libs/CRP.py:570-578"""
        if trans_prob:
            fused = self._rg_scan_fused(cells, final=True)
            if fused is not None:
                return fused
        if cells.size == 2:
            prob_cl = 0
        else:
            prob_cl = self._rg_scan_assign(cells, trans_prob)
        prob_par = self._rg_scan_params(cells, trans_prob)
        if trans_prob:
            return prob_cl + prob_par

    def _rg_scan_both(self, cells):
        """One intermediate scan of the launch states, = _rg_scan_split
        followed by _rg_scan_merge (libs/CRP.py:535-537): the restricted
        assignment scan, then the parameter updates of the two split clusters
        and of the merged cluster.  No draw separates the three MH updates and
        they do not depend on each other, so they are ONE batch (rows in the
        reference's order i, j, merge)."""
        if self._rg_scan_fused(cells) is not None:
            return
        if cells.size != 2:
            self._rg_scan_assign(cells)
        ci, cj = self._rg_split_counts(cells)
        counts = (np.stack([ci[0], cj[0], ci[0] + cj[0]]),
            np.stack([ci[1], cj[1], ci[1] + cj[1]]))
        old = np.concatenate([self.rg_params_split,
            self.rg_params_merge[None, :]])
        new, _, _ = self._mh_batch(old, counts, False)
        self.rg_params_split = new[:2]
        self.rg_params_merge = new[2]

    def _rg_scan_fused(self, cells, final=False):
        """A whole scan as ONE native call (bnpc_rg_scan_step: device sums,
        assignment scan, counts, parameter batch): the intermediate scans
        (launch clusters + merged cluster) or, `final`, the scored last scan
        of a split (launch clusters only, transition probabilities; returns
        their sum).  None if it does not apply - tiny moves, no native kernel
        table, a view that is not this move's - and nothing was drawn."""
        table = _native_kernels()
        ctx = self._dev()
        view = getattr(self, '_rg_view', None)
        if table is None or cells.size <= 4 or not getattr(ctx, '_h', None) \
                or view is None or view.size != cells.size \
                or not np.array_equal(view, cells) \
                or _lib.env('BNPC_RG_FUSED', '1') == '0':
            return None
        rg = np.array(self.rg_assignment, dtype=np.int64, order='C')
        rows = self.rg_params_split if final else np.concatenate(
            [self.rg_params_split, self.rg_params_merge[None, :]])
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        status, new, n1, n0, draws, scan_prob, probs = _lib.rg_scan_step(
            ctx, table, VIEW_MOVE, cells.size, rg, self.DP_a, rows,
            self.param_proposal_sd, TMIN, TMAX, self.FP, self.FN, self.p,
            self.q, self.beta_prior_uniform, trans_prob=final)
        self.rg_assignment = rg
        self._rg_counts = (rg.tobytes(), [
            (n1[g].astype(np.float64), n0[g].astype(np.float64))
            for g in range(2)])
        if status != 0:     # the batch left an element to SciPy
            new, probs, _ = self._mh_batch(rows, (n1, n0), final, draws=draws)
        self.rg_params_split = new[:2]
        if final:
            return scan_prob + probs.sum()
        self.rg_params_merge = new[2]
        return True

    def _rg_scan_merge(self, cells, trans_prob=False):
        """libs/CRP.py:581-587"""
        self.rg_params_merge, prob, _ = self.MH_cluster_params(
            self.rg_params_merge, cells, trans_prob,
            counts=self._rg_all_counts(cells))
        if trans_prob:
            return prob

    def _rg_scan_params(self, cells, trans_prob=False):
        """libs/CRP.py:590-606 (both launch clusters in one batch)"""
        ci, cj = self._rg_split_counts(cells)
        counts = (np.stack([ci[0], cj[0]]), np.stack([ci[1], cj[1]]))
        self.rg_params_split, prob, _ = self._mh_batch(
            self.rg_params_split, counts, trans_prob)
        if trans_prob:
            return prob.sum()

    def _rg_native_scan(self, mode, ll, target=None):
        lib = _lib.load()
        S = ll.shape[0]
        rg = _lib.as_i64(self.rg_assignment)
        out = C.c_double(0.0)
        ll = np.ascontiguousarray(ll, dtype=np.float64)
        i64, f64 = C.c_int64, C.c_double
        if mode == 0:
            with _lib.NumpyStream() as rng:
                _lib.check(lib.bnpc_rg_scan(rng, 0, S, _lib.ptr(ll, f64),
                    float(self.DP_a), _lib.ptr(rg, i64), None, C.byref(out)),
                    'rg_scan')
        else:
            target = _lib.as_i64(target)
            _lib.check(lib.bnpc_rg_scan(None, 1, S, _lib.ptr(ll, f64),
                float(self.DP_a), _lib.ptr(rg, i64), _lib.ptr(target, i64),
                C.byref(out)), 'rg_scan')
        self.rg_assignment = rg
        self._rg_counts = None
        return out.value

    def _rg_scan_assign(self, cells, trans_prob=False):
        """libs/CRP.py:609-632"""
        ll = self._rg_get_ll(cells[1:-1], self.rg_params_split)
        prob = self._rg_native_scan(0, ll)
        if trans_prob:
            return prob

    def _rg_get_ll(self, cells, params):
        """libs/CRP.py:635-638: (|cells| x 2) in one launch on the move's
        slot view (cells must be the move's non-anchor cells)."""
        theta = np.stack([params[0], params[1]]).astype(np.float32)
        view = getattr(self, '_rg_view', None)
        if view is not None and view.size == len(cells) + 2 \
                and np.array_equal(view[1:-1], cells):
            return self._dev().ll_theta(VIEW_MOVE, theta, self.FP,
                self.FN)[1:-1]
        # not the current move's cells: a view of their own
        self._rg_view = None
        self._dev().view_set(VIEW_MOVE, np.asarray(cells))
        return self._dev().ll_theta(VIEW_MOVE, theta, self.FP, self.FN)

    def _do_rg_split_MH(self, cells, size_data):
        """libs/CRP.py:641-653"""
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

    def _get_trans_prob_ratio_split(self, cells):
        """libs/CRP.py:668-682"""
        gs_split = self._rg_scan_split(cells, trans_prob=True)
        std = np.random.choice(self.param_proposal_sd, size=self.muts_total)
        gs_merge = self._log_A_sum(
            self.parameters[self.assignment[cells[0]]], self.rg_params_merge,
            std, self._rg_all_counts(cells), TMIN, TMAX)[0]
        return gs_merge - gs_split

    def _get_trans_prob_ratio_merge(self, cells):
        """libs/CRP.py:685-692"""
        gs_merge = self._rg_scan_merge(cells, trans_prob=True)
        gs_split = self._rg_get_split_prob(cells)
        return gs_split - gs_merge

    def _rg_sizes(self):
        n = self.rg_assignment.size + 2
        n_j = int(np.sum(self.rg_assignment)) + 1
        return n, n - n_j, n_j

    def _get_lprior_ratio_split(self, cells):
        """libs/CRP.py:695-713"""
        n, n_i, n_j = self._rg_sizes()
        ratio = np.log(self.DP_a) - gammaln(n)
        if n_i > 0:
            ratio += gammaln(n_j)
        if n_j > 0:
            ratio += gammaln(n_i)
        if not self.beta_prior_uniform:
            cl = self.assignment[cells[0]]
            ratio += np.cumsum(fastdist.beta_logpdf(
                    self.rg_params_split, self.p, self.q).ravel())[-1] \
                - np.cumsum(fastdist.beta_logpdf(
                    self.parameters[cl], self.p, self.q))[-1]
        return ratio

    def _get_ll_ratio(self, cells, move):
        """libs/CRP.py:716-733: three flat sums = counts . tables"""
        ci, cj = self._rg_split_counts(cells)
        call = (ci[0] + cj[0], ci[1] + cj[1])
        ll_i = self._subset_ll(self.rg_params_split[0], ci).sum()
        ll_j = self._subset_ll(self.rg_params_split[1], cj).sum()
        ll_all = self._subset_ll(self.rg_params_merge, call).sum()
        if move == 'split':
            return ll_i + ll_j - ll_all
        return ll_all - ll_i - ll_j

    def _get_lprior_ratio_merge(self, cells):
        """libs/CRP.py:736-754"""
        n, n_i, n_j = self._rg_sizes()
        n = cells.size
        ratio = gammaln(n) - np.log(self.DP_a)
        if n_i > 0:
            ratio -= gammaln(n_i)
        if n_j > 0:
            ratio -= gammaln(n_j)
        if not self.beta_prior_uniform:
            cls = self.assignment[[cells[0], cells[-1]]]
            ratio += np.cumsum(fastdist.beta_logpdf(
                    self.rg_params_merge, self.p, self.q))[-1] \
                - np.cumsum(fastdist.beta_logpdf(
                    self.parameters[cls], self.p, self.q).ravel())[-1]
        return ratio

    def _get_ltrans_prob_size_ratio_split(self, ltrans_prob_size, cluster_size):
        """libs/CRP.py:757-764"""
        _, n_i, n_j = self._rg_sizes()
        norm = np.cumsum(1 / np.append(cluster_size, [n_i, n_j]))[-1]
        rev = np.log(1 / n_i / norm) + np.log(1 / n_j / norm)
        return rev - ltrans_prob_size[0]

    def _get_ltrans_prob_size_ratio_merge(self, trans_prob_size):
        """libs/CRP.py:767-774"""
        try:
            rev = -np.log(self.cells_total) \
                - np.log(self.rg_assignment.size - 1)
        except FloatingPointError:
            rev = -np.log(self.cells_total)
        return rev - trans_prob_size

    def _rg_get_split_prob(self, cells):
        """libs/CRP.py:777-820"""
        std = np.random.choice(self.param_proposal_sd,
            size=(2, self.muts_total))

        i, j, S = cells[0], cells[-1], cells[1:-1]
        cl_i = self.assignment[i]
        cl_j = self.assignment[j]
        cnt = self._rg_split_counts(cells)
        # both rows in one batch; forward bounds 0 / 1 (libs/CRP.py:779-780)
        prob_i, prob_j = self._log_A_sum(
            self.parameters[[cl_i, cl_j]], self.rg_params_split, std,
            (np.stack([cnt[0][0], cnt[1][0]]),
                np.stack([cnt[0][1], cnt[1][1]])), 0, 1)

        if S.size == 0:
            return prob_i + prob_j + 0.0
        ll = self._rg_get_ll(S, (self.parameters[cl_i], self.parameters[cl_j]))
        target = np.where(self.assignment[S] == cl_i, 0, 1)
        prob_assign = self._rg_native_scan(1, ll, target)
        return prob_i + prob_j + prob_assign


class CRP_errors_learning(CRP):
    """libs/CRP_learning_errors.py:17-111"""

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
        """libs/CRP_learning_errors.py:47-49"""
        return super().get_lprior_full() \
            + self._error_prior_logpdf('FP', self.FP) \
            + self._error_prior_logpdf('FN', self.FN)

    def _error_prior_logpdf(self, which, x):
        """FP_prior.logpdf / FN_prior.logpdf without the frozen wrapper (the
        current rates are asked for at every step: remembered by value)."""
        a, b, mean, sd = (self.FP_prior if which == 'FP'
            else self.FN_prior).args
        return self._memo((which, float(x)),
            lambda: _tn_logpdf_scalar(x, a, b, mean, sd))

    def update_error_rates(self):
        """libs/CRP_learning_errors.py:52-55"""
        self.FP, FP_count = self.MH_error_rates('FP')
        self.FN, FN_count = self.MH_error_rates('FN')
        return FP_count, FN_count

    def get_ll_full_error(self, FP, FN):
        """libs/CRP_learning_errors.py:58-63"""
        return float(self._ll_total([FP], [FN])[0])

    def MH_error_rates(self, error_type):
        """libs/CRP_learning_errors.py:66-111; the new/old likelihood pair is
        ONE launch over the resident per-cluster counts."""
        if error_type == 'FP':
            old, sds = self.FP, self.FP_sd
        else:
            old, sds = self.FN, self.FN_sd

        std = np.random.choice(sds)
        a = (0 - old) / std
        b = (1 - old) / std
        try:
            new = _tn_rvs_scalar(a, b, old, std)
        except FloatingPointError:
            new = truncnorm.rvs(a, np.inf, loc=old, scale=std)

        fwd = _tn_logpdf_scalar(new, a, b, old, std)
        rev = _tn_logpdf_scalar(old, (0 - new) / std, (1 - new) / std, new,
            std)

        if error_type == 'FP':
            new_ll, old_ll = self._ll_total([new, old], [self.FN, self.FN])
        else:
            new_ll, old_ll = self._ll_total([self.FP, self.FP], [new, old])

        A = new_ll + self._error_prior_logpdf(error_type, new) - old_ll \
            - self._error_prior_logpdf(error_type, old) + rev - fwd
        if np.log(np.random.random()) < A:
            return new, [1, 0]
        return old, [0, 1]
