"""CPU-only tests of the product's HOST side: the model surface of
bnpc_amd.model and the native sequential sweeps of libbnpc_hip.so, driven with
a NumPy stand-in for the device primitives (tests/fake_device.py), against the
CPU oracle on the same seeds: identical assignment trajectories, traces to
1e-9."""
import contextlib
import copy
import ctypes as C
import io
import os
import pickle

import numpy as np
import pytest

from oracle import crp_numpy as O
from bnpc_amd import _lib, model as P
from bnpc_amd.mcmc import MCMC
from fake_device import FakeContext


def synth(seed, N, M, C_, miss, FP_true=0.001, FN_true=0.1):
    rng = np.random.RandomState(seed)
    geno = (rng.random_sample((C_, M)) < 0.3)
    z = rng.randint(0, C_, N)
    X = geno[z]
    u = rng.random_sample((N, M))
    obs = np.where(X == 1, u >= FN_true, u < FP_true).astype(np.float64)
    obs[rng.random_sample((N, M)) < miss] = np.nan
    return obs


def make(mod, kind, data, pb=(.25, .25)):
    if kind == 'fixed':
        return mod.CRP(data, DP_alpha=[-1, -1], param_beta=list(pb),
            FN_error=0.1, FP_error=0.001)
    return mod.CRP_errors_learning(data, DP_alpha=[-1, -1],
        param_beta=list(pb), FP_mean=0.01, FP_sd=0.01, FN_mean=0.2,
        FN_sd=0.1)


def run_chain(model, steps, seed, sm_prob=.33, sm_steps=3, eup=0.):
    mcmc = MCMC(model, sm_prob=sm_prob, dpa_prob=.25, error_prob=eup,
        sm_ratios=[.75, .25], sm_steps=sm_steps)
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((steps, int(steps * .33)), seed, 1, 0, '', True)
    return mcmc.get_results()[0]


@pytest.fixture(autouse=True)
def fake_device(monkeypatch):
    monkeypatch.setattr(_lib, 'Context', FakeContext)


# --------------------------------------------------------------- MT replica
def test_mt19937_replica_matches_numpy(golden_dir):
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, 'rng.npz'))
    for seed in (1, 42, 1608637542):
        np.random.seed(seed)
        st, extra = _lib.rng_export()
        got = [lib.bnpc_mt_random_sample(C.byref(st)) for _ in range(5)]
        assert np.array_equal(got, g[f's{seed}_random'])
        perm = np.empty(37, dtype=np.int64)
        lib.bnpc_mt_permutation(C.byref(st), 37, _lib.ptr(perm, C.c_int64))
        assert np.array_equal(perm, g[f's{seed}_perm'])
        # hand the stream back to numpy: it continues where C stopped
        _lib.rng_import(st, extra)
        assert np.array_equal(np.random.randint(0, 1000, size=11),
            g[f's{seed}_randint'])


def test_mt19937_long_run_and_big_permutation():
    lib = _lib.load()
    np.random.seed(2024)
    st, extra = _lib.rng_export()
    ref = np.random.random(3000)           # crosses several 624-word refills
    got = np.array([lib.bnpc_mt_random_sample(C.byref(st))
        for _ in range(3000)])
    assert np.array_equal(ref, got)
    for n in (1, 2, 3, 1000, 5000, 70000):
        np.random.seed(n)
        st, extra = _lib.rng_export()
        ref = np.random.permutation(n)
        out = np.empty(n, dtype=np.int64)
        lib.bnpc_mt_permutation(C.byref(st), n, _lib.ptr(out, C.c_int64))
        assert np.array_equal(ref, out)
        _lib.rng_import(st, extra)
        ref_next = np.random.get_state()[1].copy()
        np.random.seed(n)
        np.random.permutation(n)
        assert np.array_equal(np.random.get_state()[1], ref_next)


# ----------------------------------------------------------- single moves
@pytest.mark.parametrize('budget', [None, '30000'])
def test_tiled_sweep_equals_whole_matrix_sweep(budget, monkeypatch):
    """BNPC_SWEEP_BYTES: a sweep whose N x K matrix exceeds the budget is
    tiled over permutation-ordered cell chunks (each tile only against the
    clusters alive at its start); decisions and draws do not change."""
    if budget:
        monkeypatch.setenv('BNPC_SWEEP_BYTES', budget)
    else:
        monkeypatch.delenv('BNPC_SWEEP_BYTES', raising=False)
    test_gibbs_sweeps_match_oracle('learn')
    test_gibbs_opens_many_clusters_in_one_sweep()
    if budget:
        data = synth(1, 150, 70, 4, 0.15)
        p = make(P, 'fixed', data)
        np.random.seed(11)
        p.init()
        np.random.seed(1)
        p.update_assignments_Gibbs()
        assert p._ctx.calls['view_set'] >= 3      # several tiles were used
        assert p._ctx.calls['ll_rows_issue'] >= 3 and not p._ctx.tiles


def test_tiles_issued_ahead_never_serve_a_reborn_cluster(monkeypatch):
    """Tile t+1 is evaluated while tile t is walked.  A cluster that dies in
    tile t and whose id is re-used by a cluster born in tile t has a stale
    column in tile t+1 (the stand-in device poisons it with NaN): the sweep
    must evaluate that cluster afresh - and still equal the oracle."""
    monkeypatch.setenv('BNPC_SWEEP_BYTES', '20000')
    rng = np.random.RandomState(8)
    profiles = (rng.random_sample((30, 40)) < 0.5).astype(float)
    data = np.repeat(profiles, 8, axis=0)[rng.permutation(240)]
    data[rng.random_sample(data.shape) < 0.05] = np.nan
    poisoned = 0
    for seed in range(6):
        models = []
        for mod in (O, P):
            m = mod.CRP(data, DP_alpha=[60, 1], param_beta=[.25, .25],
                FN_error=0.02, FP_error=0.02)
            np.random.seed(seed)
            m.init()                # ~150 clusters of 1-2 cells: many die
            np.random.seed(50 + seed)
            m.update_assignments_Gibbs()
            models.append(m)
        o, p = models
        assert np.array_equal(o.assignment, p.assignment)
        assert list(o.cells_per_cluster.items()) == \
            list(p.cells_per_cluster.items())
        assert p._ctx.calls['ll_rows_issue'] >= 3
        poisoned += getattr(p._ctx, 'poisoned', 0)
    assert poisoned > 0         # the situation did occur


@pytest.mark.parametrize('kind', ['fixed', 'learn'])
def test_gibbs_sweeps_match_oracle(kind):
    data = synth(1, 150, 70, 4, 0.15)
    o = make(O, kind, data)
    p = make(P, kind, data)
    for m in (o, p):
        np.random.seed(11)
        m.init()
    assert np.array_equal(o.assignment, p.assignment)
    assert np.array_equal(o.parameters, p.parameters)
    for sweep in range(4):
        for m in (o, p):
            np.random.seed(100 + sweep)
            m.update_assignments_Gibbs()
            tail = np.random.random(3)
            m._tail = tail
        assert np.array_equal(o.assignment, p.assignment), sweep
        assert list(o.cells_per_cluster.items()) == \
            list(p.cells_per_cluster.items())
        assert np.array_equal(o._tail, p._tail)       # same RNG consumption
        ids = list(o.cells_per_cluster)
        assert np.array_equal(o.parameters[ids], p.parameters[ids])
        np.testing.assert_allclose(p.get_ll_full(), o.get_ll_full(),
            rtol=1e-12)


def test_update_parameters_and_errors_match_oracle():
    data = synth(2, 120, 60, 3, 0.2)
    o = make(O, 'learn', data)
    p = make(P, 'learn', data)
    for m in (o, p):
        np.random.seed(5)
        m.init()
        m.update_assignments_Gibbs()
    for rnd in range(3):
        outs = []
        for m in (o, p):
            np.random.seed(50 + rnd)
            r1 = m.update_parameters()
            r2 = m.update_error_rates()
            outs.append((r1, r2, m.FP, m.FN, np.random.random()))
        assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
        np.testing.assert_allclose(outs[1][2:], outs[0][2:], rtol=1e-12)
        ids = list(o.cells_per_cluster)
        assert np.array_equal(o.parameters[ids], p.parameters[ids])
        np.testing.assert_allclose(p.get_lprior_full(), o.get_lprior_full(),
            rtol=1e-11)
        np.testing.assert_allclose(
            p.get_ll_full_error(0.02, 0.3), o.get_ll_full_error(0.02, 0.3),
            rtol=1e-12)


@pytest.mark.parametrize('pb', [(.25, .25), (1, 1)])
@pytest.mark.parametrize('start', ['together', 'fragmented', 'gibbs'])
def test_split_merge_moves_match_oracle(pb, start):
    data = synth(3, 90, 50, 3, 0.1)
    o = make(O, 'fixed', data, pb)
    p = make(P, 'fixed', data, pb)
    for m in (o, p):
        np.random.seed(8)
        if start == 'together':         # splits get accepted
            m.init(mode='together')
        elif start == 'fragmented':     # merges get accepted
            m.init(assign=list(np.random.RandomState(1).randint(0, 12, 90)))
        else:
            m.init()
            m.update_assignments_Gibbs()
        m.update_parameters()
    n_acc = 0
    for rnd in range(40):
        outs = []
        for m in (o, p):
            np.random.seed(1000 + rnd)
            res = m.update_assignments_split_merge([.6, .4], 2)
            outs.append((res, np.random.random()))
        assert outs[0][0] == outs[1][0], rnd
        assert outs[0][1] == outs[1][1], rnd
        assert np.array_equal(o.assignment, p.assignment), rnd
        assert list(o.cells_per_cluster.items()) == \
            list(p.cells_per_cluster.items())
        ids = list(o.cells_per_cluster)
        assert np.array_equal(o.parameters[ids], p.parameters[ids])
        n_acc += outs[0][0][0][0]
        if rnd % 5 == 4:
            for m in (o, p):
                np.random.seed(rnd)
                m.update_parameters()
    if start != 'gibbs':
        assert n_acc > 0      # accepted moves were exercised


def test_rg_init_split_is_bit_identical_to_oracle():
    """The discrete `ll_j > ll_i` decision, on data with many exact ties."""
    rng = np.random.RandomState(0)
    data = (rng.random_sample((80, 24)) < 0.5).astype(float)
    data[rng.random_sample(data.shape) < 0.3] = np.nan
    o = make(O, 'fixed', data)
    p = make(P, 'fixed', data)
    for m in (o, p):
        m.init(mode='together')
    cells = np.arange(80)
    for trial in range(10):
        rng.shuffle(cells)
        for m in (o, p):
            np.random.seed(trial)
            if m is p:
                m._rg_open(cells.copy())
            m._rg_init_split(cells.copy())
        assert np.array_equal(o.rg_assignment, p.rg_assignment)
        assert np.array_equal(o.rg_params_split, p.rg_params_split)


# -------------------------------------------------------------- whole chains
@pytest.mark.parametrize('kind,eup', [('fixed', 0.), ('learn', .25)])
def test_chain_matches_oracle(kind, eup):
    data = synth(4, 80, 40, 3, 0.1)
    ro = run_chain(make(O, kind, data), 60, 7, sm_prob=.4, sm_steps=2, eup=eup)
    rp = run_chain(make(P, kind, data), 60, 7, sm_prob=.4, sm_steps=2, eup=eup)
    assert np.array_equal(ro['assignments'], rp['assignments'])
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        np.testing.assert_allclose(rp[key], ro[key], rtol=1e-9, err_msg=key)
    assert np.array_equal(ro['params'], rp['params'])


def test_other_init_modes_match_oracle():
    data = synth(6, 40, 30, 3, 0.1)
    for mode, assign in (('separate', False), ('together', False),
            ('random', [0, 1] * 20), ('random', [3] * 39 + [7])):
        o = make(O, 'fixed', data)
        p = make(P, 'fixed', data)
        for m in (o, p):
            np.random.seed(3)
            m.init(mode=mode, assign=assign)
        assert np.array_equal(o.assignment, p.assignment)
        assert o.cells_per_cluster == p.cells_per_cluster
        ids = list(o.cells_per_cluster)
        assert np.array_equal(o.parameters[ids], p.parameters[ids])
        np.testing.assert_allclose(p.get_ll_full(), o.get_ll_full(),
            rtol=1e-12)
    with pytest.raises(TypeError):
        make(P, 'fixed', data).init(mode='nope')


def test_pickle_and_deepcopy_drop_the_device_context():
    data = synth(5, 30, 20, 2, 0.1)
    p = make(P, 'learn', data)
    np.random.seed(1)
    p.init()
    ll = p.get_ll_full()
    assert p._ctx is not None
    state = p.__getstate__()
    assert 'data' not in state and state['_data_codes'].dtype == np.int8
    assert state['_ctx'] is None
    q = pickle.loads(pickle.dumps(p))
    r = copy.deepcopy(p)
    for clone in (q, r):
        assert clone._ctx is None and clone._lab is None
        assert clone.data.dtype == np.float64
        assert np.array_equal(np.isnan(clone.data), np.isnan(p.data))
        assert np.array_equal(np.nan_to_num(clone.data, nan=9),
            np.nan_to_num(p.data, nan=9))
        assert clone.get_ll_full() == ll        # context rebuilt lazily
        assert clone._ctx is not None


def test_hand_off_is_compact_and_lossless():
    """What crosses the pool's pipes (libs/MCMC.py:113-120): parameter rows of
    populated clusters only, labels in a narrow integer type - and a chain
    that is pickled mid-run continues exactly like one that is not."""
    from bnpc_amd.mcmc import Chain_steps
    data = synth(6, 80, 50, 3, 0.1)
    knobs = dict(sm_prob=.33, dpa_prob=.25, error_prob=.25,
        sm_ratios=[.75, .25], sm_steps=2,
        param_proposal_sd=np.array([0.1, 0.25, 0.5]))

    def start():
        np.random.seed(11)
        m = make(P, 'learn', data)
        m.init()
        chain = Chain_steps(m, 1, 24, 8, knobs, 0, False)
        for i in range(1, 13):
            chain.do_step()
            chain.update_results(i, i < 8)
        return chain

    a, b = start(), start()
    assert len(a.model.cells_per_cluster) * 4 < 80
    state = a.model.__getstate__()
    assert 'parameters' not in state
    shape, _, live, rows = state['_theta_rows']
    assert shape == (80, 50) and rows.shape == (live.size, 50)
    rng = np.random.get_state()
    assert rows.nbytes * 4 < a.model.parameters.nbytes
    wire = a.trace.__getstate__()['data']['assignments']
    assert wire.dtype == np.int16 and np.array_equal(wire,
        a.results['assignments'])
    a = pickle.loads(pickle.dumps(a))
    assert a.results['assignments'].dtype == b.results['assignments'].dtype
    live_ids = sorted(a.model.cells_per_cluster)
    assert np.array_equal(a.model.parameters[live_ids],
        b.model.parameters[live_ids])
    outs = []
    for chain in (a, b):
        np.random.set_state(rng)
        for i in range(13, 25):
            chain.do_step()
            chain.update_results(i, False)
        outs.append(chain.results)
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP', 'assignments', 'params'):
        assert np.array_equal(outs[0][key], outs[1][key]), key


def test_module_path_of_the_drop_in_classes():
    import libs.CRP
    import libs.CRP_learning_errors
    assert libs.CRP.CRP.__module__ == 'libs.CRP'
    assert libs.CRP_learning_errors.CRP_errors_learning.__module__ == \
        'libs.CRP_learning_errors'
    assert issubclass(libs.CRP_learning_errors.CRP_errors_learning, P.CRP)


def test_gibbs_opens_many_clusters_in_one_sweep():
    """> 16 new clusters in one sweep: the ll matrix and the column tables
    grow mid-sweep; ids of clusters that died earlier in the sweep are
    re-used (libs/CRP.py:297-299)."""
    rng = np.random.RandomState(3)
    profiles = (rng.random_sample((40, 60)) < 0.5).astype(float)
    data = np.repeat(profiles, 3, axis=0)
    data[rng.random_sample(data.shape) < 0.05] = np.nan
    models = []
    for mod in (O, P):
        m = mod.CRP(data, DP_alpha=[200, 1], param_beta=[.25, .25],
            FN_error=0.01, FP_error=0.01)
        np.random.seed(4)
        m.init(mode='together')
        k_before = len(m.cells_per_cluster)
        np.random.seed(5)
        m.update_assignments_Gibbs()
        m._tail = np.random.random(2)
        models.append(m)
    o, p = models
    assert k_before == 1 and len(o.cells_per_cluster) > 20
    assert np.array_equal(o.assignment, p.assignment)
    assert list(o.cells_per_cluster.items()) == \
        list(p.cells_per_cluster.items())
    assert np.array_equal(o._tail, p._tail)
    ids = list(o.cells_per_cluster)
    assert np.array_equal(o.parameters[ids], p.parameters[ids])
    # a second sweep from the fragmented state (clusters die and ids recycle)
    for m in (o, p):
        np.random.seed(6)
        m.update_assignments_Gibbs()
    assert np.array_equal(o.assignment, p.assignment)
    assert list(o.cells_per_cluster.items()) == \
        list(p.cells_per_cluster.items())


def test_native_sweep_rejects_corrupt_state():
    lib = _lib.load()
    N, ld = 4, 3
    st = _lib.GibbsState(N, ld, 1, 1, 0, -1, N, -1)
    rng, _ = _lib.rng_export()
    i64, f64 = C.c_int64, C.c_double
    perm = np.arange(N, dtype=np.int64)
    ll = np.zeros((N, ld))
    post_new = np.zeros(N)
    prior = np.zeros(N + 2)
    assignment = np.full(N, 2, dtype=np.int64)      # id 2 has no column
    col_of_id = np.full(N, -1, dtype=np.int64)
    col_of_id[0] = 0
    col_id = np.zeros(ld, dtype=np.int64)
    col_size = np.array([4, 0, 0], dtype=np.int64)
    order = np.zeros(ld, dtype=np.int64)
    scratch = np.zeros(2 * (ld + 1))
    rc = lib.bnpc_gibbs_sweep(C.byref(st), C.byref(rng), _lib.ptr(perm, i64),
        _lib.ptr(ll, f64), _lib.ptr(post_new, f64), _lib.ptr(prior, f64),
        _lib.ptr(assignment, i64), _lib.ptr(col_of_id, i64),
        _lib.ptr(col_id, i64), _lib.ptr(col_size, i64), _lib.ptr(order, i64),
        _lib.ptr(scratch, f64))
    assert rc != 0 and b'unknown cluster' in lib.bnpc_last_error()


def test_threaded_host_batches_are_bit_identical(monkeypatch):
    """Large (clusters x mutations) proposal batches are evaluated on host
    threads; elementwise math, so the split must not change a bit."""
    data = synth(8, 300, 120, 3, 0.1)
    outs = []
    for threads, min_elems in (('1', 1 << 30), ('4', 1 << 13)):
        monkeypatch.setenv('BNPC_HOST_THREADS', threads)
        monkeypatch.setattr(P, '_THREAD_MIN_ELEMS', min_elems)
        P._POOL.clear()
        p = make(P, 'fixed', data)
        np.random.seed(2)
        p.init()                        # ~190 clusters
        np.random.seed(3)
        res = p.update_parameters()
        ids = list(p.cells_per_cluster)
        outs.append((res, p.parameters[ids].copy(), np.random.random()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]
    o = make(O, 'fixed', data)
    np.random.seed(2)
    o.init()
    np.random.seed(3)
    assert o.update_parameters() == outs[1][0]
    assert np.array_equal(o.parameters[ids], outs[1][1])
    P._POOL.clear()


def test_prior_density_cache_changes_nothing(monkeypatch):
    """update_parameters reuses the Beta log-density of unchanged profiles
    (model._known_prior): same draws, same parameters, bit for bit, as with
    the cache switched off - also when rows are overwritten behind its back."""
    data = synth(21, 60, 90, 4, 0.15)

    def walk(cache_elems):
        monkeypatch.setattr(P.CRP, '_PRIOR_CACHE_ELEMS', cache_elems)
        np.random.seed(77)
        m = make(P, 'fixed', data, pb=(.25, .25))
        m.init()
        out = []
        for step in range(8):
            m.update_assignments_Gibbs()
            if step == 3:       # a write the cache has not seen
                first = next(iter(m.cells_per_cluster))
                m.parameters[first, ::7] = np.float32(0.5)
            if step == 5:
                m.update_assignments_split_merge([.75, .25], 2)
            out.append(m.update_parameters())
            out.append(m.parameters[sorted(m.cells_per_cluster)].copy())
        return out, m

    ref, _ = walk(0)
    got, m = walk(1 << 22)
    assert m._prior_rows is not None \
        and set(m._prior_rows[0].tolist()) == set(m.cells_per_cluster)
    for r, g in zip(ref, got):
        assert np.array_equal(np.asarray(r), np.asarray(g))
    # a cached entry is only trusted while theta is unchanged
    ids = np.array(sorted(m.cells_per_cluster))
    theta = m.parameters[ids].copy()
    theta[0, 0] = np.float32(0.123)
    from bnpc_amd import fastdist
    want = np.cumsum(fastdist.beta_logpdf(theta, m.p, m.q).ravel())[-1]
    assert m._prior_density_sum(ids, theta) == want
    monkeypatch.setenv('BNPC_NATIVE_MH', '0')
    P._NATIVE.clear()
    assert m._prior_density_sum(ids, theta) == want
    P._NATIVE.clear()


def _walk_moves(data, kind='learn', steps=10, seed=5):
    np.random.seed(seed)
    m = make(P, kind, data, pb=(.25, .25))
    m.init()
    out = []
    for step in range(steps):
        if step % 3 == 2:
            out.append(repr(m.update_assignments_split_merge([.6, .4], 2)))
        else:
            m.update_assignments_Gibbs()
        out.append(m.update_parameters())
        out.append(m.parameters[sorted(m.cells_per_cluster)].copy())
        out.append(m.get_lprior_full())
        out.append(m.assignment.copy())
    out.append(np.random.random())
    return out


def test_native_parameter_batch_is_in_use_and_changes_nothing(monkeypatch):
    """The native batch of the MH parameter moves (bnpc_mh_batch on SciPy's /
    NumPy's own scalar kernels) passes its start-up bit comparison here, and a
    walk through Gibbs sweeps, split/merge moves and parameter updates is
    bit-identical with it switched off (the SciPy-level array path), with one
    thread and with several."""
    data = synth(31, 80, 140, 4, 0.15)
    walks = []
    for native, threads in (('0', '1'), ('1', '1'), ('1', '4')):
        monkeypatch.setenv('BNPC_NATIVE_MH', native)
        monkeypatch.setenv('BNPC_HOST_THREADS', threads)
        P._NATIVE.clear()
        assert (P._native_kernels() is not None) == (native == '1')
        walks.append(_walk_moves(data))
    P._NATIVE.clear()
    for other in walks[1:]:
        for r, g in zip(walks[0], other):
            assert np.array_equal(np.asarray(r), np.asarray(g))


def test_native_batch_hands_exotic_elements_back(monkeypatch):
    """A uniform that is exactly 0 (log raises in the reference) is not the
    native batch's business: it reports status 1 with the draws taken, and the
    SciPy-level path evaluates the batch from those draws."""
    from bnpc_amd import _lib
    table = P._native_kernels()
    assert table is not None
    rng = np.random.RandomState(3)
    G, M = 2, 50
    old = np.clip(rng.uniform(size=(G, M)), P.TMIN, P.TMAX).astype(np.float32)
    n1 = rng.randint(0, 9, size=(G, M))
    n0 = rng.randint(0, 9, size=(G, M))
    sd = np.array([.1, .25, .5])
    draws = [rng.randint(0, 3, size=(G, M)), rng.uniform(size=(G, M)),
        rng.uniform(size=(G, M))]
    args = (table, old, n1, n0, sd, P.TMIN, P.TMAX, 0.01, 0.2, .25, .25, False,
        False)
    assert _lib.mh_batch(*args, draws=draws)[0] == 0
    draws[2][1, 7] = 0.0
    assert _lib.mh_batch(*args, draws=draws)[0] == 1
    draws[2][1, 7] = 0.5
    draws[1][0, 3] = 0.0
    assert _lib.mh_batch(*args, draws=draws)[0] == 1

    # the model falls back to the array path on status 1, from the same draws
    data = synth(4, 40, M, 3, 0.1)
    res = []
    for force in (False, True):
        np.random.seed(9)
        m = make(P, 'fixed', data, pb=(.25, .25))
        m.init()
        if force:
            real = _lib.mh_batch

            def exotic(*a, **k):
                out = real(*a, **k)
                return (1,) + out[1:]
            monkeypatch.setattr(_lib, 'mh_batch', exotic)
        res.append((m.update_parameters(),
            m.parameters[sorted(m.cells_per_cluster)].copy(),
            np.random.random()))
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2]
    assert np.array_equal(res[0][1], res[1][1])


def test_failed_tiled_sweep_leaves_no_tile_behind(monkeypatch):
    """An exception in the middle of a tiled sweep: the tile issued ahead is
    picked up before the error propagates, so the context is usable again."""
    monkeypatch.setenv('BNPC_SWEEP_BYTES', '30000')
    data = synth(1, 150, 70, 4, 0.15)
    p = make(P, 'fixed', data)
    np.random.seed(11)
    p.init()
    real = P.CRP._gibbs_window
    calls = {'n': 0}

    def flaky(self, *a, **k):
        calls['n'] += 1
        if calls['n'] == 2:
            raise ValueError('boom')
        return real(self, *a, **k)

    monkeypatch.setattr(P.CRP, '_gibbs_window', flaky)
    np.random.seed(1)
    with pytest.raises(ValueError, match='boom'):
        p.update_assignments_Gibbs()
    assert not p._ctx.tiles                      # nothing left in flight
    monkeypatch.setattr(P.CRP, '_gibbs_window', real)
    # assignment and sizes were not touched by the aborted sweep ...
    sizes = dict(zip(*np.unique(p.assignment, return_counts=True)))
    assert sizes == {k: v for k, v in p.cells_per_cluster.items()}
    # ... and the context takes the next tiled sweep
    np.random.seed(2)
    p.update_assignments_Gibbs()
    assert not p._ctx.tiles and p._ctx.calls['ll_rows_issue'] >= 4


@pytest.mark.parametrize('tiled', [False, True])
def test_team_scan_of_many_clusters_changes_nothing(tiled, monkeypatch):
    """The scan of a cell over many live clusters on the host thread team
    (first sweeps; bnpc_gibbs_sweep with threads > 1): same sweeps as the
    oracle, whole-matrix and tiled, dominated and non-dominated cells, with
    the team threshold lowered so that a small case reaches it."""
    monkeypatch.setenv('BNPC_SWEEP_PAR_MIN', '6')
    monkeypatch.setenv('BNPC_HOST_THREADS', '5')
    if tiled:
        monkeypatch.setenv('BNPC_SWEEP_BYTES', '40000')
    # few mutations: neighbouring clusters are close, many cells take the
    # full normalisation; many mutations: one cluster dominates
    for M, seed in ((6, 3), (70, 4)):
        data = synth(seed, 220, M, 4, 0.15)
        o = make(O, 'fixed', data)
        p = make(P, 'fixed', data)
        for m in (o, p):
            np.random.seed(21)
            m.init()
        assert len(p.cells_per_cluster) > 100
        for sweep in range(2):
            for m in (o, p):
                np.random.seed(300 + sweep)
                m.update_assignments_Gibbs()
                m._tail = np.random.random(2)
            assert np.array_equal(o.assignment, p.assignment), (M, sweep)
            assert list(o.cells_per_cluster.items()) == \
                list(p.cells_per_cluster.items())
            assert np.array_equal(o._tail, p._tail)


def test_sweep_hints_change_nothing(monkeypatch):
    """The per-cell top-2 hint (bnpc_top2) lets the native loop skip the scan
    of a cell: sweeps with and without it are the same sweeps - from a
    fragmented start where clusters die and are born under the hint, with
    clear winners (many mutations) and with close calls (few mutations) -
    and both equal the oracle's."""
    for M, seed in ((5, 6), (80, 7)):
        data = synth(seed, 160, M, 3, 0.15)
        runs = []
        for mod, hint in ((O, None), (P, '1'), (P, '0')):
            if hint is not None:
                monkeypatch.setenv('BNPC_SWEEP_HINT', hint)
            m = make(mod, 'fixed', data)
            np.random.seed(17)
            m.init(assign=list(np.random.RandomState(seed).randint(0, 40, 160)))
            out = []
            for sweep in range(3):
                np.random.seed(50 + sweep)
                m.update_assignments_Gibbs()
                out.append((m.assignment.copy(),
                    list(m.cells_per_cluster.items()), np.random.random()))
                m.update_parameters()
            runs.append(out)
        for other in runs[1:]:
            for a, b in zip(runs[0], other):
                assert np.array_equal(a[0], b[0]) and a[1] == b[1] \
                    and a[2] == b[2]


def test_sweep_hint_is_used_and_falls_back():
    """Direct calls: a row whose hint leaves no doubt is decided without
    reading the matrix (a poisoned row goes unnoticed), a row whose runner-up
    is close is scanned (the poison is seen)."""
    from bnpc_amd import _lib
    import ctypes as C
    lib = _lib.load()
    N, K, ld = 6, 3, 4
    crp = np.append(0, O.CRP.log_CRP_prior(
        np.append(np.arange(1, N + 1), 2.0), N, 2.0))
    sizes = np.array([2, 2, 2], dtype=np.int64)
    col_prior = np.ascontiguousarray(crp[sizes])
    ll = np.full((N, ld), -500.0)
    truth = np.array([0, 0, 1, 1, 2, 2])
    ll[np.arange(N), truth] = -100.0
    hint = np.zeros(N, dtype=_lib.TOP2)
    hint['best'] = -100.0 + col_prior[truth]
    hint['second'] = -500.0 + col_prior.max()
    hint['col'] = truth
    hint['col2'] = hint['col3'] = -1
    hint['third'] = hint['fourth'] = -np.inf
    post_new = np.full(N, -800.0)

    def sweep(mat, hints):
        np.random.seed(3)
        perm = _lib.as_i64(np.random.permutation(N))
        assignment = truth.astype(np.int64).copy()
        col_of_id = np.full(N, -1, dtype=np.int64)
        col_of_id[:K] = np.arange(K)
        col_id = np.array([0, 1, 2, -1], dtype=np.int64)
        col_size = np.array([2, 2, 2, 0], dtype=np.int64)
        order = np.array([0, 1, 2, 0], dtype=np.int64)
        scratch = np.empty(2 * (ld + 1))
        st = _lib.GibbsState(N, ld, K, K, 0, -1, N, -1, 1)
        if hints is not None:
            st.hint = hints.ctypes.data
            st.hint_prior = col_prior.ctypes.data
            st.hint_cols = K
        with _lib.NumpyStream() as rng:
            _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), rng, _lib.ptr(perm),
                _lib.ptr(mat), _lib.ptr(post_new), _lib.ptr(crp),
                _lib.ptr(assignment), _lib.ptr(col_of_id), _lib.ptr(col_id),
                _lib.ptr(col_size), _lib.ptr(order), _lib.ptr(scratch)),
                'sweep')
        used.append(int(st.hint_used))
        return assignment, np.random.random()

    used = []
    want = sweep(ll, None)
    assert np.array_equal(want[0], truth)
    poisoned = np.full_like(ll, np.nan)
    got = sweep(poisoned, hint)         # never looks at the matrix
    assert np.array_equal(got[0], want[0]) and got[1] == want[1]
    close = hint.copy()
    close['second'] = close['best'] - 1.0   # a runner-up 1 nat away: scan
    got = sweep(ll, close)
    assert np.array_equal(got[0], want[0]) and got[1] == want[1]

    # a cell torn between TWO columns, the third far below: decided from the
    # two entries of the hint under the current priors - same draws, same
    # picks as the full scan of the matrix, which is never read
    rival = (truth + 1) % K
    torn = ll.copy()
    torn[np.arange(N), rival] = -100.7
    pair = np.zeros(N, dtype=_lib.TOP2)
    pair['col'], pair['col2'] = truth, rival
    pair['ll_best'], pair['ll_second'] = -100.0, -100.7
    pair['e2'] = np.exp(-0.7)
    pair['best'] = -100.0 + col_prior[truth]
    pair['second'] = -100.7 + col_prior[rival]
    pair['third'] = -500.0 + col_prior.max()
    pair['col3'], pair['fourth'] = -1, -np.inf
    want = sweep(torn, None)
    assert not np.array_equal(want[0], truth)       # some cells do move
    got = sweep(torn, pair)
    assert np.array_equal(got[0], want[0]) and got[1] == want[1]
    assert used[-1] >= 4        # (a cell whose hinted cluster died is scanned)
    # (the same with a prior table that is NOT log(size) + const: the loop
    # notices and takes the exponentials of the re-scored entries instead of
    # the record's weights, which it poisons here)
    crp_true = crp.copy()
    crp[2:] += 0.01 * np.arange(crp.size - 2)
    odd = pair.copy()
    odd['best'] = -100.0 + crp[sizes][truth]
    odd['second'] = -100.7 + crp[sizes][rival]
    odd['e2'] = 1e3
    col_prior[:] = crp[sizes]
    want_odd = sweep(torn, None)
    got = sweep(torn, odd)
    assert np.array_equal(got[0], want_odd[0]) and got[1] == want_odd[1]
    assert used[-1] >= 4
    crp[:] = crp_true
    col_prior[:] = crp[sizes]
    # ... but not when the third entry is within reach: every cell is scanned
    near = pair.copy()
    near['third'] = near['second'] - 30.0
    got = sweep(torn, near)
    assert np.array_equal(got[0], want[0]) and got[1] == want[1]
    assert used[-1] == 0


def test_given_row_acceptance_ratios_equal_the_array_path(monkeypatch):
    """bnpc_log_accept (the transition terms of the split / merge ratios)
    against CRP._get_log_A through SciPy, both truncation settings, with the
    native path switched off as well."""
    from bnpc_amd import _lib
    rng = np.random.RandomState(5)
    M = 300
    data = synth(3, 50, M, 3, 0.1)
    m = make(P, 'fixed', data)
    new = np.clip(rng.uniform(size=(2, M)), P.TMIN, P.TMAX).astype(np.float32)
    old = np.clip(rng.uniform(size=(2, M)), P.TMIN, P.TMAX).astype(np.float32)
    new[:, :20] = np.float32(P.TMIN)
    old[:, 10:40] = np.float32(P.TMAX)
    std = rng.choice(m.param_proposal_sd, size=(2, M))
    counts = (rng.randint(0, 30, size=(2, M)), rng.randint(0, 30, size=(2, M)))
    for fmin, fmax in ((P.TMIN, P.TMAX), (0, 1)):
        a, b = (fmin - old) / std, (fmax - old) / std
        want = np.cumsum(m._get_log_A(new, old, None, a, b, std, True,
            counts=counts), axis=1)[:, -1]
        monkeypatch.setenv('BNPC_NATIVE_MH', '1')
        P._NATIVE.clear()
        assert P._native_kernels() is not None
        assert np.array_equal(m._log_A_sum(new, old, std, counts, fmin, fmax),
            want)
        monkeypatch.setenv('BNPC_NATIVE_MH', '0')
        P._NATIVE.clear()
        assert np.array_equal(m._log_A_sum(new, old, std, counts, fmin, fmax),
            want)
    P._NATIVE.clear()


def test_screened_parameter_batch_equals_the_plain_batch():
    """bnpc_mh_batch with screen verdicts (include/bnpc_hip.h: 0 = declined
    for certain): any screen that only rules out proposals the exact
    arithmetic declines gives the plain batch's results bit for bit - new
    parameters, declined counts, the prior densities of the result with and
    without the prior cache - whatever subset it rules out (none, some, all
    of the declined ones), on 1 and 3 threads."""
    table = P._native_kernels()
    if table is None:
        pytest.skip('native parameter batch not available')
    rng = np.random.RandomState(12)
    G, M = 4, 333
    sd = np.array([0.1, 0.25, 0.5])
    for (p, q), uniform in (((.25, .25), False), ((1, 1), True)):
        old = np.clip(rng.uniform(size=(G, M)), P.TMIN, P.TMAX) \
            .astype(np.float32)
        n1 = rng.randint(0, 40, (G, M)).astype(np.int32)
        n0 = rng.randint(0, 40, (G, M)).astype(np.int32)
        draws = (rng.randint(0, 3, (G, M)).astype(np.int32),
            rng.uniform(size=(G, M)), rng.uniform(size=(G, M)))
        known = None
        if not uniform:
            from bnpc_amd import fastdist
            known = (old.copy(), fastdist.beta_logpdf(old, p, q))
            known[0][:, ::3] = np.float32(0.321)    # a third are misses
        plain = _lib.mh_batch(table, old, n1, n0, sd, P.TMIN, P.TMAX, .01, .2,
            p, q, uniform, False, known=known, want_prior=True, draws=draws,
            threads=1)
        assert plain[0] == 0
        declined = plain[1].view(np.int32) == old.view(np.int32)
        assert 0 < declined.sum() < G * M
        for frac in (0.0, 0.6, 1.0):
            screen = np.ones((G, M), dtype=np.uint8)
            screen[declined & (rng.uniform(size=(G, M)) < frac)] = 0
            # ... and "accepted for certain" for some of the accepted ones
            screen[~declined & (rng.uniform(size=(G, M)) < frac)] = 2
            for threads in (1, 3, 8):
                a = _lib.MHArgs()
                new = np.empty((G, M), np.float32)
                prior = np.empty((G, M))
                A = np.empty((G, M))
                lp, dec = np.empty(G), np.empty(G, np.int64)
                a.G, a.M = G, M
                a.old_theta, a.n1, a.n0 = (x.ctypes.data
                    for x in (old, n1, n0))
                a.sd, a.n_sd = sd.ctypes.data, 3
                a.tmin, a.tmax, a.FP, a.FN = P.TMIN, P.TMAX, .01, .2
                a.p, a.q, a.uniform_prior, a.trans_prob = p, q, int(uniform), 0
                if known is not None:
                    a.known_theta = known[0].ctypes.data
                    a.known_prior = known[1].ctypes.data
                a.sd_idx, a.U, a.u = (x.ctypes.data for x in draws)
                a.new_theta, a.A = new.ctypes.data, A.ctypes.data
                a.prior_out = None if uniform else prior.ctypes.data
                a.log_prob, a.declined = lp.ctypes.data, dec.ctypes.data
                a.threads = threads
                a.screen = screen.ctypes.data
                # the densities' sum in index order, made by the team's last
                # rank behind the others (bnpc_mh_args.prior_seq_sum): from
                # the first entry (NaN on entry) or continued from a value
                for start in (np.nan, -12.5):
                    seq = np.array([start])
                    a.prior_seq_sum = seq.ctypes.data
                    status = C.c_int(0)
                    _lib.check(_lib.load().bnpc_mh_batch(C.addressof(table),
                        None, C.byref(a), C.byref(status)), 'mh_batch')
                    assert status.value == 0
                    assert np.array_equal(new, plain[1])
                    assert np.array_equal(dec, plain[3])
                    if not uniform:
                        assert np.array_equal(prior, plain[4])
                        flat = plain[4].ravel()
                        want = flat[0] if np.isnan(start) else start + flat[0]
                        for v in flat[1:]:
                            want = want + v     # one accumulator, index order
                        assert seq[0] == want, (threads, start)
                    else:
                        assert np.isnan(seq[0]) or seq[0] == start
    # a screened batch cannot take draws itself, nor be scored
    a.trans_prob = 0
    st, _ = _lib.rng_export()
    assert _lib.load().bnpc_mh_batch(C.addressof(table), C.byref(st),
        C.byref(a), C.byref(status)) != 0


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_tile_hints_change_nothing(seed, monkeypatch):
    """A first sweep in tiles WITH the tiles' hints (per row the largest entry
    under the priors at issue, its column, the runner-up: the loop decides a
    cell from that record - or from a column born since the tile was issued -
    wherever one cluster dominates) against the same sweep without hints
    (BNPC_SWEEP_HINT=0: every cell walks every live column): the same labels,
    cluster table, parameter rows and stream position, three sweeps in a row;
    and the hints do decide most cells once the true clusters exist."""
    data = synth(seed, 420, 300, 4, 0.1)
    monkeypatch.setenv('BNPC_SWEEP_BYTES', '60000')     # 7 tiles at first
    outs = []
    for hints in ('1', '0'):
        monkeypatch.setenv('BNPC_SWEEP_HINT', hints)
        m = make(P, 'learn', data)
        np.random.seed(10 + seed)
        m.init()
        states = []
        for sweep in range(3):
            m.update_assignments_Gibbs()
            ids = list(m.cells_per_cluster)
            states.append((m.assignment.copy(),
                [(int(a), int(b)) for a, b in m.cells_per_cluster.items()],
                m.parameters[ids].copy()))
            if sweep == 0:
                assert m._ctx.calls['view_set'] >= 5    # it was tiled
        outs.append((states, np.random.random(), getattr(m, '_hint_used', 0),
            m._ctx.calls.get('ll_rows_issue_hint', 0)))
    a, b = outs
    assert a[3] >= 5 and b[3] == 0
    assert a[2] > 0.5 * 420 and b[2] == 0, (a[2], b[2])
    for sa, sb in zip(a[0], b[0]):
        assert np.array_equal(sa[0], sb[0]) and sa[1] == sb[1]
        assert np.array_equal(sa[2], sb[2])
    assert a[1] == b[1]
