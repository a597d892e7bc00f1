"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI
(bnpc_amd._lib -> libbnpc_hip.so), against the CPU oracle and the golden
vectors captured from the reference.

Tolerances (north star: 1e-6 relative on log-likelihoods, identical
assignment trajectories):
  * sums over caller-built tables (bnpc_ll_tables): BIT-EXACT vs the oracle
    (same elements, same m-sequential order);
  * sums over device-built tables (bnpc_ll_theta, bnpc_ll_total): the device
    log differs from NumPy's by <= 1 ulp per element -> 1e-12 relative asserted;
  * column counts: exact integers;
  * trajectories: identical assignments, traces to 1e-9.
"""
import os

import numpy as np
import pytest

from oracle import crp_numpy as O
from bnpc_amd import _lib, model as P
import test_host_logic as H

pytestmark = pytest.mark.gpu


def decode(codes):
    x = codes.astype(np.float64)
    x[codes == 3] = np.nan
    return x


def oracle_ll(data, theta, FP, FN):
    m = O.CRP(data, [-1, -1], [1, 1], FN_error=FN, FP_error=FP)
    return np.stack([m._calc_ll(data[[r]], theta)
        for r in range(data.shape[0])])


def table_sums(x, L1, L0):
    out = np.empty((x.shape[0], L1.shape[0]))
    for k in range(L1.shape[0]):
        el = np.where(x == 1, L1[k], np.where(x == 0, L0[k], np.nan))
        out[:, k] = O.seqsum(el, axis=1)
    return out


def host_tables(theta, FP, FN):
    t64 = theta.astype(np.float64)
    om64 = (1 - theta).astype(np.float64)
    return (np.log(t64 * (1 - FN) + om64 * FP),
        np.log(t64 * FN + om64 * (1 - FP)))


def test_device_is_gfx950():
    assert _lib.device_count() >= 1
    name, cus = _lib.device_info(0)
    assert name.startswith('gfx950'), name
    assert cus >= 200


# ----------------------------------------------------------- golden vectors
def test_ll_theta_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'calc_ll.npz'))
    for ci in range(int(g['n_cases'])):
        data = decode(g[f'c{ci}_data'])
        theta = g[f'c{ci}_theta']
        FP, FN = g[f'c{ci}_FPFN']
        ctx = _lib.Context(data=data)
        got = ctx.ll_theta(0, theta, FP, FN)
        np.testing.assert_allclose(got, g[f'c{ci}_ll'], rtol=1e-12,
            atol=1e-12, err_msg=f'case {ci}')
        # and against the oracle on this stack
        np.testing.assert_allclose(got, oracle_ll(data, theta, FP, FN),
            rtol=1e-12, atol=1e-12)
        # flat total through counts
        assign = g[f'c{ci}_assign']
        ids = np.unique(assign)
        ctx.colcounts_by_label(assign, ids)
        tot = ctx.ll_total(theta[ids], [FP], [FN])[0]
        np.testing.assert_allclose(tot, g[f'c{ci}_flat'], rtol=1e-12)
        ctx.close()


def test_ll_tables_is_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'calc_ll.npz'))
    for ci in range(int(g['n_cases'])):
        data = decode(g[f'c{ci}_data'])
        theta = g[f'c{ci}_theta']
        FP, FN = g[f'c{ci}_FPFN']
        L1, L0 = host_tables(theta, FP, FN)
        ctx = _lib.Context(data=data)
        got = ctx.ll_tables(0, L1, L0)
        assert np.array_equal(got, table_sums(data, L1, L0)), ci
        # with NumPy-built tables the device reproduces the oracle's
        # _calc_ll bit for bit
        assert np.array_equal(got, oracle_ll(data, theta, FP, FN)), ci
        ctx.close()


@pytest.mark.parametrize('kw', ['', '8', '2'])
def test_ll_tables_bits_on_ragged_shapes_and_views(kw, monkeypatch):
    """bnpc_ll_tables on both kernels it can run through (default: the
    producer / consumer pipeline k_ll_seqp; a forced cluster tile BNPC_KW: the
    scalar-load tiling k_ll / k_ll8_asm on re-laid-out tables) at shapes
    around the 64-slot and 64-mutation tile edges, whole matrix and gathered
    views, tables holding -inf (an impossible observation) and +0.0: the
    strict-order sums of the oracle, bit for bit."""
    if kw:
        monkeypatch.setenv('BNPC_KW', kw)
    else:
        monkeypatch.delenv('BNPC_KW', raising=False)
    rng = np.random.RandomState(len(kw) + int(kw or 0) + 3)
    for N, M, K in ((1, 1, 1), (63, 64, 2), (65, 65, 3), (130, 127, 1),
            (517, 1000, 2), (64, 4097, 2), (200, 129, 5)):
        data = (rng.random_sample((N, M)) < 0.35).astype(float)
        data[rng.random_sample(data.shape) < 0.2] = np.nan
        theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
            .astype(np.float32)
        L1, L0 = host_tables(theta, 0.01, 0.2)
        if M > 2:
            L1[0, 1] = -np.inf
            L0[K - 1, M - 1] = 0.0
        ctx = _lib.Context(data=data)
        ctx.reload_options()
        want = table_sums(data, L1, L0)
        assert np.array_equal(ctx.ll_tables(0, L1, L0), want), (N, M, K)
        cells = rng.permutation(N)[:max(1, N // 2)]
        ctx.view_set(1, cells)
        assert np.array_equal(ctx.ll_tables(1, L1, L0), want[cells]), (N, M, K)
        ctx.close()


@pytest.mark.parametrize('kw', ['1', '2', '4', '8'])
def test_every_cluster_tiling_gives_identical_bits(kw, monkeypatch):
    rng = np.random.RandomState(5)
    data = (rng.random_sample((333, 257)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    theta = np.clip(rng.uniform(size=(37, 257)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx = _lib.Context(data=data)
    monkeypatch.delenv('BNPC_KW', raising=False)
    monkeypatch.setenv('BNPC_MSPLIT', '0')      # strict mutation order
    ctx.reload_options()
    ref = ctx.ll_theta(0, theta, 0.01, 0.2)
    monkeypatch.setenv('BNPC_KW', kw)
    ctx.reload_options()
    assert np.array_equal(ctx.ll_theta(0, theta, 0.01, 0.2), ref)
    # small launches split the mutations over waves: partial sums combined in
    # index order - deterministic, equal to ~1 ulp of the sum
    monkeypatch.setenv('BNPC_MSPLIT', '1')
    ctx.reload_options()
    split = ctx.ll_theta(0, theta, 0.01, 0.2)
    np.testing.assert_allclose(split, ref, rtol=1e-14)
    assert np.array_equal(split, ctx.ll_theta(0, theta, 0.01, 0.2))
    # caller-built tables are never split: bit-exact for every tiling
    L1, L0 = host_tables(theta, 0.01, 0.2)
    assert np.array_equal(ctx.ll_tables(0, L1, L0), table_sums(data, L1, L0))
    ctx.close()


# ------------------------------------------------------------- edge cases
@pytest.mark.parametrize('N,M', [(1, 1), (5, 1), (1, 64), (63, 65),
    (64, 64), (65, 63), (130, 1003), (257, 130)])
def test_ragged_shapes(N, M):
    rng = np.random.RandomState(N * 1000 + M)
    data = (rng.random_sample((N, M)) < 0.4).astype(float)
    data[rng.random_sample(data.shape) < 0.25] = np.nan
    theta = np.clip(rng.uniform(size=(3, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx = _lib.Context(data=data)
    L1, L0 = host_tables(theta, 0.02, 0.15)
    assert np.array_equal(ctx.ll_tables(0, L1, L0), table_sums(data, L1, L0))
    np.testing.assert_allclose(ctx.ll_theta(0, theta, 0.02, 0.15),
        table_sums(data, L1, L0), rtol=1e-12, atol=1e-12)
    n1, n0 = ctx.cell_counts()
    assert np.array_equal(n1, (data == 1).sum(axis=1))
    assert np.array_equal(n0, (data == 0).sum(axis=1))
    c1, c0 = ctx.colcounts([np.arange(N)])
    assert np.array_equal(c1[0], (data == 1).sum(axis=0))
    assert np.array_equal(c0[0], (data == 0).sum(axis=0))
    ctx.close()


@pytest.mark.parametrize('seed', range(12))
def test_random_shapes_clusters_and_subsets(seed):
    """Random N, M, K, missing rate and error rates; a random cell subset as a
    view; every launch shape the dispatcher can pick (whole-row, mutation
    split, each cluster tiling)."""
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([1, 3, 64, 65, 200, 513, 900]))
    M = int(rng.choice([1, 7, 64, 129, 300, 1000]))
    K = int(rng.choice([1, 2, 3, 5, 8, 9, 17, 40, 70]))
    data = (rng.random_sample((N, M)) < rng.uniform(.05, .6)).astype(float)
    data[rng.random_sample(data.shape) < rng.uniform(0, .6)] = np.nan
    theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    theta[rng.random_sample(theta.shape) < .1] = np.float32(1e-5)
    theta[rng.random_sample(theta.shape) < .1] = np.float32(1 - 1e-5)
    FP, FN = float(rng.uniform(1e-6, .2)), float(rng.uniform(1e-3, .5))
    ctx = _lib.Context(data=data)
    L1, L0 = host_tables(theta, FP, FN)
    want = table_sums(data, L1, L0)
    assert np.array_equal(ctx.ll_tables(0, L1, L0), want)
    np.testing.assert_allclose(ctx.ll_theta(0, theta, FP, FN), want,
        rtol=1e-12, atol=1e-12)
    cells = rng.randint(0, N, int(rng.randint(1, 2 * N + 2)))
    ctx.view_set(1, cells)
    assert np.array_equal(ctx.ll_tables(1, L1, L0), want[cells])
    np.testing.assert_allclose(ctx.ll_theta(1, theta, FP, FN), want[cells],
        rtol=1e-12, atol=1e-12)
    c1, c0 = ctx.colcounts([cells])
    assert np.array_equal(c1[0], (data[cells] == 1).sum(axis=0))
    assert np.array_equal(c0[0], (data[cells] == 0).sum(axis=0))
    ctx.close()


def test_all_missing_all_ones_all_zeros_and_boundary_theta():
    M = 70
    data = np.zeros((6, M))
    data[0] = np.nan
    data[1] = 1
    data[2] = 0
    data[3, ::2] = 1
    data[4, :] = np.nan
    data[4, 69] = 1
    data[5, :64] = np.nan
    theta = np.stack([np.full(M, 1e-5), np.full(M, 1 - 1e-5),
        np.linspace(1e-5, 1 - 1e-5, M)]).astype(np.float32)
    eps = np.finfo(np.float64).resolution
    ctx = _lib.Context(data=data)
    for FP, FN in ((1e-3, 0.1), (eps, eps), (0.4999, 0.4999)):
        got = ctx.ll_theta(0, theta, FP, FN)
        np.testing.assert_allclose(got, oracle_ll(data, theta, FP, FN),
            rtol=1e-12, atol=1e-12)
        assert np.all(got[0] == 0.0)          # nothing observed -> empty sum
    ctx.close()


def test_invalid_inputs_raise():
    data = np.zeros((4, 4))
    ctx = _lib.Context(data=data)
    theta = np.full((1, 4), 0.5, dtype=np.float32)
    with pytest.raises(RuntimeError, match='error rates'):
        ctx.ll_theta(0, theta, 0.0, 0.1)
    with pytest.raises(RuntimeError, match='out of range'):
        ctx.view_set(1, [0, 7])
    with pytest.raises(RuntimeError, match='view'):
        ctx.view_set(0, [0])
    with pytest.raises(RuntimeError, match='resident counts'):
        ctx.ll_total(theta, [0.1], [0.1])
    bad = np.full((2, 2), 0.5)
    with pytest.raises(RuntimeError, match='not 0, 1 or missing'):
        _lib.Context(data=bad)
    ctx.close()


def test_views_gather_any_cell_list():
    rng = np.random.RandomState(9)
    data = (rng.random_sample((300, 150)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    theta = np.clip(rng.uniform(size=(2, 150)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    L1, L0 = host_tables(theta, 0.01, 0.1)
    ctx = _lib.Context(data=data)
    for cells in (np.array([], dtype=int), np.array([7]), np.arange(64),
            rng.permutation(300), rng.randint(0, 300, 129),
            np.array([5] * 70)):
        ctx.view_set(1, cells)
        assert ctx.view_size(1) == cells.size
        got = ctx.ll_tables(1, L1, L0)
        assert got.shape == (cells.size, 2)
        assert np.array_equal(got, table_sums(data[cells], L1, L0))
    # views are independent
    ctx.view_set(1, np.arange(10))
    ctx.view_set(2, np.arange(290, 300))
    assert np.array_equal(ctx.ll_tables(1, L1, L0),
        table_sums(data[:10], L1, L0))
    assert np.array_equal(ctx.ll_tables(2, L1, L0),
        table_sums(data[290:], L1, L0))
    ctx.close()


def test_colcounts_segments_and_labels():
    rng = np.random.RandomState(10)
    data = (rng.random_sample((1000, 333)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    ctx = _lib.Context(data=data)
    segs = [rng.permutation(1000)[:n] for n in (0, 1, 255, 256, 257, 700)]
    n1, n0 = ctx.colcounts(segs)
    for g, s in enumerate(segs):
        assert np.array_equal(n1[g], (data[s] == 1).sum(axis=0))
        assert np.array_equal(n0[g], (data[s] == 0).sum(axis=0))
    assign = rng.randint(0, 40, 1000) * 3      # sparse ids
    ids = rng.permutation(np.unique(assign))   # any order
    n1, n0 = ctx.colcounts_by_label(assign, ids)
    for g, i in enumerate(ids):
        assert np.array_equal(n1[g], (data[assign == i] == 1).sum(axis=0))
        assert np.array_equal(n0[g], (data[assign == i] == 0).sum(axis=0))
    with pytest.raises(RuntimeError, match='not in ids'):
        ctx.colcounts_by_label(assign, ids[:-1])
    ctx.close()


@pytest.mark.parametrize('N,M', [(1000, 333), (63, 64), (4100, 1003)])
def test_counts_from_lane_masks_equal_the_cell_list_counts(N, M, monkeypatch):
    """Few segments: popcounts over the view's lane masks (k_counts_masks) -
    by label on the identity view, by slot label on a gathered view - give
    the integers of the cell-list kernel and of NumPy; every input / output
    route (in-place pinned memory or copies)."""
    rng = np.random.RandomState(N + M)
    data = (rng.random_sample((N, M)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    ctx = _lib.Context(data=data)
    for zero_copy in ('1', '0'):
        monkeypatch.setenv('BNPC_ZERO_COPY', zero_copy)
        for G in (1, 2, 7, 8, 9, 40, 64):
            assign = rng.randint(0, G, N) * 2
            ids = rng.permutation(np.unique(assign))
            monkeypatch.setenv('BNPC_MASK_COUNTS_MAX', '64')
            ctx.reload_options()
            n1, n0 = ctx.colcounts_by_label(assign, ids)
            monkeypatch.setenv('BNPC_MASK_COUNTS_MAX', '0')
            ctx.reload_options()
            o1, o0 = ctx.colcounts_by_label(assign, ids)
            assert np.array_equal(n1, o1) and np.array_equal(n0, o0)
            g = int(rng.randint(ids.size))
            assert np.array_equal(n1[g],
                (data[assign == ids[g]] == 1).sum(axis=0))
            # the resident copy feeds ll_total: same total either way
            theta = np.clip(rng.uniform(size=(ids.size, M)), 1e-5, 1 - 1e-5) \
                .astype(np.float32)
            a = ctx.ll_total(theta, [.01], [.2])
            monkeypatch.setenv('BNPC_MASK_COUNTS_MAX', '64')
            ctx.reload_options()
            ctx.colcounts_by_label(assign, ids)
            assert np.array_equal(a, ctx.ll_total(theta, [.01], [.2]))
        # a gathered view with repeats, slots of no segment, 3 segments
        cells = rng.randint(0, N, size=min(N, 777))
        ctx.view_set(1, cells)
        labels = rng.randint(-1, 3, size=cells.size)
        n1, n0 = ctx.view_counts(1, labels, 3)
        for g in range(3):
            sub = data[cells[labels == g]]
            assert np.array_equal(n1[g], (sub == 1).sum(axis=0))
            assert np.array_equal(n0[g], (sub == 0).sum(axis=0))
    with pytest.raises(RuntimeError, match='out of range'):
        ctx.view_counts(1, np.full(cells.size, 3), 3)
    ctx.close()


def test_ll_total_trials_match_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, 'state_functions.npz'))
    data = decode(g['data'])
    ctx = _lib.Context(data=data)
    ids = g['le_clusters']
    ctx.colcounts_by_label(g['le_assignment'], ids)
    trials = g['le_trials']
    got = np.concatenate([
        ctx.ll_total(g['le_parameters'], trials[:4, 0], trials[:4, 1]),
        ctx.ll_total(g['le_parameters'], trials[4:, 0], trials[4:, 1])])
    np.testing.assert_allclose(got, g['le_ll'], rtol=1e-12)
    # deterministic: same launch twice gives the same bits
    again = ctx.ll_total(g['le_parameters'], trials[:4, 0], trials[:4, 1])
    assert np.array_equal(again, got[:4])
    # the two halves, with other calls on the context in between (they use
    # the same staging arena and result buffer)
    ctx.ll_total_issue(g['le_parameters'], trials[:4, 0], trials[:4, 1])
    theta = np.clip(np.random.RandomState(3).uniform(size=(3, data.shape[1])),
        1e-5, 1 - 1e-5).astype(np.float32)
    ctx.ll_theta(0, theta, 0.01, 0.2)
    ctx.view_set(1, np.arange(min(40, data.shape[0])))
    ctx.view_counts(1, np.zeros(min(40, data.shape[0]), dtype=np.int64), 1)
    assert np.array_equal(ctx.ll_total_wait(), got[:4])
    with pytest.raises(RuntimeError, match='no deferred total'):
        ctx.ll_total_wait()
    ctx.close()


# --------------------------------------------- model surface on the device
def test_state_functions_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'state_functions.npz'))
    data = decode(g['data'])
    for pi, pb in enumerate(g['param_betas']):
        pre = f'p{pi}_'
        m = P.CRP(data, [-1, -1], list(pb), FN_error=0.1, FP_error=1e-3)
        m.DP_a = float(g[pre + 'DP_a0'])
        m.init_DP_prior()
        m.assignment = g[pre + 'assignment1'].copy()
        cl = g[pre + 'clusters1']
        m.parameters = np.zeros(data.shape, dtype=np.float32)
        m.parameters[cl] = g[pre + 'parameters1']
        m.cells_per_cluster = dict(
            zip(cl.tolist(), g[pre + 'sizes1'].tolist()))
        np.testing.assert_allclose(m.get_lpost_single_new_cluster(),
            g[pre + 'new_cluster'], rtol=1e-12)
        np.testing.assert_allclose(m.get_ll_full(), g[pre + 'll_full1'],
            rtol=1e-12)
        np.testing.assert_allclose(m.get_lprior_full(), g[pre + 'lprior1'],
            rtol=1e-10)
        got = np.stack([m.get_lpost_single(c, cl)
            for c in range(0, data.shape[0], 7)])
        np.testing.assert_allclose(got, g[pre + 'lpost_single'], rtol=1e-12)
        for j in range(int(g[pre + 'n_logA'])):
            cells = g[pre + f'logA{j}_cells']
            new, old = g[pre + f'logA{j}_new'], g[pre + f'logA{j}_old']
            std = g[pre + f'logA{j}_std']
            a, b = (P.TMIN - old) / std, (P.TMAX - old) / std
            for clip in (0, 1):
                A = m._get_log_A(new, old, cells, a, b, std, bool(clip))
                np.testing.assert_allclose(A, g[pre + f'logA{j}_A{clip}'],
                    rtol=1e-9, atol=1e-9)
        cells = g[pre + 'rg_cells']
        m._rg_open(cells)
        m.rg_assignment = g[pre + 'rg_assignment_init'].astype(np.int64)
        m.rg_params_split = g[pre + 'rg_params_split'].copy()
        m.rg_params_merge = g[pre + 'rg_params_merge'].copy()
        np.testing.assert_allclose(
            m._rg_get_ll(cells[1:-1], m.rg_params_split), g[pre + 'rg_ll'],
            rtol=1e-12)
        np.testing.assert_allclose(m._get_ll_ratio(cells, 'split'),
            g[pre + 'rg_ll_ratio_split'], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(m._get_ll_ratio(cells, 'merge'),
            g[pre + 'rg_ll_ratio_merge'], rtol=1e-9, atol=1e-9)
        m.close()


@pytest.mark.parametrize('kind', ['fixed', 'learn'])
def test_gibbs_sweeps_match_oracle(kind):
    H.test_gibbs_sweeps_match_oracle(kind)


@pytest.mark.parametrize('ahead', [2, 1])
def test_tiled_sweep_on_device(ahead, monkeypatch):
    """Tiny tiles (dozens per sweep) against the oracle, with two tiles in
    flight while the host walks one (the default) and with one.  (The budget
    is also the parameter batch's: its screens run in slices of 8 rows here.)"""
    monkeypatch.setenv('BNPC_SWEEP_BYTES', '30000')
    monkeypatch.setattr(P, '_TILES_AHEAD', ahead)
    H.test_gibbs_sweeps_match_oracle('learn')
    H.test_gibbs_opens_many_clusters_in_one_sweep()
    data = H.synth(0, 1000, 200, 10, 0.10)
    ro = H.run_chain(H.make(O, 'fixed', data), 9, 42)
    rp = H.run_chain(H.make(P, 'fixed', data), 9, 42)
    assert np.array_equal(ro['assignments'], rp['assignments'])


def test_update_parameters_and_errors_match_oracle():
    H.test_update_parameters_and_errors_match_oracle()


@pytest.mark.parametrize('pb', [(.25, .25), (1, 1)])
@pytest.mark.parametrize('start', ['together', 'fragmented', 'gibbs'])
def test_split_merge_moves_match_oracle(pb, start):
    H.test_split_merge_moves_match_oracle(pb, start)


def test_rg_init_split_is_bit_identical_to_oracle():
    H.test_rg_init_split_is_bit_identical_to_oracle()


@pytest.mark.parametrize('kind,eup', [('fixed', 0.), ('learn', .25)])
def test_chain_matches_oracle(kind, eup):
    H.test_chain_matches_oracle(kind, eup)


def test_chain_matches_oracle_on_the_public_scipy_path(monkeypatch):
    """The fallbacks behind the start-up self-checks - no native parameter
    batch, no direct calls of SciPy's private kernels, the documented
    scipy.stats API only - walk the oracle's trajectory on the device too
    (that path had CPU coverage only)."""
    from bnpc_amd import fastdist as F
    monkeypatch.setenv('BNPC_NATIVE_MH', '0')
    P._NATIVE.clear()
    for key, val in (('checked', True), ('fast', False), ('shared', False),
            ('left', False)):
        monkeypatch.setitem(F._state, key, val)
    try:
        assert P._native_kernels() is None
        H.test_chain_matches_oracle('learn', .25)
        H.test_split_merge_moves_match_oracle((.25, .25), 'together')
    finally:
        P._NATIVE.clear()


def test_other_init_modes_match_oracle():
    H.test_other_init_modes_match_oracle()


def test_example_data_trajectory_matches_reference(golden_dir):
    """example_data/data.csv, seed 42, fixed error rates, 200 steps: the
    device path walks the REFERENCE's assignment trajectory
    (sha ec91db5ad46ef182)."""
    from bnpc_amd.io import load_data
    t = np.load(os.path.join(golden_dir, 'trajectories.npz'))
    data = load_data(os.path.join(golden_dir, 'example_data.csv'))
    res = H.run_chain(H.make(P, 'fixed', data), 200, 42)
    assert np.array_equal(res['assignments'], t['ex_fixed_assignments'])
    np.testing.assert_allclose(res['ML'], t['ex_fixed_ML'], rtol=1e-9)
    np.testing.assert_allclose(res['MAP'], t['ex_fixed_MAP'], rtol=1e-9)


def test_example_data_learned_errors_prefix_matches_reference(golden_dir):
    """Config 1 as BASELINE.json states it (the CLI without -FP/-FN is the
    learned-error model, libs/CRP_learning_errors.py:52-111): example_data,
    seed 42.  The DEVICE chain walks the reference's own trajectory for the
    first 40 recorded states (beyond that the golden SciPy 1.7.1 and this
    stack's scalar truncnorm differ by an ulp, SURVEY.md Appendix A - the same
    prefix tests/test_oracle_golden.py pins for the oracle), ML / FN / FP to
    1e-8; and the 60 x 40 split-merge-heavy learned-error chain for 50."""
    from bnpc_amd.io import load_data
    t = np.load(os.path.join(golden_dir, 'trajectories.npz'))
    data = load_data(os.path.join(golden_dir, 'example_data.csv'))
    res = H.run_chain(H.make(P, 'learn', data), 200, 42, eup=.25)
    n = 40
    assert np.array_equal(res['assignments'][:n],
        t['ex_learn_assignments'][:n])
    for key in ('ML', 'FN', 'FP', 'DP_alpha'):
        np.testing.assert_allclose(res[key][:n], t[f'ex_learn_{key}'][:n],
            rtol=1e-8, err_msg=key)
    np.testing.assert_allclose(res['MAP'][:n], t['ex_learn_MAP'][:n],
        rtol=1e-8)
    small = decode(t['small_data'])
    res = H.run_chain(H.make(P, 'learn', small), 120, 7, sm_prob=.5,
        sm_steps=2, eup=.25)
    n = 50
    assert np.array_equal(res['assignments'][:n],
        t['sm_learn_assignments'][:n])
    for key in ('ML', 'FN', 'FP'):
        np.testing.assert_allclose(res[key][:n], t[f'sm_learn_{key}'][:n],
            rtol=1e-8, err_msg=key)


def test_eight_chains_in_workers_equal_in_process_chains():
    """Configs 4 / 5 are `-n 8`: eight forked workers, eight device contexts
    at once (libs/MCMC.py:100-135).  Config 4's generator at 2000 x 400, 30
    steps through MCMC.run with the pool: every worker's results are
    bit-equal to the chain of the same seed run alone in a process of its own,
    every worker runs on the device ordinal device_for_chain gives it, and no
    /dev/shm block is left behind (tests/eight_chains_check.py, in a fresh
    interpreter: this one has touched the GPU and would spawn, not fork)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable,
        os.path.join(root, 'tests', 'eight_chains_check.py'), '8', '2000',
        '400', '30'], capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert 'EIGHT CHAINS OK' in res.stdout, res.stdout[-3000:]


def test_config2_chain_matches_oracle():
    """BASELINE config 2 (1000 x 200, 10 % missing, fixed FP/FN), 12 steps
    including the first sweep from K0 ~ 630 clusters."""
    data = H.synth(0, 1000, 200, 10, 0.10)
    ro = H.run_chain(H.make(O, 'fixed', data), 12, 42)
    rp = H.run_chain(H.make(P, 'fixed', data), 12, 42)
    assert np.array_equal(ro['assignments'], rp['assignments'])
    np.testing.assert_allclose(rp['ML'], ro['ML'], rtol=1e-9)


@pytest.mark.parametrize('cfg', ['c3', 'c4', 'c3k'])
def test_full_size_trajectory_matches_oracle_fixture(cfg, golden_dir):
    """BASELINE configs 3 and 4 at FULL size (5000 x 1000 and 10000 x 2000,
    20 % missing, learned errors, CLI-default moves), 9 steps including the
    first sweep from K0 = 3152 / 6325 clusters (config 4: after four
    split/merge attempts at K0, each with a K0 x M proposal batch): the device
    chain walks the trajectory the CPU oracle produced
    (tests/golden/make_c3_trajectory.py; 3 and 24 CPU-minutes).  `c3k`
    (more than 64 clusters at bench size under the driver's
    eyes): config 3's shape with 200 true clusters - the chain runs at K ~ 200
    from its second step on."""
    import bench
    import libs.CRP_learning_errors as dev
    from bnpc_amd.mcmc import MCMC
    import contextlib
    import io
    t = np.load(os.path.join(golden_dir, f'{cfg}_trajectory.npz'))
    N, M, C, miss, learned = bench.CONFIGS[cfg]
    data = bench.synth(0, N, M, C, miss)
    model = bench.make_model(None, dev, data, learned)
    mcmc = MCMC(model, error_prob=.25, **bench.MCMC_PARAMS)
    with contextlib.redirect_stdout(io.StringIO()):
        mcmc.run((9, 3), 42, 1, 0, '', True)
    res = mcmc.get_results()[0]
    assert np.array_equal(res['assignments'], t['assignments'])
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        np.testing.assert_allclose(res[key], t[key], rtol=1e-9, err_msg=key)


def test_k150_chain_matches_oracle_fixture(golden_dir):
    """a chain with MORE THAN 64 CLUSTERS against the
    oracle in the driver-run suite - bench.py's `k150` workload (2000 x 500,
    150 true clusters), 20 steps from the initial state (K0 = 1268) stepped
    as bench.py steps its chain; the oracle's walk is the fixture
    (tests/golden/make_k150_trajectory.py).  Identical assignments after
    every step, traces to 1e-9, every step ONE native call, and from the
    second sweep on nearly every cell decided from its device record."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_k150',
        os.path.join(golden_dir, 'make_k150_trajectory.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    import libs.CRP as dev_fixed
    import libs.CRP_learning_errors as dev_learn
    t = np.load(os.path.join(golden_dir, 'k150_trajectory.npz'))
    model, res = mk.walk(dev_fixed, dev_learn)
    assert np.array_equal(res['assignments'], t['assignments'])
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        np.testing.assert_allclose(res[key], t[key], rtol=1e-9, err_msg=key)
    K = [len(np.unique(a)) for a in res['assignments']]
    assert K[0] > 1000 and min(K[1:]) > 64, K
    stats = model.host_stats()
    assert stats['native_steps'] == mk.STEPS, stats
    # the first sweep (K0 random clusters: 2000 of the cells swept) scans many
    # of its cells; of all the others more than 95 % come from their records
    N = res['assignments'].shape[1]
    assert stats['swept'] >= 5 * N, stats
    later = stats['swept'] - N
    assert stats['hint_used'] - N <= later
    assert stats['hint_used'] > 0.95 * later, stats
    model.close()


# ------------------------------------------ full BASELINE size, properties
def test_full_size_properties_5000x1000():
    """Config 3 shape (5000 x 1000, 20 % missing) with K0 = 3152 clusters:
    size-independent properties + a sampled comparison with the oracle."""
    N, M, K = 5000, 1000, 3152
    data = H.synth(0, N, M, 10, 0.20)
    rng = np.random.RandomState(1)
    theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    FP, FN = 0.01, 0.2
    ctx = _lib.Context(data=data)
    ll = ctx.ll_theta(0, theta, FP, FN)
    assert ll.shape == (N, K) and np.all(np.isfinite(ll)) and np.all(ll < 0)

    # (a) sampled oracle comparison
    rows = rng.choice(N, 24, replace=False)
    cols = rng.choice(K, 48, replace=False)
    want = oracle_ll(data[rows], theta[cols], FP, FN)
    np.testing.assert_allclose(ll[np.ix_(rows, cols)], want, rtol=1e-12)

    # (b) splitting the clusters over two launches changes nothing
    left = ctx.ll_theta(0, theta[:1000], FP, FN)
    assert np.array_equal(left, ll[:, :1000])

    # (c) permutation equivariance through a gathered view
    perm = rng.permutation(N)[:1500]
    ctx.view_set(1, perm)
    sub = ctx.ll_theta(1, theta[:64], FP, FN)
    np.testing.assert_allclose(sub, ll[perm, :64], rtol=1e-14)

    # (d) the flat total equals the sum of the assigned entries
    assign = rng.randint(0, K, N)
    ids = np.unique(assign)
    ctx.colcounts_by_label(assign, ids)
    tot = ctx.ll_total(theta[ids], [FP], [FN])[0]
    np.testing.assert_allclose(tot, ll[np.arange(N), assign].sum(),
        rtol=1e-11)

    # (e) counts: clusters partition the cells; 1s + 0s + missing = N
    n1, n0 = ctx.colcounts_by_label(assign, ids)
    a1, a0 = ctx.colcounts([np.arange(N)])
    assert np.array_equal(n1.sum(axis=0), a1[0])
    assert np.array_equal(n0.sum(axis=0), a0[0])
    assert np.array_equal(a1[0] + a0[0] + np.isnan(data).sum(axis=0),
        np.full(M, N))
    ctx.close()


# ------------------------------------------------------------------- the CLI
def test_cli_end_to_end_in_process(golden_dir, tmp_path):
    """`run_BnpC.py example_data/data.csv -FP .001 -FN .1 -s 200 --seed 42`
    (chain in the main process): the reference's trajectory, its outputs."""
    import run_BnpC
    t = np.load(os.path.join(golden_dir, 'trajectories.npz'))
    args = run_BnpC.parse_args([os.path.join(golden_dir, 'example_data.csv'),
        '-FP', '0.001', '-FN', '0.1', '-n', '1', '-s', '200', '--seed', '42',
        '-np', '-e', 'ML', 'MAP', '-o', str(tmp_path), '-v', '0', '--debug'])
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        results = run_BnpC.main(args)
    assert np.array_equal(results[0]['assignments'], t['ex_fixed_assignments'])
    for name in ('assignment.txt', 'errors.txt', 'args.txt'):
        assert os.path.getsize(os.path.join(str(tmp_path), name)) > 0
    from bnpc_amd.io import load_txt
    assign = load_txt(os.path.join(str(tmp_path), 'assignment.txt'))
    assert len(assign) == 100 and len(set(assign)) == 5


def test_cli_two_chains_in_pool_workers(golden_dir, tmp_path):
    """-n 2: each chain in its own forked worker with its own device context
    (the parent never touches the GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'run_BnpC.py'),
        os.path.join(golden_dir, 'example_data.csv'), '-n', '2', '-s', '40',
        '--seed', '7', '-np', '-e', 'MAP', '-sc', '-o', str(tmp_path)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
        cwd=root)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'secs. per MCMC step' in res.stdout
    with open(os.path.join(str(tmp_path), 'assignment.txt')) as f:
        assert len(f.read().strip().splitlines()) == 3


def test_linearity_and_erasure_properties():
    """The device sum is linear in the tables, and erasing observations
    (0/1 -> missing) removes exactly their terms."""
    rng = np.random.RandomState(21)
    N, M, K = 700, 333, 5
    data = (rng.random_sample((N, M)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.1] = np.nan
    A1, A0 = -rng.random_sample((K, M)) * 5, -rng.random_sample((K, M))
    B1, B0 = -rng.random_sample((K, M)) * 2, -rng.random_sample((K, M)) * 7
    ctx = _lib.Context(data=data)
    a = ctx.ll_tables(0, A1, A0)
    b = ctx.ll_tables(0, B1, B0)
    ab = ctx.ll_tables(0, A1 + B1, A0 + B0)
    np.testing.assert_allclose(ab, a + b, rtol=1e-12)
    # scaling by a power of two is exact in floating point
    assert np.array_equal(ctx.ll_tables(0, 4 * A1, 4 * A0), 4 * a)
    # erase 15 % of the observations
    erase = (rng.random_sample(data.shape) < 0.15) & ~np.isnan(data)
    erased = data.copy()
    erased[erase] = np.nan
    ctx2 = _lib.Context(data=erased)
    a2 = ctx2.ll_tables(0, A1, A0)
    for k in range(K):
        gone = np.where(erase & (data == 1), A1[k], 0.0) \
            + np.where(erase & (data == 0), A0[k], 0.0)
        np.testing.assert_allclose(a[:, k] - a2[:, k], gone.sum(axis=1),
            rtol=1e-9, atol=1e-9)
    # erasing everything leaves empty sums
    ctx3 = _lib.Context(data=np.full((N, M), np.nan))
    assert not ctx3.ll_tables(0, A1, A0).any()
    n1, n0 = ctx3.colcounts([np.arange(N)])
    assert not n1.any() and not n0.any()
    for c in (ctx, ctx2, ctx3):
        c.close()


def test_full_size_properties_50000x5000():
    """Config 5 shape (50000 x 5000, 20 % missing): the largest BASELINE
    size, K = 512 clusters; sampled oracle rows + size-independent
    properties."""
    N, M, K = 50000, 5000, 512
    data = H.synth(0, N, M, 50, 0.20)
    rng = np.random.RandomState(2)
    theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    FP, FN = 0.01, 0.2
    ctx = _lib.Context(data=data)
    ll = ctx.ll_theta(0, theta, FP, FN)
    assert ll.shape == (N, K) and np.all(np.isfinite(ll))
    rows = rng.choice(N, 6, replace=False)
    cols = rng.choice(K, 16, replace=False)
    np.testing.assert_allclose(ll[np.ix_(rows, cols)],
        oracle_ll(data[rows], theta[cols], FP, FN), rtol=1e-12)
    n1c, n0c = ctx.cell_counts()
    assert np.array_equal(n1c[rows], (data[rows] == 1).sum(axis=1))
    # a gathered tile of cells in any order gives the same rows
    tile = rng.permutation(N)[:3000]
    ctx.view_set(1, tile)
    np.testing.assert_allclose(ctx.ll_theta(1, theta[:96], FP, FN),
        ll[tile, :96], rtol=1e-14)
    # flat total == sum of the assigned entries; counts partition the cells
    assign = rng.randint(0, K, N)
    n1, n0 = ctx.colcounts_by_label(assign, np.arange(K))
    tot = ctx.ll_total(theta, [FP], [FN])[0]
    np.testing.assert_allclose(tot, ll[np.arange(N), assign].sum(), rtol=1e-11)
    a1, a0 = ctx.colcounts([np.arange(N)])
    assert np.array_equal(n1.sum(axis=0), a1[0])
    assert np.array_equal(a1[0] + a0[0] + np.isnan(data).sum(axis=0),
        np.full(M, N))
    ctx.close()


def test_config5_moves_chain_matches_oracle():
    """Config 5's move schedule (-smp 0.5 -sms 5, learned error rates) at a
    size the CPU oracle walks in seconds (2000 x 500, 20 % missing): 40 steps
    from the random start - ~20 split/merge moves with 5 restricted scans
    each - with identical assignments and traces to 1e-9."""
    data = H.synth(5, 2000, 500, 12, 0.20)
    kw = dict(sm_prob=.5, sm_steps=5, eup=.25)
    ro = H.run_chain(H.make(O, 'learn', data), 40, 42, **kw)
    rp = H.run_chain(H.make(P, 'learn', data), 40, 42, **kw)
    assert np.array_equal(ro['assignments'], rp['assignments'])
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        np.testing.assert_allclose(rp[key], ro[key], rtol=1e-9)
    assert np.array_equal(ro['params'], rp['params'])


def test_config5_steady_state_at_full_size_matches_oracle_fixture(golden_dir):
    """Config 5 at FULL size (50000 x 5000, 20 % missing, learned errors,
    -smp 0.5 -sms 5) in its steady state, step for step against the CPU
    oracle (tests/golden/c5_schedule.py + make_c5_trajectory.py: 38 CPU
    minutes): init(assign=near-truth labels), 6 scheduled steps (Gibbs sweeps
    over 50 000 cells x ~50 clusters, parameter batches of 50 x 5000 entries
    through the device screen, error-rate updates), a forced split and a
    forced merge - restricted scans on views of 955 - 1972 cells x 5000
    mutations, `k_counts_masks` with 50 segments, the 4-chunk split of
    `k_ll8_asm`.  Identical assignments, restricted-Gibbs log (move, cells,
    accepted), parameter rows (digests + a sample), stream position; traces
    to 1e-9."""
    import sys
    sys.path.insert(0, golden_dir)
    import bench
    import c5_schedule as S
    import libs.CRP_learning_errors as dev
    t = np.load(os.path.join(golden_dir, 'c5_trajectory.npz'))
    N, M, C_, miss, learned = bench.CONFIGS['c5']
    data = H.synth(0, N, M, C_, miss)
    res = S.drive(dev, data)
    assert np.array_equal(res['moves'], t['moves'])
    assert (res['moves'][:, 1] >= 900).all() and set(res['moves'][:, 0]) \
        == {0, 1}
    assert np.array_equal(res['assignments'], t['assignments'])
    assert list(res['K']) == list(t['K'])
    for key in ('ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        np.testing.assert_allclose(res[key], t[key], rtol=1e-9, err_msg=key)
    assert list(res['digest']) == list(t['digest'])
    assert np.array_equal(
        res['last_params'].ravel()[t['last_params_index']],
        t['last_params_sample'])
    assert res['stream_check'] == float(t['stream_check'])


def test_first_sweep_at_config5_size_does_not_depend_on_the_tiling(
        monkeypatch, golden_dir):
    """The first sweep of config 5 at FULL size (50000 x 5000 from K0 ~ 31600
    clusters: a 12.6 GB matrix, never materialised) with the default 256 MiB
    tiles (36 tiles of 1024+ cells, side-lane columns for clusters reborn in
    between), with 64 MiB tiles (115 tiles of ~265 cells) and with 2 GiB
    tiles (6 tiles) - other tile boundaries, other launch shapes: same assignment, same clusters in the
    same order, same parameter rows, same position of the random stream.
    Oracle identity of the tiled sweep is shown at forced-small tiles
    (test_tiled_sweep_on_device) and, at THIS size, on the sweep's first 96
    cells: the CPU oracle walked them from the same seed
    (tests/golden/make_c5_first_cells.py: the cluster every one of them drew -
    two thirds open a cluster of their own, whose profile is drawn from the
    stream - among 31 644 clusters), and a cell is visited once per sweep, so
    the labels the device's whole sweep leaves them with must be those -
    tiles of k_ll8_lds sums, wide hints, births, stream position included."""
    import bench
    import libs.CRP_learning_errors as dev
    N, M, C_, miss, learned = bench.CONFIGS['c5']
    data = H.synth(0, N, M, C_, miss)
    outs = []
    for tile_bytes in (None, 64 << 20, 2 << 30):
        if tile_bytes is None:
            monkeypatch.delenv('BNPC_TILE_BYTES', raising=False)
            monkeypatch.delenv('BNPC_SWEEP_BYTES', raising=False)
        else:
            monkeypatch.setenv('BNPC_TILE_BYTES', str(tile_bytes))
            monkeypatch.setenv('BNPC_SWEEP_BYTES', str(tile_bytes))
        np.random.seed(42)
        m = bench.make_model(dev, dev, data, learned)
        m.init()
        K0 = len(m.cells_per_cluster)
        assert 8 * N * K0 > (8 << 30)          # far beyond either budget
        m.update_assignments_Gibbs()
        ids = list(m.cells_per_cluster)
        outs.append((m.assignment.copy(), list(m.cells_per_cluster.items()),
            m.parameters[ids].copy(), np.random.random()))
        m.close()
    a = outs[0]
    for b in outs[1:]:
        assert np.array_equal(a[0], b[0])
        assert a[1] == b[1] and a[3] == b[3]
        assert np.array_equal(a[2], b[2])
    assert len(a[1]) < 200                      # the sweep collapsed K0
    first = np.load(os.path.join(golden_dir, 'c5_first_cells.npz'))
    assert int(first['seed']) == 42 and int(first['K0']) == K0
    assert first['cells'].size >= 64 and first['born'].sum() >= 8
    assert np.array_equal(a[0][first['cells']], first['drawn'])
    # ... and the rest of that fixture: the device sweep STOPPED
    # after those cells stands where the oracle stood - as many clusters, as
    # many of them opened by these cells, the same next uniform on the stream
    n_first = int(first['cells'].size)
    monkeypatch.delenv('BNPC_TILE_BYTES', raising=False)
    monkeypatch.delenv('BNPC_SWEEP_BYTES', raising=False)
    np.random.seed(42)
    m = bench.make_model(dev, dev, data, learned)
    m.init()
    m._sweep_stop = n_first
    m.update_assignments_Gibbs()
    peek = np.random.random()
    assert np.array_equal(m.assignment[first['cells']], first['drawn'])
    assert len(m.cells_per_cluster) == int(first['K_after'][-1])
    assert sum(m.cells_per_cluster.values()) == N
    untouched = np.setdiff1d(np.arange(N), first['cells'])
    np.random.seed(42)
    m0 = bench.make_model(dev, dev, data, learned)
    m0.init()
    assert np.array_equal(m.assignment[untouched], m0.assignment[untouched])
    m0.close()
    assert peek == float(first['peek'])
    m.close()


def test_pinned_result_buffer():
    rng = np.random.RandomState(4)
    data = (rng.random_sample((500, 90)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    theta = np.clip(rng.uniform(size=(6, 90)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx = _lib.Context(data=data)
    want = ctx.ll_theta(0, theta, 0.01, 0.2)
    view = ctx.ll_theta_pinned(0, theta, 0.01, 0.2, 6 + 16)
    assert view.shape == (500, 22)
    assert np.array_equal(view[:, :6], want)
    view[:, 6] = 1.5                      # writable in place (new columns)
    assert np.all(view[:, 6] == 1.5)
    # the flat total uses its own staging: the pinned matrix stays intact
    ctx.colcounts_by_label(rng.randint(0, 6, 500), np.arange(6))
    ctx.ll_total(theta, [0.01], [0.2])
    assert np.array_equal(view[:, :6], want)
    ctx.close()


def test_resident_parameter_rows():
    """bnpc_theta_put + bnpc_ll_rows_pinned == bnpc_ll_theta on the gathered
    rows (bit for bit), also after rows were overwritten / the store grew."""
    rng = np.random.RandomState(6)
    data = (rng.random_sample((400, 130)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    params = np.clip(rng.uniform(size=(50, 130)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx = _lib.Context(data=data)
    ctx.theta_put(0, params[:30])
    rows = np.array([3, 29, 0, 17, 8])
    want = ctx.ll_theta(0, params[rows], 0.01, 0.2)
    got = ctx.ll_rows_pinned(0, rows, 0.01, 0.2, rows.size + 16)
    assert np.array_equal(got[:, :rows.size], want)
    params[17] = np.float32(0.25)
    ctx.theta_put(17, params[17])                 # overwrite one row
    ctx.theta_put(45, params[45:50])              # grow past the end
    rows = np.array([17, 49, 45, 3])
    want = ctx.ll_theta(0, params[rows], 0.01, 0.2)
    got = ctx.ll_rows_pinned(0, rows, 0.01, 0.2, 4)
    assert np.array_equal(got, want)
    with pytest.raises(RuntimeError, match='resident parameter store'):
        ctx.ll_rows_pinned(0, np.array([50]), 0.01, 0.2, 1)
    ctx.close()


def test_issued_tiles_overlap_other_calls():
    """bnpc_ll_rows_issue / bnpc_ll_rows_wait: three tiles in flight on their
    own pinned buffers (two device buffers behind them, copies on their own
    stream) give the bits of the synchronous call; calls made in
    between (a column for a freshly opened cluster, a parameter row rewritten
    - they run on the side lane) neither disturb them nor wait for them."""
    rng = np.random.RandomState(7)
    data = (rng.random_sample((900, 200)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    params = np.clip(rng.uniform(size=(300, 200)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    ctx = _lib.Context(data=data)
    ctx.theta_put(0, params)
    # (row 299 is rewritten below: no tile reads it)
    rows_a = rng.permutation(299)[:250]
    rows_b = rng.permutation(299)[:120]
    cells_a, cells_b = np.arange(0, 500), np.arange(400, 900)
    ctx.view_set(3, cells_a)
    ctx.view_set(4, cells_b)
    want_a = ctx.ll_rows_pinned(3, rows_a, 0.01, 0.2, rows_a.size).copy()
    want_b = ctx.ll_rows_pinned(4, rows_b, 0.02, 0.1, rows_b.size + 5).copy()

    rows_c = rng.permutation(299)[:77]
    cells_c = rng.permutation(900)[:333]
    ctx.view_set(5, cells_c)
    want_c = ctx.ll_rows_pinned(5, rows_c, 0.01, 0.2, rows_c.size).copy()

    ctx.ll_rows_issue(3, rows_a, 0.01, 0.2, rows_a.size, 0)
    ctx.ll_rows_issue(4, rows_b, 0.02, 0.1, rows_b.size + 5, 1)
    ctx.ll_rows_issue(5, rows_c, 0.01, 0.2, rows_c.size, 2)
    with pytest.raises(RuntimeError, match='unconsumed tile'):
        ctx.ll_rows_issue(3, rows_a, 0.01, 0.2, rows_a.size, 0)
    # side-lane work while both tiles are in flight
    fresh = np.clip(rng.uniform(size=(1, 200)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    col = ctx.ll_theta(3, fresh, 0.01, 0.2)
    ctx.theta_put(299, fresh[0])
    got_b = ctx.ll_rows_wait(1, cells_b.size, rows_b.size + 5)
    got_a = ctx.ll_rows_wait(0, cells_a.size, rows_a.size).copy()
    # a fourth issue re-uses the device buffer of the second while the third
    # may still be copying
    ctx.ll_rows_issue(3, rows_a, 0.01, 0.2, rows_a.size, 0)
    got_c = ctx.ll_rows_wait(2, cells_c.size, rows_c.size)
    assert np.array_equal(got_c, want_c)
    assert np.array_equal(got_a, want_a)
    got_a = ctx.ll_rows_wait(0, cells_a.size, rows_a.size)
    assert np.array_equal(got_a, want_a)
    assert np.array_equal(got_b[:, :rows_b.size], want_b[:, :rows_b.size])
    assert np.array_equal(col[:, 0],
        ctx.ll_theta(0, fresh, 0.01, 0.2)[cells_a, 0])
    with pytest.raises(RuntimeError, match='no tile was issued'):
        ctx.ll_rows_wait(0, cells_a.size, rows_a.size)
    with pytest.raises(RuntimeError, match='slot'):
        ctx.ll_rows_issue(3, rows_a, 0.01, 0.2, rows_a.size, _lib.TILE_SLOTS)
    ctx.close()


def test_context_from_bit_planes_equals_context_from_matrix(tmp_path):
    """bnpc_create_planes (memory-mapped bit-plane file straight to HBM)
    against bnpc_create (float64 matrix): same per-cell counts, same sums bit
    for bit; inconsistent planes are refused."""
    from bnpc_amd import bitplanes as B
    rng = np.random.RandomState(8)
    N, M = 333, 257
    data = (rng.random_sample((N, M)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    path = str(tmp_path / ('x' + B.SUFFIX))
    B.BitPlanes.from_data(data).save(path)
    planes = B.load(path)
    assert isinstance(planes.planes, np.memmap)
    theta = np.clip(rng.uniform(size=(9, M)), 1e-5, 1 - 1e-5) \
        .astype(np.float32)
    a = _lib.Context(data=data)
    b = _lib.Context(data=planes)
    assert all(np.array_equal(x, y)
        for x, y in zip(a.cell_counts(), b.cell_counts()))
    assert np.array_equal(a.ll_theta(0, theta, .01, .2),
        b.ll_theta(0, theta, .01, .2))
    L1, L0 = host_tables(theta[:2], .01, .2)
    assert np.array_equal(b.ll_tables(0, L1, L0), table_sums(data, L1, L0))
    a.close()
    b.close()
    bad = np.array(planes.planes)
    bad[0, 0, 1] |= bad[0, 0, 0] | 1       # a bit set in both planes
    bad[0, 0, 0] |= 1
    with pytest.raises(RuntimeError, match='inconsistent'):
        _lib.Context(data=B.BitPlanes(bad, M))


def test_sweep_hint_from_the_device_is_the_rows_top_two():
    """bnpc_ll_theta_pinned_top2: per cell the two largest entries of
    ll + prior and the FIRST column of the largest, computed from the very
    matrix that is returned (ties included)."""
    rng = np.random.RandomState(12)
    N, M = 700, 130
    data = (rng.random_sample((N, M)) < 0.3).astype(float)
    data[rng.random_sample(data.shape) < 0.2] = np.nan
    ctx = _lib.Context(data=data)
    # (up to 64 columns: one thread per row, priors as kernel arguments; more:
    # one wave per row, priors in device memory; rows of more than 1024
    # columns are not written through)
    for K in (1, 2, 13, 64, 65, 130, 257, 1100, 2500):
        theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5) \
            .astype(np.float32)
        if K > 100:             # near-duplicates: rows torn between columns
            theta[70:90] = theta[5]
            theta[70:90, :3] = np.clip(theta[70:90, :3] + .01, 1e-5, 1 - 1e-5)
        if K > 2:
            theta[2] = theta[0]             # an exact tie between columns
        prior = -rng.uniform(0, 9, size=K)
        if K > 2:
            prior[2] = prior[0]
        ll, hint = ctx.ll_theta_pinned_top2(0, theta, .01, .2, K + 3, prior)
        # rows with a fourth entry within reach of the runner-up (the ones a
        # sweep scans) are in the host matrix already, with the hints
        here = hint['row_here'] == 1
        assert np.array_equal(here, (hint['fourth'] > hint['second'] - 72.0)
            & (hint['second'] > hint['best'] - 48.0) & (K <= 1024))
        assert here.any() or K <= 3 or K > 1024
        early = ll[here, :K].copy()
        ctx.matrix_wait()       # the matrix is copied behind the hints
        assert np.array_equal(early, ll[here, :K])
        assert not here.any() or K > 3
        live_ll, live_hint = ll, hint
        ll, hint = ll.copy(), hint.copy()
        assert np.array_equal(ll[:, :K], ctx.ll_theta(0, theta, .01, .2))
        # both views survive the calls a sweep makes while it uses them (a
        # column for a cluster opened half-way, counts, totals)
        ctx.ll_theta(0, theta[:1], .02, .3)
        ctx.colcounts_by_label(rng.randint(0, 3, N), np.arange(3))
        assert np.array_equal(live_ll, ll, equal_nan=True)
        assert np.array_equal(live_hint, hint)
        post = ll[:, :K] + prior[None, :]
        col = np.argmax(post, axis=1)       # first maximum
        assert np.array_equal(hint['col'], col)
        assert np.array_equal(hint['best'], post[np.arange(N), col])
        rest = post.copy()
        rest[np.arange(N), col] = -np.inf
        assert np.array_equal(hint['second'], rest.max(axis=1))
        # every field, against the record built from the matrix on the host
        # (four largest entries, columns and log-likelihoods of three)
        want = _lib.hints_from_matrix(ll[:, :K], prior)
        for name in ('best', 'second', 'third', 'fourth', 'col', 'col2',
                'col3'):
            assert np.array_equal(hint[name], want[name]), (K, name)
        for lik, col in (('ll_best', 'col'), ('ll_second', 'col2'),
                ('ll_third', 'col3')):
            there = want[col] >= 0
            assert np.array_equal(hint[lik][there], want[lik][there]), (K, lik)
        # the weights of the second / third column relative to the first
        # (float32 of the device's exp(): the loop allows 2.5e-7, a quarter of
        # its band of 1e-6)
        for e, col in (('e2', 'col2'), ('e3', 'col3')):
            there = want[col] >= 0
            assert np.all(hint[e][~there] == 0), (K, e)
            assert np.allclose(hint[e][there], want[e][there], rtol=2.5e-7,
                atol=1e-37), (K, e)      # (float32 denormals: absolute)
        # the two halves of the native step's sweep: the sums, then - the
        # visiting order drawn meanwhile - the records IN that order (record
        # r from row order[r]); the matrix stays indexed by cell, rows that
        # will be scanned are written through at their own place
        order = rng.permutation(N)
        ll2, hint2 = ctx.ll_theta_pinned_top2_in_order(0, theta, .01, .2,
            K + 3, prior, order)
        hint2 = hint2.copy()
        for name in hint.dtype.names:
            assert np.array_equal(hint2[name], hint[name][order],
                equal_nan=True), (K, name)
        here2 = np.zeros(N, dtype=bool)
        here2[order] = hint2['row_here'] == 1
        assert np.array_equal(here2, here)
        assert np.array_equal(ll2[here2, :K], ll[here2, :K])
        ctx.matrix_wait()
        assert np.array_equal(ll2[:, :K], ll[:, :K])
    # the second half without the first, or with another evaluation in
    # between (it reuses the matrix's device buffer), is refused
    import ctypes as C
    lib = _lib.load()
    order = _lib.as_i64(rng.permutation(N))
    hint_p, host_p = C.c_void_p(), C.POINTER(C.c_double)()
    assert lib.bnpc_hints_in_order_issue(ctx._h, _lib.ptr(order, C.c_int64),
        C.byref(hint_p)) == 2
    assert 'no sums issued' in lib.bnpc_last_error().decode()
    theta = np.ascontiguousarray(theta[:5])
    prior = np.zeros(5)
    assert lib.bnpc_ll_theta_pinned_sums_issue(ctx._h, 0,
        _lib.ptr(theta, C.c_float), 5, .01, .2, 8, _lib.ptr(prior),
        C.byref(host_p)) == 0
    ctx.ll_theta(0, theta[:1], .02, .3)
    assert lib.bnpc_hints_in_order_issue(ctx._h, _lib.ptr(order, C.c_int64),
        C.byref(hint_p)) == 2
    bad = order.copy()
    bad[7] = N
    assert lib.bnpc_ll_theta_pinned_sums_issue(ctx._h, 0,
        _lib.ptr(theta, C.c_float), 5, .01, .2, 8, _lib.ptr(prior),
        C.byref(host_p)) == 0
    assert lib.bnpc_hints_in_order_issue(ctx._h, _lib.ptr(bad, C.c_int64),
        C.byref(hint_p)) == 2
    assert 'out of range' in lib.bnpc_last_error().decode()
    ctx.sync()
    ctx.close()


def test_chain_with_hundreds_of_clusters_matches_oracle(monkeypatch):
    """A running chain with 64 < K: every sweep and every parameter update
    inside the one-call step (no phase handed back), most cells decided from
    the device's hint - the same chain as the method-by-method walk and as
    the oracle, state by state, from the first sweep (K0 ~ 760, its whole
    matrix one hinted launch) on."""
    data = H.synth(35, 1200, 300, 90, 0.2)
    steps = 40
    nat = _chain_by_steps(P, 'learn', data, steps, 21, (.25, .25), '1',
        monkeypatch)
    ref = _chain_by_steps(P, 'learn', data, steps, 21, (.25, .25), '0',
        monkeypatch)
    orc = _chain_by_steps(O, 'learn', data, steps, 21, (.25, .25), '0',
        monkeypatch, peek=False)
    K_end = len(nat[0][-1][1])
    assert 80 <= K_end <= 110, K_end
    assert nat[1]['native_steps'] == steps, nat[1]
    assert ref[1]['native_steps'] == 0
    # from the second sweep on nearly every cell is decided from its record
    assert nat[1]['hint_used'] > 0.8 * nat[1]['swept'], nat[1]
    assert ref[1]['hint_used'] > 0.8 * ref[1]['swept'], ref[1]
    # ... most of them in the loop's lane for such cells (native step and
    # method-by-method walk alike: the same loop)
    assert nat[1]['lane_used'] > 0.7 * nat[1]['swept'], nat[1]
    assert ref[1]['lane_used'] == nat[1]['lane_used'], (ref[1], nat[1])
    for i, (a, b) in enumerate(zip(nat[0], ref[0])):
        for x, y in zip(a, b):
            if isinstance(x, np.ndarray):
                assert x.dtype == y.dtype and np.array_equal(x, y), i
            else:
                assert x == y, (i, x, y)
    for key in nat[2]:
        assert np.array_equal(nat[2][key], ref[2][key]), key
    for i, (a, c) in enumerate(zip(nat[0], orc[0])):
        assert np.array_equal(a[0], c[0]), i
        assert a[1] == c[1], i
        np.testing.assert_allclose(a[7], c[7], rtol=1e-9)


def test_fused_restricted_scan_changes_nothing(monkeypatch):
    """bnpc_rg_scan_step (device sums + assignment scan + counts + parameter
    batch in one native call) against the same scan made call by call: a
    split/merge-heavy chain is identical either way, and identical to the
    oracle's."""
    data = H.synth(9, 600, 150, 5, 0.15)
    kw = dict(sm_prob=.6, sm_steps=4, eup=.25)
    runs = []
    for fused in ('1', '0'):
        monkeypatch.setenv('BNPC_RG_FUSED', fused)
        runs.append(H.run_chain(H.make(P, 'learn', data), 30, 5, **kw))
    ro = H.run_chain(H.make(O, 'learn', data), 30, 5, **kw)
    for r in runs:
        assert np.array_equal(r['assignments'], ro['assignments'])
        np.testing.assert_allclose(r['ML'], ro['ML'], rtol=1e-9)
        assert np.array_equal(r['params'], ro['params'])
    assert np.array_equal(runs[0]['ML'], runs[1]['ML'])


# ------------------------------------------- device screen of the MH batches
def _screen_case(rng, ctx, data, K, theta_mode, prior, FP, FN, u_mode):
    """One batch: K clusters by label on `data`, old parameters by mode, the
    exact decisions from the SciPy-level arithmetic (CRP._mh_math)."""
    N, M = data.shape
    assign = rng.randint(0, K, N)
    assign[:K] = np.arange(K)
    n1, n0 = ctx.colcounts_by_label(assign, np.arange(K))
    if theta_mode == 'uniform':
        old = rng.uniform(size=(K, M))
    elif theta_mode == 'posterior':     # where a converged chain sits
        old = (n1 + .25) / (n1 + n0 + .5) + rng.normal(size=(K, M)) * 0.01
    else:                               # 'edges': at and next to the bounds
        old = rng.choice([P.TMIN, P.TMAX, 2e-5, 1 - 2e-5, 1e-4, .5], (K, M))
    old = np.clip(old, P.TMIN, P.TMAX).astype(np.float32)
    sd = np.array([0.1, 0.25, 0.5])
    sd_idx = rng.randint(0, 3, (K, M)).astype(np.int32)
    U = rng.uniform(size=(K, M))
    u = rng.uniform(size=(K, M))
    probe = P.CRP.__new__(P.CRP)
    probe.param_proposal_sd = sd
    probe.p, probe.q = prior
    probe.beta_prior_uniform = bool(prior[0] == prior[1] == 1)
    probe.FP, probe.FN = FP, FN
    std = sd[sd_idx]

    def exact(u_):
        with np.errstate(all='ignore'):
            return probe._mh_math(old, std, U, u_, n1, n0, False, None)
    new, A, decline, _ = exact(u)
    if u_mode == 'knife':
        # the uniform that sits ON the decision: exp(A) nudged both ways
        with np.errstate(all='ignore'):
            near = np.exp(np.clip(A, -700, -1e-300))
        eps = rng.choice([1e-15, 1e-12, 1e-9, 1e-6], (K, M)) \
            * rng.choice([-1, 1], (K, M))
        u = np.clip(near * (1 + eps), 1e-300, 1 - 1e-16)
        new, A, decline, _ = exact(u)
    flags, new32 = ctx.mh_screen(0, old, sd, (sd_idx, U, u), P.TMIN, P.TMAX,
        FP, FN, prior[0], prior[1], probe.beta_prior_uniform, with_theta=True)
    # flag 3: accepted for certain AND the device vouches for the proposal's
    # float32 bits - they are the exact path's, bit for bit
    given = flags == 3
    assert not (given & decline).any()
    assert np.array_equal(new32[given].view(np.int32),
        new[given].view(np.int32)), np.argwhere(
            given & (new32.view(np.int32) != new.view(np.int32)))[:3]
    _screen_case.given += int(given.sum())
    _screen_case.accepted += int((flags >= 2).sum())
    flags = np.where(given, 2, flags).astype(np.uint8)
    return flags, decline, A, u


_screen_case.given = _screen_case.accepted = 0


def test_mh_screen_never_rules_out_a_proposal_the_exact_arithmetic_accepts():
    """k_mh_screen (flag 0 = declined for certain) against the SciPy-level
    arithmetic on batches a chain meets and on adversarial ones: clusters of
    a few to thousands of cells, parameters drawn uniformly / sitting at
    their posterior / on the truncation bounds, both priors, mild and
    extreme error rates, and uniforms placed within 1e-15 .. 1e-6 of the
    decision itself.  Every flag-0 element is declined by the exact path,
    every flag-2 element accepted; away from the knife edge the screen
    decides nearly everything."""
    rng = np.random.RandomState(77)
    data = H.synth(4, 3000, 257, 5, 0.2)
    ctx = _lib.Context(data=data)
    ruled, declined_total = 0, 0
    _screen_case.given = _screen_case.accepted = 0
    try:
        # (the proposals the device vouches for, flag 3, on half a million
        # entries per case: clusters of one or two cells accept most of them)
        for theta_mode in ('uniform', 'posterior'):
            for prior in ((.25, .25), (1, 1)):
                _screen_case(rng, ctx, data, 2000, theta_mode, prior, .01, .2,
                    'random')
        # nearly every accepted proposal comes with its bits (not the ones
        # within 0.4 % of a rounding boundary, not those below ~1e-4)
        assert _screen_case.given > 200000 \
            and _screen_case.given > 0.9 * _screen_case.accepted, \
            (_screen_case.given, _screen_case.accepted)
        for K in (2, 9, 40):
            for theta_mode in ('uniform', 'posterior', 'edges'):
                for prior in ((.25, .25), (1, 1), (.75, 2.)):
                    for FP, FN in ((.01, .2), (1e-4, .45), (.3, 1e-3)):
                        for u_mode in ('random', 'knife'):
                            flags, decline, A, u = _screen_case(rng, ctx,
                                data, K, theta_mode, prior, FP, FN, u_mode)
                            bad = ((flags == 0) & ~decline) \
                                | ((flags == 2) & decline)
                            assert not bad.any(), (K, theta_mode, prior, FP,
                                FN, u_mode, np.argwhere(bad)[:3])
                            assert np.isin(flags, (0, 1, 2)).all()
                            if u_mode == 'random':
                                ruled += int((flags != 1).sum())
                                declined_total += flags.size
        assert ruled > 0.9 * declined_total, (ruled, declined_total)
    finally:
        ctx.close()


def test_screened_batch_on_the_device_equals_the_plain_batch():
    """_lib.mh_batch through bnpc_mh_batch_dev (draws into pinned memory ->
    k_mh_screen against the resident counts -> exact arithmetic on what is
    left) against the plain host batch from the same stream position: same
    parameters, declined counts, prior densities, stream position; the
    screen leaves the host a small share of the batch."""
    table = P._native_kernels()
    if table is None:
        pytest.skip('native parameter batch not available')
    rng = np.random.RandomState(5)
    data = H.synth(2, 2000, 300, 6, 0.2)
    ctx = _lib.Context(data=data)
    try:
        K = 7
        assign = rng.randint(0, K, 2000)
        n1, n0 = ctx.colcounts_by_label(assign, np.arange(K))
        old = np.clip((n1 + .25) / (n1 + n0 + .5), P.TMIN, P.TMAX) \
            .astype(np.float32)
        sd = np.array([0.1, 0.25, 0.5])
        outs = []
        for dev in (None, ctx):
            np.random.seed(11)
            outs.append(_lib.mh_batch(table, old, n1, n0, sd, P.TMIN, P.TMAX,
                .01, .2, .25, .25, False, False, want_prior=True, ctx=dev,
                counts_src=0) + (np.random.random(),))
        a, b = outs
        assert a[0] == b[0] == 0
        for i in (1, 3, 4):
            assert np.array_equal(a[i], b[i]), i
        assert a[6] == b[6]
        seen, kept = ctx.mh_screen_stats()
        assert seen == K * 300 and kept < 0.4 * seen, (seen, kept)
        # update_parameters in one call: the counts are made in the same call
        # (bnpc_label_counts_and_batch) and handed back with the results
        c1 = np.full_like(n1, -7)
        c0 = np.full_like(n0, -7)
        np.random.seed(11)
        c = _lib.mh_batch(table, old, c1, c0, sd, P.TMIN, P.TMAX, .01, .2,
            .25, .25, False, False, want_prior=True, ctx=ctx,
            label=(assign, np.arange(K))) + (np.random.random(),)
        assert c[0] == 0 and np.array_equal(c1, n1) and np.array_equal(c0, n0)
        for i in (1, 3, 4):
            assert np.array_equal(a[i], c[i]), i
        assert a[6] == c[6]
    finally:
        ctx.close()


def test_native_births_change_nothing(monkeypatch):
    """Clusters opened inside the native loop (the default) and through the
    Python branch (BNPC_NATIVE_BIRTHS=0), from a state that opens dozens of
    clusters in one sweep and recycles the ids of clusters that died: same
    assignments, cluster table, parameter rows, stream position."""
    rng = np.random.RandomState(3)
    profiles = (rng.random_sample((40, 60)) < 0.5).astype(float)
    data = np.repeat(profiles, 3, axis=0)
    data[rng.random_sample(data.shape) < 0.05] = np.nan
    outs = []
    for native in ('1', '0'):
        monkeypatch.setenv('BNPC_NATIVE_BIRTHS', native)
        m = P.CRP(data, DP_alpha=[200, 1], param_beta=[.25, .25],
            FN_error=0.01, FP_error=0.01)
        np.random.seed(4)
        m.init(mode='together')
        np.random.seed(5)
        m.update_assignments_Gibbs()
        np.random.seed(6)
        m.update_assignments_Gibbs()
        ids = list(m.cells_per_cluster)
        outs.append((m.assignment.copy(), list(m.cells_per_cluster.items()),
            m.parameters[ids].copy(), np.random.random()))
        m.close()
    a, b = outs
    assert len(a[1]) > 20
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and a[3] == b[3]


@pytest.mark.parametrize('pb', [(.25, .25), (1, 1)])
def test_native_moves_change_nothing(pb, monkeypatch):
    """Split / merge moves as one native call (bnpc_sm_move, the default)
    against the same moves walked step by step by the binding
    (BNPC_NATIVE_MOVES=0) and by the oracle: after EVERY move the result
    flags, the assignment, the cluster table, the parameter rows and the
    stream position are the same; both kinds of move get accepted and
    declined along the way."""
    data = H.synth(21, 700, 180, 6, 0.15)
    seen = {}
    for tag, mod, native in (('native', P, '1'), ('steps', P, '0'),
            ('oracle', O, '0')):
        monkeypatch.setenv('BNPC_NATIVE_MOVES', native)
        m = H.make(mod, 'learn', data, pb)
        np.random.seed(3)
        # the generator's clusters, 0 and 1 lumped together (splits get
        # accepted), the others cut in two at random (merges get accepted)
        gen = np.random.RandomState(21)
        gen.random_sample((6, 180))
        z = gen.randint(0, 6, 700)
        start = np.where(z < 2, 0, 2 * z + gen.randint(0, 2, 700))
        m.init(assign=list(start))
        m.update_parameters()
        trace = []
        for rnd in range(120):
            np.random.seed(500 + rnd)
            res = m.update_assignments_split_merge([.5, .5], 3)
            ids = list(m.cells_per_cluster)
            trace.append((res, m.assignment.copy(),
                list(m.cells_per_cluster.items()), m.parameters[ids].copy(),
                np.random.random()))
            if rnd % 4 == 3:
                m.update_parameters()
                if rnd % 8 == 7:
                    m.update_assignments_Gibbs()
        seen[tag] = (trace, getattr(m, '_native_moves', 0))
        if hasattr(m, 'close'):
            m.close()
    assert seen['native'][1] > 100 and seen['steps'][1] == 0
    accepted = {0: 0, 1: 0}
    for rnd, (a, b, c) in enumerate(zip(seen['native'][0], seen['steps'][0],
            seen['oracle'][0])):
        for other in (b, c):
            assert a[0] == other[0], rnd
            assert np.array_equal(a[1], other[1]), rnd
            assert [(int(k), int(v)) for k, v in a[2]] \
                == [(int(k), int(v)) for k, v in other[2]], rnd
            assert np.array_equal(a[3], other[3]), rnd
            assert a[4] == other[4], rnd
        accepted[int(a[0][1])] += a[0][0][0]
    assert accepted[0] > 0 and accepted[1] > 0, accepted


def _chain_by_steps(mod, kind, data, steps, seed, pb, native, monkeypatch,
        knobs=None, peek=True):
    """A chain stepped one Chain.step at a time; after every step: the state
    (labels, cluster table in dict order, parameter rows, alpha, FN, FP), the
    recorded scalars and one uniform peeked off the stream (the stream is put
    back).  Returns (trace, model statistics)."""
    from bnpc_amd.mcmc import Chain_steps
    monkeypatch.setenv('BNPC_NATIVE_STEP', native)
    np.random.seed(seed)
    m = H.make(mod, kind, data, pb)
    m.init()
    params = dict(sm_prob=.33, dpa_prob=.25, error_prob=.25 if kind == 'learn'
        else 0., sm_ratios=[.75, .25], sm_steps=3,
        param_proposal_sd=np.array([.1, .25, .5]))
    params.update(knobs or {})
    chain = Chain_steps(m, 1, steps, steps // 3, params, 0, False)
    trace = []
    for i in range(1, steps + 1):
        chain.step(i, i < steps // 3)
        ids = list(m.cells_per_cluster)
        state = np.random.get_state()
        u = np.random.random() if peek else 0.
        np.random.set_state(state)
        res = chain.results
        trace.append((m.assignment.copy(),
            [(int(a), int(b)) for a, b in m.cells_per_cluster.items()],
            m.parameters[ids].copy(), float(m.DP_a), float(m.FN),
            float(m.FP), m.CRP_prior.copy(), float(res['ML'][i]),
            float(res['MAP'][i]), float(res['DP_alpha'][i]),
            float(res['FN'][i]), float(res['FP'][i]),
            res['assignments'][i].copy(), u))
    chain.trace.finish()
    out = (trace, m.host_stats() if hasattr(m, 'host_stats') else {},
        {key: np.array(val) for key, val in chain.results.items()
            if isinstance(val, np.ndarray)},
        {name: chain.tally.counts[name].copy() for name in chain.tally.MOVES})
    if hasattr(m, 'close'):
        m.close()
    return out


@pytest.mark.parametrize('kind,pb,knobs', [
    ('learn', (.25, .25), None),
    ('fixed', (1, 1), None),
    ('learn', (.75, 2.), dict(sm_prob=.6, sm_steps=2, dpa_prob=.6,
        error_prob=.7, sm_ratios=[.4, .6])),
])
def test_native_steps_change_nothing(kind, pb, knobs, monkeypatch):
    """A whole step as ONE native call (bnpc_chain_step, the default) against
    the same steps walked method by method by the binding
    (BNPC_NATIVE_STEP=0): after EVERY one of 150 steps - the first sweeps
    from hundreds of clusters (handed back to the binding), then the
    converged regime - the labels, the cluster table in dict order, the
    parameter rows, alpha with its N + 2 prior vector, FN, FP, the recorded
    ML / MAP / traces and the stream position are bit-identical; the
    acceptance tallies and the parameter trace too; and the oracle agrees on
    the labels (ML to 1e-9)."""
    data = H.synth(33, 900, 220, 7, 0.15)
    steps = 150
    nat = _chain_by_steps(P, kind, data, steps, 77, pb, '1', monkeypatch, knobs)
    ref = _chain_by_steps(P, kind, data, steps, 77, pb, '0', monkeypatch, knobs)
    assert nat[1]['native_steps'] >= steps - 5, nat[1]
    assert ref[1]['native_steps'] == 0
    assert nat[1]['native_moves'] > 10
    for i, (a, b) in enumerate(zip(nat[0], ref[0])):
        for x, y in zip(a, b):
            if isinstance(x, np.ndarray):
                assert x.dtype == y.dtype and np.array_equal(x, y), i
            else:
                assert x == y, (i, x, y)
    assert sorted(nat[2]) == sorted(ref[2])
    for key in nat[2]:
        assert nat[2][key].dtype == ref[2][key].dtype, key
        assert np.array_equal(nat[2][key], ref[2][key]), key
    for name in nat[3]:
        assert np.array_equal(nat[3][name], ref[3][name]), name
    assert nat[3]['parameters'].sum() > 0 and nat[3]['splits'].sum() > 0
    orc = _chain_by_steps(O, kind, data, steps, 77, pb, '0', monkeypatch,
        knobs, peek=False)
    for i, (a, c) in enumerate(zip(nat[0], orc[0])):
        assert np.array_equal(a[0], c[0]), i
        assert a[1] == c[1], i
        np.testing.assert_allclose(a[7], c[7], rtol=1e-9)


@pytest.mark.parametrize('mode,kind,pb,knobs', [
    ('2', 'learn', (.25, .25), None),
    ('2', 'learn', (.75, 2.), dict(sm_prob=.6, sm_steps=2, dpa_prob=.6,
        error_prob=.7, sm_ratios=[.4, .6])),
    ('3', 'learn', (.25, .25), None),
    ('0', 'fixed', (1, 1), None),
])
def test_draws_taken_ahead_change_nothing(mode, kind, pb, knobs, monkeypatch):
    """the draws of a step's parameter batch taken AHEAD
    by a walker on the aside thread - on a copy of the stream, under the
    sweep's kernel and loop / under the tail of a move - and adopted by the
    batch iff the live stream stands exactly where the walker's stood
    (bnpc_mh_ahead_begin, MhAhead in bnpc_kernels.hip).  BNPC_MH_AHEAD=2 takes
    them for a batch of ANY size (the default starts at 8192 entries), 3
    takes and then throws them away (the discard path at every step), 0 never
    takes any: after EVERY one of 120 steps the chain is the chain walked
    method by method without a walker - labels, cluster table, parameter
    rows, alpha, error rates, recorded traces, the position of the stream -
    from the first sweeps (hundreds of clusters, births: walkers dropped)
    into the converged regime, moves accepted and rejected included."""
    data = H.synth(33, 900, 220, 7, 0.15)
    steps = 120
    monkeypatch.setenv('BNPC_MH_AHEAD', mode)
    nat = _chain_by_steps(P, kind, data, steps, 77, pb, '1', monkeypatch, knobs)
    monkeypatch.delenv('BNPC_MH_AHEAD')
    ref = _chain_by_steps(P, kind, data, steps, 77, pb, '0', monkeypatch, knobs)
    assert nat[1]['native_steps'] >= steps - 5, nat[1]
    begun, taken = nat[1]['ahead_begun'], nat[1]['ahead_taken']
    assert ref[1]['ahead_begun'] == 0
    if mode == '2':
        # a walker per native sweep / move; most are adopted (not: a sweep
        # with a birth, an alpha update after a move that changed K, a move
        # handed back)
        assert begun >= 0.8 * steps, nat[1]
        assert taken >= 0.6 * begun, nat[1]
        # (rows: a parameter batch's K + 1, a restricted scan's 2 or 3)
        assert nat[1]['ahead_rows'] >= 3 * taken
    elif mode == '3':
        assert begun >= 0.8 * steps and taken == 0, nat[1]
    else:
        assert begun == 0 and taken == 0, nat[1]
    for i, (a, b) in enumerate(zip(nat[0], ref[0])):
        for x, y in zip(a, b):
            if isinstance(x, np.ndarray):
                assert x.dtype == y.dtype and np.array_equal(x, y), i
            else:
                assert x == y, (i, x, y)
    for key in nat[2]:
        assert np.array_equal(nat[2][key], ref[2][key]), key
    for name in nat[3]:
        assert np.array_equal(nat[3][name], ref[3][name]), name


def test_native_step_hands_phases_back_and_resumes(monkeypatch):
    """Phases the library hands back to the binding (ch->need) and resumes
    after: without the sweep's hints, or with a sweep budget too small for one
    matrix, every Gibbs sweep goes back to update_assignments_Gibbs
    (NEED_GIBBS) while the rest of each step stays native - the chain is the
    same chain, state by state."""
    data = H.synth(34, 600, 150, 5, 0.1)
    base = _chain_by_steps(P, 'learn', data, 60, 5, (.25, .25), '0',
        monkeypatch)
    for env in ({'BNPC_SWEEP_HINT': '0'}, {'BNPC_SWEEP_BYTES': '90000'}):
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        got = _chain_by_steps(P, 'learn', data, 60, 5, (.25, .25), '1',
            monkeypatch)
        for key in env:
            monkeypatch.delenv(key)
        assert got[1]['native_steps'] >= 55
        for i, (a, b) in enumerate(zip(got[0], base[0])):
            for x, y in zip(a, b):
                if isinstance(x, np.ndarray):
                    assert np.array_equal(x, y), (env, i)
                else:
                    assert x == y, (env, i, x, y)


def test_step_whose_recording_is_handed_back(monkeypatch):
    """NEED_RECORD: the concentration parameter at or below the location of
    its Gamma prior (`-ap 0.01 30` on a data set of two clones: DP_a is
    floored at 1 + eps, the prior's support starts at 30) has log-density
    -inf in the reference, a value the
    library leaves to SciPy - every step's recording goes back to the binding
    (put_state), with the step's tallies complete; the same chain as the
    oracle's, MAP = -inf included.  Also: do_step without a recording target
    followed by update_results (the reference driver's two calls)."""
    from bnpc_amd.mcmc import MCMC
    import contextlib
    import io
    data = H.synth(36, 300, 80, 2, 0.1)

    def run(mod, two_calls=False):
        model = mod.CRP_errors_learning(data, DP_alpha=[0.01, 30.],
            param_beta=[.25, .25], FP_mean=0.01, FP_sd=0.01, FN_mean=0.2,
            FN_sd=0.1)
        mcmc = MCMC(model, sm_prob=.33, dpa_prob=.5, error_prob=.25,
            sm_ratios=[.75, .25], sm_steps=3)
        if two_calls:
            from bnpc_amd import mcmc as mcmc_mod
            monkeypatch.setattr(mcmc_mod.Chain, 'step',
                lambda self, step, burn_in=True: (self.do_step(),
                    self.update_results(step, burn_in)))
        with contextlib.redirect_stdout(io.StringIO()):
            mcmc.run((40, 13), 9, 1, 0, '', True)
        if two_calls:
            monkeypatch.undo()
        ran = mcmc.chains[0].model      # (the chain steps a copy)
        stats = ran.host_stats() if hasattr(ran, 'host_stats') else {}
        return mcmc.get_results()[0], stats
    ro, _ = run(O)
    rp, stats = run(P)
    assert stats['native_steps'] == 40, stats
    assert np.array_equal(ro['assignments'], rp['assignments'])
    assert np.isneginf(ro['MAP']).sum() > 20 \
        and np.array_equal(np.isneginf(ro['MAP']), np.isneginf(rp['MAP']))
    fin = np.isfinite(ro['MAP'])
    np.testing.assert_allclose(rp['MAP'][fin], ro['MAP'][fin], rtol=1e-9)
    np.testing.assert_allclose(rp['ML'], ro['ML'], rtol=1e-9)
    np.testing.assert_allclose(rp['FN'], ro['FN'], rtol=1e-9)
    r2, stats2 = run(P, two_calls=True)
    assert stats2['native_steps'] == 40, stats2
    for key in ('assignments', 'ML', 'MAP', 'DP_alpha', 'FN', 'FP'):
        assert np.array_equal(rp[key], r2[key]), key


@pytest.mark.parametrize('switch,value', [('BNPC_MH_SCREEN', '0'),
    ('BNPC_NATIVE_BETA', '0'), ('BNPC_STREAM_LIVE', '0'),
    ('BNPC_STREAM_LIVE', 'rng'), ('BNPC_ZERO_COPY', '0'),
    ('BNPC_DONE_WORDS', '0'), ('BNPC_MH_SCREEN', '2')])
def test_fallback_switches_walk_the_same_chain(switch, value, monkeypatch):
    """The documented fallbacks (README, environment switches) that no other
    test flips: the parameter batches without the device screen, Beta draws
    through NumPy, the stream exchanged through get_state / set_state instead
    of addressed in place (whole or the cached Gaussian only), small payloads
    through the copy engine, waits through the stream instead of the kernels'
    completion words - the same chain, bit for bit."""
    data = H.synth(8, 700, 180, 6, 0.15)

    def fresh():
        for cache in (P._NATIVE, P._BETA, P._STEP, P._MOVES_OK, _lib._live,
                _lib._gauss_live):
            cache.clear()
    monkeypatch.delenv(switch, raising=False)
    fresh()
    base = H.run_chain(H.make(P, 'learn', data), 40, 3, eup=.25)
    paths = P.fast_paths()
    assert all(paths.values()), paths
    monkeypatch.setenv(switch, value)
    fresh()
    try:
        got = H.run_chain(H.make(P, 'learn', data), 40, 3, eup=.25)
        off = P.fast_paths()
    finally:
        monkeypatch.delenv(switch)
        fresh()
    if switch == 'BNPC_NATIVE_BETA':
        assert not off['native_beta'] and not off['native_step']
    if switch == 'BNPC_STREAM_LIVE':
        assert not off['gauss_live'] and off['rng_live'] == (value == 'rng')
    for key in ('assignments', 'ML', 'MAP', 'DP_alpha', 'FN', 'FP', 'params'):
        assert np.array_equal(base[key], got[key]), key
