"""Direct tests of the native sequential loops (bnpc_sweeps.cpp) against
pure-Python / NumPy emulations of the reference loops
(libs/CRP.py:260-288 Gibbs, :616-629 and :808-818 restricted scans) on random
log-likelihood matrices - independent of any device code."""
import ctypes as C

import numpy as np
import pytest

from oracle import crp_numpy as O
from bnpc_amd import _lib

i64, f64 = C.c_int64, C.c_double


def reference_gibbs(ll, post_new, crp_prior, assignment, sizes, new_columns):
    """The reference's sweep over a given log-likelihood matrix; a cell that
    draws a new cluster gets the lowest free id and the next prepared column."""
    N = ll.shape[0]
    assignment = assignment.copy()
    cpc = dict(sizes)
    col_of = {cid: c for c, cid in enumerate(cpc)}
    ll = ll.copy()
    n_new = 0
    for cell in np.random.permutation(N):
        old = assignment[cell]
        if cpc[old] == 1:
            del cpc[old]
        else:
            cpc[old] -= 1
        ids = np.fromiter(cpc.keys(), dtype=int)
        cols = [col_of[i] for i in ids]
        post = ll[cell, cols] + crp_prior[np.fromiter(cpc.values(), dtype=int)]
        probs = O.CRP._normalize_log_probs(np.append(post, post_new[cell]))
        pick = np.random.choice(np.append(ids, -1), p=probs)
        if pick == -1:
            pick = next(i for i in range(N) if i not in cpc)
            ll = np.concatenate([ll, new_columns[n_new][:, None]], axis=1)
            col_of[pick] = ll.shape[1] - 1
            n_new += 1
        assignment[cell] = pick
        cpc[pick] = cpc.get(pick, 0) + 1
    return assignment, cpc, n_new


def native_gibbs(ll0, post_new, crp_prior, assignment, sizes, new_columns):
    lib = _lib.load()
    N, K = ll0.shape
    ld = K + 2                                  # forces the growth path
    ll = np.empty((N, ld))
    ll[:, :K] = ll0
    perm = _lib.as_i64(np.random.permutation(N))
    assignment = _lib.as_i64(assignment.copy())
    ids = np.fromiter(sizes.keys(), dtype=np.int64)
    col_of_id = np.full(N, -1, dtype=np.int64)
    col_of_id[ids] = np.arange(K)
    col_id = np.full(ld, -1, dtype=np.int64)
    col_id[:K] = ids
    col_size = np.zeros(ld, dtype=np.int64)
    col_size[:K] = np.fromiter(sizes.values(), dtype=np.int64)
    order = np.zeros(ld, dtype=np.int64)
    order[:K] = np.arange(K)
    scratch = np.empty(2 * (ld + 1))
    st = _lib.GibbsState(N, ld, K, K, 0, -1, N, -1)
    rng, extra = _lib.rng_export()
    n_new = 0
    while True:
        _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), C.byref(rng),
            _lib.ptr(perm, i64), _lib.ptr(ll, f64), _lib.ptr(post_new, f64),
            _lib.ptr(crp_prior, f64), _lib.ptr(assignment, i64),
            _lib.ptr(col_of_id, i64), _lib.ptr(col_id, i64),
            _lib.ptr(col_size, i64), _lib.ptr(order, i64),
            _lib.ptr(scratch, f64)), 'sweep')
        if st.new_cell < 0:
            break
        cell = int(st.new_cell)
        new_id = int(np.flatnonzero(col_of_id < 0)[0])
        if st.n_cols == ld:
            ll = np.concatenate([ll, np.empty((N, 4))], axis=1)
            col_id = np.concatenate([col_id, np.full(4, -1, np.int64)])
            col_size = np.concatenate([col_size, np.zeros(4, np.int64)])
            order = np.concatenate([order, np.zeros(4, np.int64)])
            ld += 4
            scratch = np.empty(2 * (ld + 1))
            st.ld = ld
        col = int(st.n_cols)
        ll[:, col] = new_columns[n_new]
        n_new += 1
        col_id[col], col_size[col], col_of_id[new_id] = new_id, 1, col
        order[st.n_active] = col
        st.n_active += 1
        st.n_cols += 1
        assignment[cell] = new_id
    _lib.rng_import(rng, extra)
    live = order[:st.n_active]
    return assignment, {int(col_id[c]): int(col_size[c]) for c in live}, n_new


@pytest.mark.parametrize('seed', range(24))
def test_gibbs_sweep_fuzz(seed):
    rng = np.random.RandomState(seed)
    N = int(rng.choice([1, 2, 17, 60, 200]))
    K = int(rng.randint(1, min(N, 12) + 1))
    spread = rng.choice([0.5, 5.0, 400.0])      # near ties ... far apart
    ll = -rng.random_sample((N, K)) * spread - 3
    post_new = -rng.random_sample(N) * spread - (3 if seed % 2 else 30)
    alpha = 2.5
    crp_prior = np.append(0, O.CRP.log_CRP_prior(
        np.append(np.arange(1, N + 1), alpha), N, alpha))
    labels = rng.randint(0, K, N)
    labels[:K] = np.arange(K)                   # every cluster is populated
    ids = rng.permutation(N)[:K] if N >= K else np.arange(K)
    assignment = ids[labels]
    sizes = {int(i): int((assignment == i).sum()) for i in ids}
    new_columns = [-rng.random_sample(N) * spread - 3 for _ in range(N + 1)]

    np.random.seed(100 + seed)
    want = reference_gibbs(ll, post_new, crp_prior, assignment, sizes,
        new_columns)
    tail_want = np.random.random(2)
    np.random.seed(100 + seed)
    got = native_gibbs(ll, post_new, crp_prior, assignment, sizes,
        new_columns)
    tail_got = np.random.random(2)
    assert np.array_equal(want[0], got[0])
    assert list(want[1].items()) == list(got[1].items())
    assert want[2] == got[2]
    assert np.array_equal(tail_want, tail_got)


def reference_rg_scan(ll, DP_a, rg, sample, target=None):
    """libs/CRP.py:609-632 (sample) and :808-818 (score `target`)."""
    rg = rg.copy()
    S = ll.shape[0]
    n = S + 2
    prob = np.zeros(S)
    order = np.random.permutation(S) if sample else range(S)
    for cell in order:
        rg[cell] = -1
        n_j = O.seqsum(rg) + 2
        n_i = n - n_j - 1
        lpost = ll[cell] + O.CRP.log_CRP_prior([n_i, n_j], n, DP_a)
        lprobs = O.CRP._normalize_log(lpost)
        if sample:
            pick = np.random.choice([0, 1], p=np.exp(lprobs))
        else:
            pick = target[cell]
        rg[cell] = pick
        prob[cell] = lprobs[pick]
    return rg, O.seqsum(prob)


@pytest.mark.parametrize('seed', range(24))
def test_rg_scan_fuzz(seed):
    lib = _lib.load()
    rng = np.random.RandomState(seed)
    S = int(rng.choice([1, 2, 3, 40, 333]))
    spread = rng.choice([0.1, 3.0, 50.0, 900.0])
    ll = np.ascontiguousarray(-rng.random_sample((S, 2)) * spread)
    if seed == 0:
        ll[:, 1] = ll[:, 0]                     # exact ties
    rg0 = rng.randint(0, 2, S).astype(np.int64)
    target = rng.randint(0, 2, S).astype(np.int64)
    DP_a = 3.7

    np.random.seed(seed)
    want_rg, want_p = reference_rg_scan(ll, DP_a, rg0, True)
    want_tail = np.random.random()
    np.random.seed(seed)
    st, extra = _lib.rng_export()
    rg = rg0.copy()
    out = C.c_double(0)
    _lib.check(lib.bnpc_rg_scan(C.byref(st), 0, S, _lib.ptr(ll, f64), DP_a,
        _lib.ptr(rg, i64), None, C.byref(out)), 'rg_scan')
    _lib.rng_import(st, extra)
    assert np.array_equal(rg, want_rg)
    np.testing.assert_allclose(out.value, want_p, rtol=1e-13, atol=1e-13)
    assert np.random.random() == want_tail

    want_rg, want_p = reference_rg_scan(ll, DP_a, rg0, False, target)
    rg = rg0.copy()
    _lib.check(lib.bnpc_rg_scan(None, 1, S, _lib.ptr(ll, f64), DP_a,
        _lib.ptr(rg, i64), _lib.ptr(target, i64), C.byref(out)), 'rg_scan')
    assert np.array_equal(rg, target) and np.array_equal(want_rg, target)
    np.testing.assert_allclose(out.value, want_p, rtol=1e-13, atol=1e-13)


def test_sweep_window_and_tile_rows():
    """pos/pos_end windows with position-indexed rows (tiled sweeps) visit
    the same cells with the same draws as one whole-matrix call."""
    lib = _lib.load()
    rng = np.random.RandomState(3)
    N, K = 90, 5
    ll = -rng.random_sample((N, K)) * 4
    post_new = np.full(N, -1e6)
    crp_prior = np.append(0, O.CRP.log_CRP_prior(
        np.append(np.arange(1, N + 1), 2.0), N, 2.0))
    start = rng.randint(0, K, N).astype(np.int64)
    start[:K] = np.arange(K)

    def run(windows):
        np.random.seed(9)
        perm = _lib.as_i64(np.random.permutation(N))
        st_rng, extra = _lib.rng_export()
        assignment = start.copy()
        ids = np.arange(K, dtype=np.int64)
        sizes = np.bincount(start, minlength=K).astype(np.int64)
        for lo, hi, tile in windows:
            k = ids.size
            ld = k + 1
            rows = perm[lo:hi] if tile else np.arange(N)
            mat = np.empty((rows.size, ld))
            mat[:, :k] = ll[rows][:, ids]
            col_of_id = np.full(N, -1, dtype=np.int64)
            col_of_id[ids] = np.arange(k)
            col_id = np.append(ids, -1).astype(np.int64)
            col_size = np.append(sizes, 0).astype(np.int64)
            order = np.append(np.arange(k), 0).astype(np.int64)
            scratch = np.empty(2 * (ld + 1))
            st = _lib.GibbsState(N, ld, k, k, lo, -1, hi, lo if tile else -1)
            _lib.check(lib.bnpc_gibbs_sweep(C.byref(st), C.byref(st_rng),
                _lib.ptr(perm, i64), _lib.ptr(mat, f64),
                _lib.ptr(post_new, f64), _lib.ptr(crp_prior, f64),
                _lib.ptr(assignment, i64), _lib.ptr(col_of_id, i64),
                _lib.ptr(col_id, i64), _lib.ptr(col_size, i64),
                _lib.ptr(order, i64), _lib.ptr(scratch, f64)), 'sweep')
            assert st.pos == hi and st.new_cell == -1
            live = order[:st.n_active]
            ids, sizes = col_id[live].copy(), col_size[live].copy()
        return assignment, ids.tolist(), sizes.tolist()

    whole = run([(0, N, False)])
    tiled = run([(0, 7, True), (7, 50, True), (50, N, True)])
    assert np.array_equal(whole[0], tiled[0]) and whole[1:] == tiled[1:]
    # a malformed window is refused
    st = _lib.GibbsState(N, 2, 1, 1, 5, -1, 3, -1)
    st_rng, _ = _lib.rng_export()
    z = np.zeros(8)
    zi = np.zeros(N, dtype=np.int64)
    rc = lib.bnpc_gibbs_sweep(C.byref(st), C.byref(st_rng), _lib.ptr(zi, i64),
        _lib.ptr(z, f64), _lib.ptr(z, f64), _lib.ptr(z, f64),
        _lib.ptr(zi, i64), _lib.ptr(zi, i64), _lib.ptr(zi, i64),
        _lib.ptr(zi, i64), _lib.ptr(zi, i64), _lib.ptr(z, f64))
    assert rc != 0 and b'window' in lib.bnpc_last_error()


def test_native_mh_draws_equal_numpy():
    sd = np.array([0.1, 0.25, 0.5])
    for seed, (G, M) in enumerate([(1, 1), (3, 100), (10, 1000), (2, 4097)]):
        np.random.seed(seed)
        want = [(np.random.choice(sd, size=M), np.random.uniform(size=M),
            np.random.random(M)) for _ in range(G)]
        tail = np.random.random(3)
        np.random.seed(seed)
        idx, U, u = _lib.mh_draws(G, M, 3)
        assert np.array_equal(np.random.random(3), tail)   # same consumption
        for g in range(G):
            assert np.array_equal(sd[idx[g]], want[g][0])
            assert np.array_equal(U[g], want[g][1])
            assert np.array_equal(u[g], want[g][2])


def test_stream_in_place_and_copy_fallback(monkeypatch):
    """Native draws act on NumPy's own MT19937 state in place; if that
    address is not available a copy is exchanged - same stream either way."""
    lib = _lib.load()
    assert _lib.rng_live() is not None
    outs = []
    for live in (True, False):
        if not live:
            monkeypatch.setattr(_lib, 'rng_live', lambda: None)
        np.random.seed(77)
        a = np.random.random()
        with _lib.NumpyStream() as rng:
            b = lib.bnpc_mt_random_sample(rng)
            perm = np.empty(50, dtype=np.int64)
            lib.bnpc_mt_permutation(rng, 50, _lib.ptr(perm, i64))
        c = np.random.random()
        outs.append((a, b, perm.tolist(), c))
    assert outs[0] == outs[1]
    np.random.seed(77)
    ref = (np.random.random(), np.random.random(),
        np.random.permutation(50).tolist(), np.random.random())
    assert outs[0] == ref


def test_dominated_cdf_closed_form_equals_numpy_cumsum():
    """One cluster dominates -> p = (floor, ..., 1.0, ..., floor).  The native
    sweep does not walk np.cumsum(p) but evaluates it in closed form
    (bnpc_dominated_cdf): identical bits for every position of the winner,
    from 1 to 40000 entries."""
    lib = _lib.load()
    floor = np.exp(np.clip(np.array([-1e3]), O.log_EPSILON, 0))[0]
    rng = np.random.RandomState(0)
    for A in (0, 1, 2, 7, 300, 4097, 40000):
        tops = {0, A, A // 2, A // 3} | set(rng.randint(0, A + 1, 4).tolist())
        for top in sorted(tops):
            p = np.full(A + 1, floor)
            p[top] = 1.0
            want = np.cumsum(p)
            got = np.empty(A + 1)
            _lib.check(lib.bnpc_dominated_cdf(A, top, _lib.ptr(got, f64)),
                'dominated_cdf')
            assert np.array_equal(want, got), (A, top)
    assert lib.bnpc_dominated_cdf(3, 4, _lib.ptr(np.empty(4), f64)) == 2


def test_dominated_rows_with_thousands_of_clusters():
    """The shape of a first sweep: thousands of live clusters, one of them
    far ahead for every cell.  Native loop == NumPy loop."""
    rng = np.random.RandomState(3)
    N, K = 2100, 2000
    ll = -rng.random_sample((N, K)) * 50 - 400
    best = rng.randint(0, K, N)
    ll[np.arange(N), best] = -5.0
    post_new = np.full(N, -300.0)
    post_new[::97] = -1.0           # some cells open a cluster instead
    alpha = 2.5
    crp_prior = np.append(0, O.CRP.log_CRP_prior(
        np.append(np.arange(1, N + 1), alpha), N, alpha))
    labels = np.append(np.arange(K), rng.randint(0, K, N - K))
    sizes = {}
    for lab in labels:
        sizes[int(lab)] = sizes.get(int(lab), 0) + 1
    new_columns = [-rng.random_sample(N) * 5 - 2 for _ in range(N)]
    np.random.seed(9)
    ref = reference_gibbs(ll, post_new, crp_prior, labels, dict(sizes),
        new_columns)
    np.random.seed(9)
    got = native_gibbs(ll, post_new, crp_prior, labels, dict(sizes),
        new_columns)
    assert np.array_equal(ref[0], got[0]) and ref[1] == got[1]
    assert ref[2] == got[2] and ref[2] > 0


def test_non_finite_posteriors_are_errors():
    """NaN / all -inf log posteriors have no counterpart in the native loops
    (the reference takes its FloatingPointError branches there): the calls
    fail loudly, before any draw for the cell, instead of opening clusters."""
    lib = _lib.load()
    N, K, ld = 4, 2, 3
    crp = np.append(0, O.CRP.log_CRP_prior(
        np.append(np.arange(1, N + 1), 2.0), N, 2.0))
    for poison in (np.nan, -np.inf):
        ll = np.full((N, ld), -5.0)
        ll[2, :] = poison
        post_new = np.full(N, -9.0)
        post_new[2] = poison
        np.random.seed(1)
        perm = _lib.as_i64(np.arange(N))
        assignment = np.array([0, 0, 1, 1], dtype=np.int64)
        col_of_id = np.array([0, 1, -1, -1], dtype=np.int64)
        col_id = np.array([0, 1, -1], dtype=np.int64)
        col_size = np.array([2, 2, 0], dtype=np.int64)
        order = np.array([0, 1, 0], dtype=np.int64)
        scratch = np.empty(2 * (ld + 1))
        st = _lib.GibbsState(N, ld, K, K, 0, -1, N, -1)
        with _lib.NumpyStream() as rng:
            rc = lib.bnpc_gibbs_sweep(C.byref(st), rng, _lib.ptr(perm),
                _lib.ptr(ll), _lib.ptr(post_new), _lib.ptr(crp),
                _lib.ptr(assignment), _lib.ptr(col_of_id), _lib.ptr(col_id),
                _lib.ptr(col_size), _lib.ptr(order), _lib.ptr(scratch))
        assert rc == 4 and st.pos == 2
        assert b'non-finite' in lib.bnpc_last_error()
        rg = np.zeros(N, dtype=np.int64)
        out = C.c_double(0)
        llrg = np.full((N, 2), -3.0)
        llrg[1] = poison
        rc = lib.bnpc_rg_scan(None, 1, N, _lib.ptr(llrg), 2.0, _lib.ptr(rg),
            _lib.ptr(np.zeros(N, dtype=np.int64)), C.byref(out))
        assert rc == 4
