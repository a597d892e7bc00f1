"""Direct tests of the native sequential loops (bnpc_sweeps.cpp) against
pure-Python / NumPy emulations of the reference loops
(libs/CRP.py:260-288 Gibbs, :616-629 and :808-818 restricted scans) on random
log-likelihood matrices - independent of any device code."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import crp_numpy as O
from bnpc_amd import _lib, model as P

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


def native_gibbs(ll0, post_new, crp_prior, assignment, sizes, new_columns,
            hint=None, used=None, in_order=False):
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
    if hint is not None:        # (hints, priors at launch) for columns 0..K-1
        records = hint[0]
        if in_order:            # record p = the cell at position p of perm
            records = np.ascontiguousarray(records[perm])
            st.hint_in_order = 1
        st.hint = records.ctypes.data
        st.hint_prior = hint[1].ctypes.data
        st.hint_cols = K
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
    if used is not None:
        used.append(int(st.hint_used))
        native_gibbs.last = (int(st.hint_used), int(st.pair_used),
            int(st.triple_used), int(st.lane_used), int(st.stride_used))
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


# ------------------------------------------- short cuts of the native loops
def test_loop_shortcuts_change_nothing(monkeypatch):
    """The two arithmetic short cuts of the native loops - no exp() for an
    entry more than 60 below the runner-up of a scanned cell, no log1p / exp
    for a restricted-scan pair more than 40 apart - against
    the same loops with BNPC_LOOP_SHORTCUTS=0: 300 random sweeps (matrices
    with near ties, moderate and 400-wide spreads; clusters dying and being
    born) and 400 random scans give identical assignments, cluster tables,
    numbers of clusters opened, log-probability BITS and stream positions."""
    lib = _lib.load()

    def sweep(seed):
        rng = np.random.RandomState(seed)
        N = int(rng.choice([17, 60, 200]))
        K = int(rng.randint(2, 13))
        spread = rng.choice([0.5, 5.0, 70.0, 400.0])
        ll = -rng.random_sample((N, K)) * spread - 3
        # a third of the cells torn between two close candidates
        close = rng.random_sample(N) < 0.33
        a, b = rng.randint(0, K, N), rng.randint(0, K, N)
        ll[close, a[close]] = ll[close, b[close]] = -3.0
        post_new = -rng.random_sample(N) * spread - (3 if seed % 3 else 30)
        alpha = 2.5
        crp_prior = np.append(0, O.CRP.log_CRP_prior(
            np.append(np.arange(1, N + 1), alpha), N, alpha))
        labels = rng.randint(0, K, N)
        labels[:K] = np.arange(K)
        ids = rng.permutation(N)[:K]
        assignment = ids[labels]
        sizes = {int(i): int((assignment == i).sum()) for i in ids}
        new_columns = [-rng.random_sample(N) * spread - 3
            for _ in range(N + 1)]
        np.random.seed(1000 + seed)
        out = native_gibbs(ll, post_new, crp_prior, assignment, sizes,
            new_columns)
        return out[0], list(out[1].items()), out[2], np.random.random(2)

    def scan(seed):
        rng = np.random.RandomState(5000 + seed)
        S = int(rng.choice([3, 40, 333]))
        spread = rng.choice([0.1, 3.0, 50.0, 90.0, 900.0])
        ll = np.ascontiguousarray(-rng.random_sample((S, 2)) * spread)
        rg = rng.randint(0, 2, S).astype(np.int64)
        target = rng.randint(0, 2, S).astype(np.int64)
        np.random.seed(seed)
        st, extra = _lib.rng_export()
        out = C.c_double(0)
        _lib.check(lib.bnpc_rg_scan(C.byref(st), 0, S, _lib.ptr(ll, f64), 3.7,
            _lib.ptr(rg, i64), None, C.byref(out)), 'rg_scan')
        sampled = (rg.copy(), out.value, int(st.pos), bytes(st.key))
        _lib.check(lib.bnpc_rg_scan(None, 1, S, _lib.ptr(ll, f64), 3.7,
            _lib.ptr(rg, i64), _lib.ptr(target, i64), C.byref(out)),
            'rg_scan')
        return sampled, out.value

    monkeypatch.setenv('BNPC_LOOP_SHORTCUTS', '1')
    new_sweeps = [sweep(s) for s in range(300)]
    new_scans = [scan(s) for s in range(400)]
    monkeypatch.setenv('BNPC_LOOP_SHORTCUTS', '0')
    moves = 0
    for seed, new in enumerate(new_sweeps):
        old = sweep(seed)
        assert np.array_equal(old[0], new[0]), seed
        assert old[1] == new[1] and old[2] == new[2], seed
        assert np.array_equal(old[3], new[3]), seed
        moves += old[0].size
    for seed, new in enumerate(new_scans):
        old = scan(seed)
        assert np.array_equal(old[0][0], new[0][0]), seed
        assert old[0][1:] == new[0][1:], seed       # log-prob bits, stream
        assert old[1] == new[1], seed
    assert moves > 20000


# ------------------------------------------------------ host thread team
_STRESS = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
from bnpc_amd import _lib, model as P
lib = _lib.load()

def batch(jobs, tasks, ranks, seed):
    done, want = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.bnpc_team_stress(jobs, tasks, ranks, seed, C.byref(done),
        C.byref(want)), 'team_stress')
    assert done.value == want.value and want.value >= jobs, (done.value,
        want.value)

jobs = int(sys.argv[1])
for ranks in (1, 2, 3, 16, 64):
    batch(jobs, 64, ranks, ranks)
# a fork() between batches: the child builds a team of its own
pid = os.fork()
if pid == 0:
    try:
        batch(jobs // 4, 64, 16, 99)
        batch(jobs // 4, 5, 3, 98)
        os._exit(0)
    except BaseException:
        os._exit(1)
_, status = os.waitpid(pid, 0)
assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
batch(jobs // 4, 64, 16, 7)         # the parent's team still works
print('ok')
'''


@pytest.mark.parametrize('spin_us', ['0', '50'])
def test_thread_team_stress(spin_us):
    """10^4 jobs (BNPC_STRESS_JOBS; the 10^5 runs are logged under
    profiles/r03) of 1-64 tasks per rank count 1 / 2 / 3 / 16 / 64 (the team
    grows in place), tiny jobs back to back, a fork() in between, with and
    without the spin before a worker parks (BNPC_HOST_SPIN_US): every task
    runs exactly once and nothing hangs (the lost
    wake-up of dd42c54 would hang here)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BNPC_HOST_SPIN_US=spin_us)
    res = subprocess.run([sys.executable, '-c', _STRESS % dict(root=root),
        os.environ.get('BNPC_STRESS_JOBS', '10000')], env=env,
        capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'ok' in res.stdout, res.stderr[-2000:]


def test_team_entry_points_from_two_threads():
    """two host threads that each call a team-using entry point
    (ctypes drops the GIL) used to corrupt the team's job word and hang.
    They now take turns: both loops finish and every result is the
    one-thread result."""
    import threading
    from bnpc_amd import hostkernels
    table = hostkernels.table()
    if table is None:
        pytest.skip('SciPy kernel table not available')
    rng = np.random.RandomState(0)
    sd = np.array([0.1, 0.25, 0.5])

    def problem(G, M):
        old = np.clip(rng.uniform(size=(G, M)), 1e-5, 1 - 1e-5) \
            .astype(np.float32)
        n1 = rng.randint(0, 500, (G, M)).astype(np.int32)
        n0 = rng.randint(0, 500, (G, M)).astype(np.int32)
        draws = (rng.randint(0, 3, (G, M)).astype(np.int32),
            rng.uniform(size=(G, M)), rng.uniform(size=(G, M)))
        return old, n1, n0, draws

    def call(p, threads):
        old, n1, n0, draws = p
        # (the per-shape scratch of _lib.mh_batch is shared: private buffers)
        G, M = old.shape
        a = _lib.MHArgs()
        bufs = dict(new=np.empty((G, M), np.float32), A=np.empty((G, M)),
            lp=np.empty(G), dec=np.empty(G, np.int64))
        a.G, a.M = G, M
        a.old_theta, a.n1, a.n0 = (x.ctypes.data for x in (old, n1, n0))
        a.sd, a.n_sd = sd.ctypes.data, 3
        a.tmin, a.tmax, a.FP, a.FN, a.p, a.q = 1e-5, 1 - 1e-5, .01, .2, 1, 1
        a.uniform_prior, a.trans_prob = 1, 0
        a.sd_idx, a.U, a.u = (x.ctypes.data for x in draws)
        a.new_theta, a.A = bufs['new'].ctypes.data, bufs['A'].ctypes.data
        a.log_prob, a.declined = bufs['lp'].ctypes.data, bufs['dec'].ctypes.data
        a.threads = threads
        status = C.c_int(0)
        _lib.check(_lib.load().bnpc_mh_batch(C.addressof(table), None,
            C.byref(a), C.byref(status)), 'mh_batch')
        assert status.value == 0
        return bufs['new'], bufs['dec']

    probs = [problem(3, 700), problem(7, 1300)]
    want = [call(p, 1) for p in probs]
    errors = []

    def loop(i):
        try:
            for _ in range(100):
                new, dec = call(probs[i], 8)
                assert np.array_equal(new, want[i][0])
                assert np.array_equal(dec, want[i][1])
        except BaseException as err:        # noqa: BLE001
            errors.append(err)

    workers = [threading.Thread(target=loop, args=(i,), daemon=True)
        for i in range(2)]
    for w in workers:
        w.start()
    for w in workers:
        w.join(timeout=120)
    assert not any(w.is_alive() for w in workers), 'team calls hang'
    assert not errors, errors


# ------------------------------------------------- legacy Beta / Gamma sampler
@pytest.mark.parametrize('live', ['1', '0'])
def test_native_beta_equals_numpy_golden(golden_dir, live, monkeypatch):
    """bnpc_mt_beta against vectors NumPy itself drew on the reference's
    stack (tests/golden/rng_beta.npz: single-cell shapes under two priors,
    cluster-sized shapes, a mix over every branch of the legacy sampler),
    with and without a cached Gaussian on entry; the gamma draws NumPy makes
    AFTERWARDS (they share the cached Gaussian), the cache itself and the
    stream position must come out as in the capture - through the located
    in-place pointer and through the get_state / set_state fallback."""
    monkeypatch.setenv('BNPC_STREAM_LIVE', '1' if live == '1' else 'rng')
    _lib._gauss_live.clear()
    try:
        assert (_lib.gauss_live() is not None) == (live == '1')
        t = np.load(os.path.join(golden_dir, 'rng_beta.npz'))
        for seed in (3, 42):
            for pre_gauss in (0, 1):
                np.random.seed(seed)
                if pre_gauss:
                    np.random.normal()
                key = f's{seed}_g{pre_gauss}_'
                for name in ('cell_q', 'cell_u', 'cluster', 'mixed'):
                    got = _lib.beta(t[name + '_a'], t[name + '_b'])
                    assert np.array_equal(got, t[key + name]), (key, name)
                gam = [np.random.gamma(3.5, 0.7), np.random.gamma(0.4, 2.0),
                    np.random.gamma(1.0, 1.5)]
                assert np.array_equal(gam, t[key + 'gamma'])
                st = np.random.get_state()
                assert st[3] == t[key + 'has_gauss'][0]
                assert st[4] == t[key + 'gauss'][0]
                assert np.array_equal(np.random.random(3), t[key + 'tail'])
    finally:
        _lib._gauss_live.clear()


def test_native_beta_theta_and_this_numpy():
    """The profile-row form (counts -> clipped float32 row) against the NumPy
    of THIS process on random counts incl. empty columns, interleaved with
    NumPy's own normal / gamma draws; bad shapes are errors, not draws."""
    rng = np.random.RandomState(8)
    for M, hi in ((1, 2), (130, 2), (1000, 900), (5000, 40)):
        n1 = rng.randint(0, hi, M).astype(np.int32)
        n0 = rng.randint(0, hi, M).astype(np.int32)
        for p, q in ((.25, .25), (1., 1.), (.75, 2.)):
            np.random.seed(M)
            np.random.normal(size=3)
            want = np.clip(np.random.beta(p + n1, q + n0), 1e-5, 1 - 1e-5) \
                .astype(np.float32)
            w_tail = (np.random.gamma(2.5), np.random.normal(),
                np.random.random())
            np.random.seed(M)
            np.random.normal(size=3)
            got = _lib.beta_theta(p, q, n1, n0, 1, 1e-5, 1 - 1e-5)
            g_tail = (np.random.gamma(2.5), np.random.normal(),
                np.random.random())
            assert np.array_equal(want, got) and w_tail == g_tail
    pos = np.random.get_state()[2]
    with pytest.raises(RuntimeError):
        _lib.beta(np.array([1., 0.]), np.array([1., 1.]))
    with pytest.raises(RuntimeError):
        _lib.beta(np.array([1., np.nan]), np.array([1., 1.]))
    assert np.random.get_state()[2] == pos      # nothing was drawn


def test_sweep_hints_fuzz(monkeypatch):
    """The per-cell hint (three largest entries, the two best columns and
    their log-likelihoods) against the plain scan on 200 random sweeps whose
    matrices hold what a running chain holds: clear winners, cells torn
    between two near-duplicate columns (the halves of a fresh split) with
    everything else far below, three-way ties, clusters that die and are
    born under the hint.  Same assignments, cluster tables, births, stream;
    most cells are decided without a scan."""
    decided = cells = pairs = triples = 0
    in_lane, in_stride = {}, {}
    for seed in range(200):
        rng = np.random.RandomState(7000 + seed)
        N = int(rng.choice([30, 90, 200]))
        K = int(rng.randint(3, 13))
        ll = -rng.random_sample((N, K)) * 300 - 200       # far below
        a = rng.randint(0, K, N)
        b = (a + 1 + rng.randint(0, K - 1, N)) % K
        ll[np.arange(N), a] = -50 - rng.random_sample(N) * 5
        kind = rng.random_sample(N)
        two = kind < 0.4                # torn between two columns
        ll[two, b[two]] = ll[two, a[two]] - rng.random_sample(two.sum()) * 3
        three = kind > 0.8              # a third within reach: decided
        c = (b + 1) % K                 # among the three (or scanned)
        c[c == a] = (c[c == a] + 1) % K
        ll[three, b[three]] = ll[three, a[three]] \
            - rng.random_sample(three.sum()) * 3
        ll[three, c[three]] = ll[three, a[three]] \
            - rng.random_sample(three.sum()) * 20
        four = kind > 0.97              # ... and a fourth: scanned
        d = (c + 1) % K
        d[(d == a) | (d == b) | (d == c)] = -1
        sel = four & (d >= 0) & (K > 3)
        ll[sel, d[sel]] = ll[sel, a[sel]] - 25
        post_new = -rng.random_sample(N) * 50 - (300 if seed % 4 else 60)
        alpha = 2.5
        crp_prior = np.append(0, O.CRP.log_CRP_prior(
            np.append(np.arange(1, N + 1), alpha), N, alpha))
        labels = rng.randint(0, K, N)
        if seed % 3 == 0:
            # a settled chain: nine cells in ten sit in the cluster that
            # dominates them (what the lane's stride takes in runs) - with
            # winners far enough above the rest for the dominance margin
            labels = np.where(rng.random_sample(N) < 0.9, a, labels)
            clear = ~two & ~three
            ll[clear, a[clear]] += 40
        labels[:K] = np.arange(K)
        ids = rng.permutation(N)[:K]
        assignment = ids[labels]
        sizes = {int(i): int((assignment == i).sum()) for i in ids}
        new_columns = [-rng.random_sample(N) * 300 - 100
            for _ in range(N + 1)]
        col_prior = np.ascontiguousarray(
            crp_prior[np.fromiter(sizes.values(), dtype=np.int64)])
        hint = _lib.hints_from_matrix(ll, col_prior)
        outs, used = [], []
        # (no hints / records by cell / records in visiting order)
        for h, in_order, lane in ((None, False, None),
                ((hint, col_prior), False, None),
                ((hint, col_prior), True, None),
                ((hint, col_prior), True, '0'),
                ((hint, col_prior), seed % 2 == 0, '0.05'),
                ((hint, col_prior), seed % 2 == 1, '0.3'),
                ((hint, col_prior), seed % 2 == 0, 'nostride')):
            # (lane: BNPC_SWEEP_LANE - the lane of the cells decided from
            # their records switched off, or handing a share of its cells
            # over to the general iteration with their uniforms drawn)
            if lane is None:
                monkeypatch.delenv('BNPC_SWEEP_LANE', raising=False)
            else:
                monkeypatch.setenv('BNPC_SWEEP_LANE', lane)
            np.random.seed(seed)
            got = native_gibbs(ll, post_new, crp_prior, assignment, sizes,
                new_columns, hint=h, used=used, in_order=in_order)
            outs.append((got[0], list(got[1].items()), got[2],
                np.random.random(2)))
            if h is not None:
                in_lane[lane] = in_lane.get(lane, 0) + native_gibbs.last[3]
                in_stride[lane] = in_stride.get(lane, 0) \
                    + native_gibbs.last[4]
        for other in outs[1:]:
            assert np.array_equal(outs[0][0], other[0]), seed
            assert outs[0][1] == other[1] and outs[0][2] == other[2], seed
            assert np.array_equal(outs[0][3], other[3]), seed
        assert used[1] == used[2] == used[3] == used[4] == used[5] \
            == used[6], seed
        decided += used[1]
        pairs += native_gibbs.last[1]
        triples += native_gibbs.last[2]
        cells += N
    assert decided > 0.7 * cells, (decided, cells)
    assert pairs > 0.1 * cells and triples > 0.05 * cells, (pairs, triples)
    # the lane took its share (both record layouts: twice the cells), less
    # when it hands cells over, nothing when it is switched off
    assert in_lane[None] > 0.25 * 2 * cells, (in_lane, cells)
    assert in_lane['0'] == 0 and 0 < in_lane['0.3'] < in_lane[None], in_lane
    # ... and the lane's stride (runs of cells that stay where they are): a
    # good share of the lane's cells in the settled sweeps, fewer when every
    # third uniform ends a run, none when it is switched off - the lane then
    # takes the same cells one by one
    assert in_stride[None] > 0.1 * in_lane[None], (in_stride, in_lane)
    assert 0 < in_stride['0.3'] < in_stride[None] / 2, in_stride
    assert in_stride['nostride'] == 0 and in_stride['0'] == 0, in_stride
    assert in_lane['nostride'] * 2 == in_lane[None] \
        or in_lane['nostride'] > 0.45 * in_lane[None], in_lane


@pytest.mark.parametrize('K', [65, 150, 400, 1500])
def test_sweep_hints_with_hundreds_of_columns(K, monkeypatch):
    """The same comparison for sweeps of MORE than 64 columns (a running
    chain with hundreds of clusters; the first sweep of a data set whose
    matrix fits the host budget): the hint's columns are found in the live
    list by bisection instead of a table, pairs and triples are decided among
    hundreds of floor entries.  Two kinds of matrices: settled ones (clear
    winners, torn cells) and first-sweep ones (every column near every other
    one - all cells scanned - until clusters are born that dominate)."""
    decided = cells = pairs = triples = 0
    in_lane = {}
    for seed in range(12):
        rng = np.random.RandomState(9000 + 31 * K + seed)
        N = int(rng.choice([2 * K, 3 * K + 7, 4 * K]))
        first_sweep = seed % 3 == 2
        if first_sweep:
            ll = -400 - rng.random_sample((N, K)) * 6
        else:
            ll = -rng.random_sample((N, K)) * 300 - 200       # far below
            a = rng.randint(0, K, N)
            b = (a + 1 + rng.randint(0, K - 1, N)) % K
            ll[np.arange(N), a] = -50 - rng.random_sample(N) * 5
            kind = rng.random_sample(N)
            two = kind < 0.3
            ll[two, b[two]] = ll[two, a[two]] \
                - rng.random_sample(two.sum()) * 3
            three = kind > 0.8
            c = (b + 1) % K
            c[c == a] = (c[c == a] + 1) % K
            ll[three, b[three]] = ll[three, a[three]] \
                - rng.random_sample(three.sum()) * 3
            ll[three, c[three]] = ll[three, a[three]] \
                - rng.random_sample(three.sum()) * 20
            four = kind > 0.97
            d = (c + 1) % K
            d[(d == a) | (d == b) | (d == c)] = -1
            sel = four & (d >= 0)
            ll[sel, d[sel]] = ll[sel, a[sel]] - 25
            # exact ties between two columns (first in list order wins)
            tie = (kind > 0.3) & (kind < 0.33)
            ll[tie, b[tie]] = ll[tie, a[tie]]
        post_new = -rng.random_sample(N) * 50 - 300
        if seed % 4 == 0:       # a few cells that may open a cluster
            near = rng.random_sample(N) < 0.04
            post_new[near] = -60 - rng.random_sample(near.sum()) * 10
        if first_sweep:
            post_new = -395 - rng.random_sample(N) * 6
        alpha = 2.5
        crp_prior = np.append(0, O.CRP.log_CRP_prior(
            np.append(np.arange(1, N + 1), alpha), N, alpha))
        labels = rng.randint(0, K, N)
        if not first_sweep:     # most cells sit in the cluster they like best
            labels = np.where(rng.random_sample(N) < 0.8, a, labels)
        labels[rng.permutation(N)[:K]] = np.arange(K)
        ids = rng.permutation(N)[:K]
        assignment = ids[labels]
        sizes = {int(i): int((assignment == i).sum()) for i in ids}
        if first_sweep:
            # born clusters: each far above everything for a tenth of the cells
            new_columns = []
            for j in range(N + 1):
                col = -400 - rng.random_sample(N) * 300
                col[rng.random_sample(N) < 0.1] = -60 - rng.random_sample() * 5
                new_columns.append(col)
        else:
            new_columns = [-rng.random_sample(N) * 300 - 100
                for _ in range(N + 1)]
        col_prior = np.ascontiguousarray(
            crp_prior[np.fromiter(sizes.values(), dtype=np.int64)])
        hint = _lib.hints_from_matrix(ll, col_prior)
        outs, used = [], []
        for h, in_order, lane in ((None, False, None),
                ((hint, col_prior), False, None),
                ((hint, col_prior), True, None),
                ((hint, col_prior), True, '0'),
                ((hint, col_prior), seed % 2 == 0, '0.05'),
                ((hint, col_prior), seed % 2 == 1, '0.3')):
            # (lane: BNPC_SWEEP_LANE - the lane of the cells decided from
            # their records switched off, or handing a share of its cells
            # over to the general iteration with their uniforms drawn)
            if lane is None:
                monkeypatch.delenv('BNPC_SWEEP_LANE', raising=False)
            else:
                monkeypatch.setenv('BNPC_SWEEP_LANE', lane)
            np.random.seed(seed)
            got = native_gibbs(ll, post_new, crp_prior, assignment, sizes,
                new_columns, hint=h, used=used, in_order=in_order)
            outs.append((got[0], list(got[1].items()), got[2],
                np.random.random(2)))
            if h is not None:
                in_lane[lane] = in_lane.get(lane, 0) + native_gibbs.last[3]
        for other in outs[1:]:
            assert np.array_equal(outs[0][0], other[0]), seed
            assert outs[0][1] == other[1] and outs[0][2] == other[2], seed
            assert np.array_equal(outs[0][3], other[3]), seed
        assert used[1] == used[2] == used[3] == used[4] == used[5], seed
        if not first_sweep:
            decided += used[1]
            pairs += native_gibbs.last[1]
            triples += native_gibbs.last[2]
            cells += N
        else:
            assert got[2] > 0 and used[1] > 0, (seed, got[2], used[1])
    # (a cell whose favourite cluster has died since the launch is scanned)
    assert decided > 0.7 * cells, (decided, cells)
    assert pairs > 0.1 * cells and triples > 0.05 * cells, (pairs, triples)
    assert in_lane['0'] == 0 and 0 < in_lane['0.3'] < in_lane[None], in_lane


# ------------------------------------------------ native split / merge moves
def test_np_sum_restated_is_numpys():
    """bnpc_sm_move sums the per-mutation terms of the likelihood ratio the
    way np.sum does (pairwise inside runs of the iterator's 8192-element
    buffer): the checker hook against NumPy over the sizes around every
    change of regime."""
    rng = np.random.RandomState(0)
    sizes = list(range(1, 300)) + [511, 512, 513, 1000, 1023, 1024, 1025,
        2000, 4999, 5000, 8191, 8192, 8193, 16384, 16385, 20000, 31600]
    for n in sizes:
        for _ in range(3):
            a = rng.standard_normal(n) * 10 ** rng.uniform(-3, 3, n)
            assert _lib.np_sum(a) == np.sum(a), n
            assert _lib.np_sum(-np.abs(a)) == np.sum(-np.abs(a)), n


def test_native_move_proposals_equal_the_bindings():
    """The proposal part of bnpc_sm_move (np.random.choice with p, without
    replacement with p, permutation(n)[:2], randint) against the binding's
    NumPy statements on the same stream: cells with the anchors in place, the
    size terms to the bit, the stream position - over random cluster tables
    (dict order != id order, singletons, repeated first picks of a merge)."""
    from bnpc_amd import hostkernels, model as P
    table = hostkernels.table()
    if table is None:
        pytest.skip('no host kernel table on this stack')
    rng = np.random.RandomState(1)

    class Probe(P.CRP):
        def __init__(self):
            pass
    repeats = 0
    for trial in range(200):
        N = int(rng.choice([5, 12, 60, 400, 3000]))
        K0 = int(rng.randint(1, max(2, min(N, 40))))
        lab = rng.randint(0, K0, N) if trial % 7 else np.arange(N)
        names = rng.permutation(N)
        m = Probe()
        m.assignment = np.ascontiguousarray(names[lab], dtype=np.int64)
        order = rng.permutation(np.unique(m.assignment))
        m.cells_per_cluster = {int(c): int((m.assignment == c).sum())
            for c in order}
        ids = np.fromiter(m.cells_per_cluster.keys(), dtype=np.int64)
        sizes = np.fromiter(m.cells_per_cluster.values(), dtype=np.int64)
        for move in (0, 1):
            if (move == 1 and ids.size < 2) \
                    or (move == 0 and (sizes == 1).all()):
                continue
            seed = int(rng.randint(1 << 30))
            np.random.seed(seed)
            if move == 0:
                _, cells, (ltrans, others) = m._propose_split()
                want = (cells, 0, float(ltrans[0]))
            else:
                _, _, cells_j, cells, size = m._propose_merge()
                want = (cells, cells.size - cells_j.size, float(size))
                np.random.seed(seed)
                u = np.random.random_sample(2)
                cdf = np.cumsum((1 / sizes) / (1 / sizes).sum())
                pick = (cdf / cdf[-1]).searchsorted(u, side='right')
                repeats += pick[0] == pick[1]
                np.random.seed(seed)
                m._propose_merge()
            after = np.random.random()
            np.random.seed(seed)
            got = _lib.move_propose(table, move, ids, sizes, m.assignment)
            assert got is not None
            assert np.array_equal(got[0], want[0]), (trial, move)
            assert got[3] == want[2], (trial, move)
            assert np.random.random() == after, (trial, move)
            if move == 0:
                assert np.array_equal(got[4], others)
            else:
                assert got[1] == want[1]
    assert repeats > 5


def test_quick_picks_equal_the_full_arithmetic_around_every_boundary():
    """The native loops first try to take a pick from ONE exp() - the
    uniform against the interval ends that exp() implies - and fall back to
    the reference's arithmetic when the uniform is within 1e-11 of an end
    (DESIGN.md 5a).  Here both variants are driven with uniforms placed ON
    the ends and 1e-16 ... 1e-3 either side of them, for random gaps,
    cluster counts and positions: wherever the quick variant decides, it
    decides what the full arithmetic decides; it declines only near an end;
    far from the ends it always decides."""
    lib = _lib.load()
    rng = np.random.RandomState(11)
    pick = C.c_int64(0)

    def pair(quick, d2, A, top, sec, u):
        _lib.check(lib.bnpc_pair_pick(quick, d2, A, top, sec, u,
            C.byref(pick)), 'pair_pick')
        return pick.value

    def two(quick, p0, p1, u):
        _lib.check(lib.bnpc_two_way_pick(quick, p0, p1, u, C.byref(pick)),
            'two_way_pick')
        return pick.value

    offsets = [0.0] + [s * 10.0 ** e for e in range(-16, -2) for s in (1, -1)]
    floor = 1e-15
    declined = decided = declined32 = decided32 = 0
    for trial in range(4000):
        A = int(rng.randint(2, 65))
        top, sec = (int(x) for x in rng.permutation(A + 1)[:2])
        d2 = -float(rng.choice([0.0, 1e-9, 0.3, 2.0, 9.0, 20.0, 33.0, 36.0])
            * rng.random_sample())
        x = np.exp(d2)
        p_top, p_sec = 1 / (1 + x), x / (1 + x)
        m1, m2 = min(top, sec), max(top, sec)
        e1, e2 = (p_top, p_sec) if m1 == top else (p_sec, p_top)
        total = 1 + (A - 1) * floor
        ends = [m1 * floor, m1 * floor + e1,
            m1 * floor + e1 + (m2 - m1 - 1) * floor]
        ends.append(ends[-1] + e2)
        for end in ends:
            for off in offsets:
                u = end / total + off
                if not 0.0 <= u < 1.0:
                    continue
                full = pair(0, d2, A, top, sec, u)
                quick = pair(1, d2, A, top, sec, u)
                if quick < 0:
                    declined += 1
                    # (near an end - of this interval or, for an interval
                    # narrower than the offset, of its other side)
                    assert abs(off) < 1e-10 or min(e1, e2) < 2 * abs(off) \
                        + 1e-10, (d2, A, top, sec, off)
                else:
                    decided += 1
                    assert quick == full, (d2, A, top, sec, u, off)
                # the form the sweep loop uses with a hint record: the
                # exponential as float32, a band of 1e-6
                loop = pair(2, d2, A, top, sec, u)
                if loop < 0:
                    declined32 += 1
                    assert abs(off) < 1e-5 or min(e1, e2) < 2 * abs(off) \
                        + 1e-5, (d2, A, top, sec, off)
                else:
                    decided32 += 1
                    assert loop == full, (d2, A, top, sec, u, off)
        for _ in range(4):              # anywhere: decided, and the same
            u = rng.random_sample()
            assert pair(1, d2, A, top, sec, u) == pair(0, d2, A, top, sec, u)
            loop = pair(2, d2, A, top, sec, u)
            assert loop < 0 or loop == pair(0, d2, A, top, sec, u)
    assert decided > 50000 and declined > 50000
    assert decided32 > 30000 and declined32 > 70000

    declined = decided = 0
    for trial in range(6000):
        gap = float(rng.choice([0.0, 1e-9, 0.5, 3.0, 15.0, 39.9])
            * rng.random_sample())
        base = -rng.random_sample() * 800
        p0, p1 = (base, base - gap) if trial % 2 else (base - gap, base)
        x = np.exp(-gap)
        q0 = 1 / (1 + x) if p0 >= p1 else x / (1 + x)
        for off in offsets:
            u = q0 + off
            if not 0.0 <= u < 1.0:
                continue
            full = two(0, p0, p1, u)
            quick = two(1, p0, p1, u)
            if quick < 0:
                declined += 1
                assert abs(off) < 1e-10
            else:
                decided += 1
                assert quick == full, (p0, p1, u, off)
        u = rng.random_sample()
        assert two(1, p0, p1, u) == two(0, p0, p1, u)
    assert decided > 20000 and declined > 20000


def test_unscored_restricted_scan_equals_the_scored_one():
    """bnpc_rg_scan without a log-probability output (the intermediate scans
    of a move: picks from one exp() per cell where that is safe) against the
    same scan with it: same assignments, same stream position."""
    lib = _lib.load()
    for seed in range(300):
        rng = np.random.RandomState(9000 + seed)
        S = int(rng.choice([3, 40, 333, 2000]))
        spread = rng.choice([0.1, 3.0, 50.0, 90.0, 900.0])
        ll = np.ascontiguousarray(-rng.random_sample((S, 2)) * spread)
        start = rng.randint(0, 2, S).astype(np.int64)
        outs = []
        for scored in (True, False):
            rg = start.copy()
            np.random.seed(seed)
            st, extra = _lib.rng_export()
            out = C.c_double(0)
            _lib.check(lib.bnpc_rg_scan(C.byref(st), 0, S, _lib.ptr(ll, f64),
                3.7, _lib.ptr(rg, i64), None,
                C.byref(out) if scored else None), 'rg_scan')
            outs.append((rg, int(st.pos), bytes(st.key)))
        assert np.array_equal(outs[0][0], outs[1][0]), seed
        assert outs[0][1:] == outs[1][1:], seed


def test_bulk_draws_equal_numpys_at_every_block_boundary():
    """The vectorised stream (512-bit state refill, tempering + conversion of
    8 doubles at a time, accepted interval candidates by register compress)
    against NumPy's own randint / uniform / random from the same state: batch
    lengths around the vector widths and the 624-word block, starting at even
    and odd word positions of the block, several interval widths; values and
    the state left behind."""
    bad = []
    for n_sd in (1, 2, 3, 5, 8, 100, 257):
        for M in (1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 311, 312, 313, 623,
                624, 625, 1000, 2500):
            for pre in (0, 1, 300, 611, 623):
                np.random.seed(n_sd * 31 + M)
                if pre:
                    np.random.random_sample(pre // 2)
                if pre % 2:
                    np.random.randint(0, 2 ** 31)   # an odd word offset
                start = np.random.get_state()
                got = _lib.mh_draws(2, M, n_sd)
                after = np.random.get_state()
                np.random.set_state(start)
                ok = True
                for g in range(2):
                    idx = np.random.randint(0, n_sd, M)
                    U = np.random.uniform(size=M)
                    u = np.random.random(M)
                    ok &= np.array_equal(idx, got[0][g]) \
                        and np.array_equal(U, got[1][g]) \
                        and np.array_equal(u, got[2][g])
                now = np.random.get_state()
                ok &= now[2] == after[2] and np.array_equal(now[1], after[1])
                if not ok:
                    bad.append((n_sd, M, pre))
    assert not bad, bad[:10]


def test_quick_pick_among_three_equals_the_scan_around_every_boundary():
    """A cell with three candidate clusters (bnpc_triple_pick): the pick from
    two exp() against the scan's own arithmetic over the whole row, with the
    uniform ON every interval end and 1e-16 ... 1e-3 either side, for random
    gaps (ties, a third entry beyond the 60-below rule, beyond the floor),
    cluster counts and positions."""
    lib = _lib.load()
    rng = np.random.RandomState(12)
    pick = C.c_int64(0)

    def triple(quick, q, a, A, u):
        _lib.check(lib.bnpc_triple_pick(quick, q.ctypes.data, a.ctypes.data, A,
            u, C.byref(pick)), 'triple_pick')
        return pick.value

    offsets = [0.0] + [s * 10.0 ** e for e in range(-16, -2) for s in (1, -1)]
    floor = 1e-15
    declined = decided = decided32 = 0
    for trial in range(3000):
        A = int(rng.randint(2, 65))
        a = np.ascontiguousarray(rng.permutation(A + 1)[:3], dtype=np.int64)
        gaps = -rng.choice([0.0, 1e-9, 0.3, 2.0, 9.0, 30.0, 45.0, 70.0, 95.0],
            2) * rng.random_sample(2)
        if trial % 5 == 0:
            gaps[1] = gaps[0]                   # a tie among the other two
        q = np.ascontiguousarray(
            rng.permutation(np.array([0.0, gaps[0], gaps[0] + gaps[1]]))
            - rng.random_sample() * 300)
        x = np.exp(q - q.max())
        p = x / x.sum()
        total = 1 + (A - 2) * floor
        ends, edge, prev = [], 0.0, -1
        for i in np.argsort(a):
            lo = edge + (a[i] - prev - 1) * floor
            ends += [lo, lo + p[i]]
            edge, prev = lo + p[i], a[i]
        for end in ends:
            for off in offsets:
                u = end / total + off
                if not 0.0 <= u < 1.0:
                    continue
                quick = triple(1, q, a, A, u)
                if quick < 0:
                    declined += 1
                else:
                    decided += 1
                    assert quick == triple(0, q, a, A, u), (q, a, A, u, off)
                # (the loop's form: float32 weights, a band of 1e-6)
                loop = triple(2, q, a, A, u)
                if loop >= 0:
                    decided32 += 1
                    assert loop == triple(0, q, a, A, u), (q, a, A, u, off)
                else:
                    assert abs(off) < 1e-5 or p.min() < 2 * abs(off) + 1e-5, \
                        (q, a, A, u, off)
        for _ in range(3):
            u = rng.random_sample()
            quick = triple(1, q, a, A, u)
            assert quick < 0 or quick == triple(0, q, a, A, u), (q, a, A, u)
            loop = triple(2, q, a, A, u)
            assert loop < 0 or loop == triple(0, q, a, A, u), (q, a, A, u)
    assert decided > 50000 and declined > 50000 and decided32 > 30000


# ------------------------------------------------------ pieces of the native step
def test_native_step_scalars_pass_their_start_up_comparison():
    """np.random.gamma on the stream and the Gamma log-density of the
    concentration parameter as bnpc_chain_step restates them are NumPy's /
    SciPy's bits on this stack (else whole steps are not made natively)."""
    assert P._native_step_allowed()
    table = P._native_kernels()
    from bnpc_amd import fastdist
    rng = np.random.RandomState(5)
    for _ in range(200):
        a = float(rng.uniform(1.5, 300))
        x = float(rng.uniform(1 + 1e-15, 2 * a))
        assert _lib.gamma_logpdf_scalar(table, x, a, 1) \
            == float(fastdist.gamma_logpdf(x, a, 1))
    assert _lib.gamma_logpdf_scalar(table, 0.5, 3.0, 1) is None   # x <= loc
    for seed in range(40):
        shape = float(rng.choice([.3, 1., 1.7, 40.2, 5000.7]))
        scale = float(rng.uniform(.05, 9))
        np.random.seed(seed)
        if seed % 3 == 0:
            np.random.normal()          # a cached Gaussian on entry
        state = np.random.get_state()
        want = (np.random.gamma(shape, scale), np.random.random())
        np.random.set_state(state)
        got = (_lib.gamma(shape, scale), np.random.random())
        assert want == got, (seed, shape, scale)


@pytest.mark.parametrize('neg_alpha', [True, False])
def test_native_alpha_update_is_the_bindings(neg_alpha):
    """CRP.update_DP_alpha + init_DP_prior (libs/CRP.py:386-410, 191-194) as
    the native step makes them (bnpc_chain_update_alpha): the same DP_a, the
    same N + 2 prior vector bit for bit, the stream and the cached Gaussian
    left where NumPy leaves them - over 300 updates in a row."""
    import ctypes as C
    N, M = 777, 5
    data = np.zeros((N, M))
    model = P.CRP(data, DP_alpha=[-1, -1] if neg_alpha else [3.5, 0.25],
        param_beta=[.25, .25], FN_error=0.1, FP_error=0.01)
    table = P._native_kernels()
    nat = _lib.NativeChain(N, M)
    st = nat.st
    st.dpa_shape = float(model.DP_a_gamma[0])
    st.dpa_rate = float(model.DP_a_gamma[1])
    model.init_DP_prior()
    prior = model.CRP_prior.copy()
    st.crp_prior = _lib.ptr(prior)
    st.DP_a = model.DP_a
    rng = np.random.RandomState(3)
    np.random.seed(99)
    for it in range(300):
        K = int(rng.randint(1, 60))
        model.cells_per_cluster = {i: 1 for i in range(K)}
        st.K = K
        state = np.random.get_state()
        model.update_DP_alpha()
        after = np.random.get_state()
        np.random.set_state(state)
        with _lib.NumpyGaussStream() as (r, gauss):
            st.gauss = gauss
            _lib.check(nat._lib.bnpc_chain_update_alpha(C.addressof(table), r,
                C.byref(st)), 'chain_update_alpha')
        now = np.random.get_state()
        assert st.DP_a == model.DP_a, it
        assert np.array_equal(prior, model.CRP_prior), it
        assert now[2:] == after[2:] and np.array_equal(now[1], after[1]), it
    nat.close()


def test_live_stream_pointers_follow_a_replaced_bit_generator():
    """np.random.set_bit_generator gives the global stream another state
    block: the cached in-place pointers (rng_live, gauss_live) are looked up
    again, native draws keep equalling NumPy's."""
    if not hasattr(np.random, 'set_bit_generator'):
        pytest.skip('NumPy without set_bit_generator')
    old = np.random.get_bit_generator()
    try:
        np.random.seed(1)
        _lib.beta(np.array([2.5]), np.array([3.5]))     # pointers cached
        first = _lib.rng_live()
        np.random.set_bit_generator(np.random.MT19937(123))
        for seed in (5, 6):
            np.random.seed(seed)
            if seed == 6:
                np.random.normal()
            state = np.random.get_state()
            want = (np.random.beta(2.5, 3.5), np.random.random())
            np.random.set_state(state)
            got = (_lib.beta(np.array([2.5]), np.array([3.5]))[0],
                np.random.random())
            assert want == got
        assert _lib.rng_live() is not first
    finally:
        np.random.set_bit_generator(old)
    np.random.seed(3)
    state = np.random.get_state()
    want = np.random.beta(.5, .25)
    np.random.set_state(state)
    assert _lib.beta(np.array([.5]), np.array([.25]))[0] == want


def test_native_moves_pass_their_start_up_comparison():
    """The run-time gate of bnpc_sm_move: np.sum's order and the
    legacy proposals as the library restates them equal NumPy's on this
    stack, and the comparison leaves the stream where it was."""
    np.random.seed(8)
    state = np.random.get_state()
    P._MOVES_OK.clear()
    assert P._native_moves_allowed()
    now = np.random.get_state()
    assert now[2:] == state[2:] and np.array_equal(now[1], state[1])
