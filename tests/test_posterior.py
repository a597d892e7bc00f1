"""Posterior estimator: oracle and product
against golden vectors captured from the reference
(tests/golden/make_posterior_golden.py); the GPU co-clustering kernel against
the oracle's exact integer counts."""
import os

import numpy as np
import pytest
from scipy.spatial.distance import pdist

from oracle import posterior_numpy as Q
from bnpc_amd import _lib, postproc


def decode(codes):
    x = codes.astype(np.float64)
    x[codes == 3] = np.nan
    return x


@pytest.fixture(scope='module')
def G(golden_dir):
    g = np.load(os.path.join(golden_dir, 'posterior.npz'))
    results = []
    for i in range(2):
        r = {k: g[f'r{i}_{k}'] for k in ('assignments', 'params', 'DP_alpha',
            'FN', 'FP', 'ML', 'MAP')}
        r['burn_in'] = int(g[f'r{i}_burn_in'])
        results.append(r)
    return g, results, decode(g['data'])


def differ_counts(assignments):
    out = np.zeros(assignments.shape[1] * (assignments.shape[1] - 1) // 2,
        dtype=np.int32)
    for a in assignments:
        out += pdist(np.stack([a, a]).T, 'hamming').astype(np.int32)
    return out


def test_oracle_matches_reference(G):
    g, results, data = G
    a0 = results[0]['assignments'][results[0]['burn_in']:]
    dist = Q.get_dist(a0)
    assert np.array_equal(dist, g['dist0'])
    labels = Q.get_MPEAR(a0)
    assert np.array_equal(labels, g['mpear0'])
    assert Q.calc_MPEAR(1 - dist, labels) == float(g['mpear0_score'])
    lat = Q.latents_posterior(results, data)
    assert np.array_equal(lat['assignment'], g['mean0_assignment'])
    assert np.array_equal(lat['genotypes'], g['mean0_genotypes'])
    np.testing.assert_allclose(lat['a'], g['mean0_a'], rtol=1e-14)
    np.testing.assert_allclose(lat['FN_geno'], g['mean0_FN_geno'], rtol=1e-14)
    np.testing.assert_allclose(lat['FP_geno'], g['mean0_FP_geno'], rtol=1e-14)


def test_point_estimates_and_psrf_match_reference(G):
    g, results, data = G
    for est in ('ML', 'MAP'):
        inf = postproc.point_estimate(postproc.best_chain(results, est), est,
            data)
        assert inf['step'] == int(g[f'{est}_step'])
        assert np.array_equal(inf['assignment'], g[f'{est}_assignment'])
        assert np.array_equal(inf['genotypes'].T, g[f'{est}_genotypes'])
        np.testing.assert_allclose(inf['FN_geno'], g[f'{est}_FN_geno'],
            rtol=1e-14)
    psrf = postproc.get_lugsail_batch_means_est(
        [(r['ML'], r['burn_in']) for r in results])
    np.testing.assert_allclose(psrf, g['psrf'], rtol=1e-12)


def test_product_posterior_host_logic(G, monkeypatch):
    """The product's estimator with the device kernel replaced by the exact
    NumPy counts: the reference's clustering and genotypes."""
    g, results, data = G
    from fake_device import FakePosterior
    monkeypatch.setattr(_lib, 'codist',
        lambda a, device=None: differ_counts(np.asarray(a)))
    monkeypatch.setattr(_lib, 'Posterior', FakePosterior)
    inf = postproc.posterior_estimate(results, data)
    assert np.array_equal(inf['assignment'], g['mean0_assignment'])
    assert np.array_equal(inf['genotypes'].T, g['mean0_genotypes'])
    np.testing.assert_allclose(inf['FN'], g['mean0_FN'], rtol=1e-14)
    np.testing.assert_allclose(inf['FN_geno'], g['mean0_FN_geno'], rtol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize('S,N,K', [(1, 2, 2), (33, 63, 4), (64, 64, 3),
    (100, 65, 7), (37, 300, 12), (250, 1000, 10)])
def test_codist_kernel_is_exact(S, N, K):
    rng = np.random.RandomState(S * 1000 + N)
    a = rng.randint(0, K, size=(S, N))
    a[:, : N // 3] = a[:, :1]            # a block of always-together cells
    got = _lib.codist(a)
    assert got.dtype == np.int32
    assert np.array_equal(got, differ_counts(a))
    assert np.array_equal(postproc.get_dist(a), Q.get_dist(a))


def test_cut_tree_labels_equal_scipys():
    """postproc.cut_tree_labels against scipy.cluster.hierarchy.cut_tree on
    Ward / average / single trees of random points incl. duplicated points
    (tied merges) and every cluster count from 1 to N - 1 (asked for N
    clusters TOGETHER with other counts, SciPy returns zeros - its row 0 quirk;
    alone it returns the identity, as here; the estimator's candidate range
    ends below N either way)."""
    from scipy.cluster.hierarchy import cut_tree, linkage
    rng = np.random.RandomState(4)
    for N, method in ((2, 'ward'), (7, 'ward'), (60, 'ward'), (60, 'average'),
            (133, 'single'), (300, 'ward')):
        pts = rng.normal(size=(N, 3))
        pts[N // 2:] = pts[: N - N // 2] + (rng.random_sample((N - N // 2, 3))
            < 0.5) * 0.1                    # near / exact duplicates
        tree = linkage(pts, method=method)
        counts = np.arange(1, N) if N <= 133 else \
            np.array([1, 2, 3, 10, 50, 298, 299])
        want = cut_tree(tree, n_clusters=counts)
        got = postproc.cut_tree_labels(tree, counts)
        assert np.array_equal(got, want), (N, method)
        assert np.array_equal(postproc.cut_tree_labels(tree, [N]),
            cut_tree(tree, n_clusters=[N]))


def test_mpear_from_integers_picks_the_references_cut(G):
    """The scores the pipeline computes from exact integers (label counts,
    the sum of the pair counts, per candidate the sum over its same-label
    pairs) against the reference-order float evaluation (calc_MPEAR) on the
    golden samples: equal to 1e-12, same winning cut."""
    from fake_device import FakePosterior
    from scipy.cluster.hierarchy import cut_tree, linkage
    g, results, data = G
    a0 = results[0]['assignments'][results[0]['burn_in']:]
    post = FakePosterior(a0)
    dist = post.dist()
    assert np.array_equal(dist, g['dist0'])
    tree = linkage(dist, method='ward')
    cand = np.arange(2, 12)
    labels = np.ascontiguousarray(cut_tree(tree, n_clusters=cand).T)
    got = postproc.mpear_scores(post.mpear_sums(labels), labels,
        post.differ_sum, a0.shape[0])
    want = [postproc.calc_MPEAR(1 - dist, lab) for lab in labels]
    np.testing.assert_allclose(got, want, rtol=1e-12)
    assert int(np.argmax(got)) == int(np.argmax(want))
    assert np.array_equal(postproc.get_MPEAR(a0, dist=dist), g['mpear0'])


@pytest.mark.gpu
@pytest.mark.parametrize('S,N,K,C', [(5, 2, 2, 1), (33, 63, 4, 3),
    (64, 130, 3, 40), (100, 65, 7, 70), (37, 300, 12, 130), (120, 1000, 10, 29)])
def test_posterior_pipeline_on_device_is_exact(S, N, K, C):
    """bnpc_post: the pair counts kept on the device equal the oracle's, their
    sum and the mean distance (divided on the device) equal NumPy's bit for
    bit, and the per-candidate sums over same-label pairs (k_mpear_sums: 1,
    2 or 3 candidate chunks, tiles on and off the diagonal, ragged edges)
    equal the direct integer evaluation."""
    from fake_device import FakePosterior
    rng = np.random.RandomState(S * 1000 + N)
    a = rng.randint(0, K, size=(S, N))
    a[:, : N // 3] = a[:, :1]
    post, ref = _lib.Posterior(a), FakePosterior(a)
    try:
        assert np.array_equal(post.differ(), ref.differ())
        assert post.differ_sum == ref.differ_sum
        assert np.array_equal(post.dist(), ref.dist())
        labels = rng.randint(0, rng.randint(1, 40, size=(C, 1)), size=(C, N))
        assert np.array_equal(post.mpear_sums(labels), ref.mpear_sums(labels))
    finally:
        post.close()


@pytest.mark.gpu
@pytest.mark.parametrize('S,N,K', [(5, 2, 2), (7, 3, 2), (33, 63, 4),
    (64, 130, 3), (100, 65, 7), (37, 300, 12), (120, 1000, 10), (400, 2500, 9)])
def test_ward_linkage_on_device_is_scipys(S, N, K):
    """bnpc_post_ward + the binding's sort / relabel against
    scipy.cluster.hierarchy.linkage(dist, 'ward') on mean co-clustering
    distances - discrete values k / S with MANY exact ties, blocks of cells
    at distance 0: the same linkage matrix bit for bit (children, heights,
    sizes), hence the same tree cuts."""
    from scipy.cluster.hierarchy import linkage
    rng = np.random.RandomState(S * 1000 + N)
    a = rng.randint(0, K, size=(S, N))
    a[:, : N // 3] = a[:, :1]               # always together: distance 0
    noise = rng.random_sample((S, N)) < 0.3
    base = rng.randint(0, K, N)
    a = np.where(noise, a, base[None, :])
    a[:, : N // 3] = a[:, :1]
    post = _lib.Posterior(a)
    try:
        want = linkage(post.dist(), method='ward')
        got = post.ward()
        assert np.array_equal(got, want)
    finally:
        post.close()


@pytest.mark.gpu
def test_product_posterior_on_device(G):
    g, results, data = G
    inf = postproc.posterior_estimate(results, data)
    assert np.array_equal(inf['assignment'], g['mean0_assignment'])
    assert np.array_equal(inf['genotypes'].T, g['mean0_genotypes'])


@pytest.mark.gpu
def test_codist_full_size_properties():
    """5000 cells x 400 samples (the config-3 cell count): symmetry-free
    invariants of the condensed counts + a sampled exact comparison."""
    rng = np.random.RandomState(0)
    S, N = 400, 5000
    base = rng.randint(0, 10, N)
    a = np.tile(base, (S, 1))
    flip = rng.random_sample((S, N)) < 0.05
    a[flip] = rng.randint(0, 10, flip.sum())
    d = _lib.codist(a)
    assert d.size == N * (N - 1) // 2 and d.min() >= 0 and d.max() <= S
    # relabelling the clusters of any sample changes nothing
    b = (a + 3) % 10
    assert np.array_equal(_lib.codist(b), d)
    # sampled rows against the direct count
    for i in rng.choice(N - 1, 5, replace=False):
        row = (a[:, [i]] != a[:, i + 1:]).sum(axis=0)
        start = i * (2 * N - i - 1) // 2
        assert np.array_equal(d[start:start + N - i - 1], row)
    # a permutation of the samples changes nothing
    assert np.array_equal(_lib.codist(a[rng.permutation(S)]), d)


def test_mpear_cut_choice_follows_the_references_loop():
    """utils.py:116-130 keeps the FIRST of the largest scores, skips NaN
    scores (`score > best` is False for NaN) and returns None when no
    candidate passes - the vectorised choice does the same."""
    from bnpc_amd import postproc
    cuts = np.arange(12).reshape(4, 3)
    pick = postproc._first_maximum
    assert np.array_equal(pick(cuts, [0.1, 0.7, 0.7]), cuts[:, 1])
    assert np.array_equal(pick(cuts, [np.nan, 0.2, 0.1]), cuts[:, 1])
    assert np.array_equal(pick(cuts, [0.3, np.nan, 0.3]), cuts[:, 0])
    assert pick(cuts, [np.nan, np.nan, np.nan]) is None
    assert pick(cuts, [-np.inf, -np.inf, -np.inf]) is None
    # no cluster of more than two cells in any sample: the candidate range
    # np.arange(2, 0) is empty, the reference's loop does not run
    alone = np.tile(np.arange(6), (5, 1))
    assert postproc.get_MPEAR(alone, dist=np.full(6 * 5 // 2, 0.5)) is None


@pytest.mark.gpu
def test_ward_routes_give_the_same_clustering(monkeypatch):
    """BNPC_WARD_DEVICE: the linkage on the device as a replayed graph (the
    default), with plain launches ('plain') and SciPy's own routine on the
    fetched distances ('0') - the same MPEAR clustering."""
    from bnpc_amd import postproc
    rng = np.random.RandomState(4)
    base = rng.randint(0, 6, 900)
    a = np.tile(base, (120, 1)).astype(np.int32)
    flip = rng.random_sample(a.shape) < 0.04
    a[flip] = rng.randint(0, 6, flip.sum())
    got = {}
    for route in ('1', 'plain', '0'):
        monkeypatch.setenv('BNPC_WARD_DEVICE', route)
        got[route] = postproc.get_MPEAR(a)
    assert np.array_equal(got['1'], got['plain'])
    assert np.array_equal(got['1'], got['0'])
