"""TEST INFRASTRUCTURE - CPU oracle of the posterior estimator (SURVEY.md
section 8(f) rank 4), a NumPy/SciPy restatement of
/root/reference/libs/utils.py: get_dist :90-97, _get_MPEAR :100-130,
_calc_MPEAR :133-145, get_mean_hierarchy_assignment :148-192,
_concat_chain_results :206-223, _get_latents_posterior_chain :226-244.
Pinned by tests/golden/posterior.npz (captured from the imported reference).
Third-party pieces, called from the installed SciPy as the reference does:
scipy.spatial.distance.pdist, scipy.cluster.hierarchy.linkage / cut_tree,
scipy.special.binom; bottleneck.move_std (window 2) is restated as "all cells
of the group share one label".
"""
import numpy as np
from scipy.cluster.hierarchy import cut_tree, linkage
from scipy.spatial.distance import pdist
from scipy.special import binom

EPSILON = np.finfo(np.float64).resolution


def get_dist(assignments):
    """Mean posterior co-clustering distance, condensed (utils.py:90-97)."""
    steps, cells = assignments.shape
    differ = np.zeros(cells * (cells - 1) // 2, dtype=np.int32)
    for sample in assignments:
        differ += pdist(np.stack([sample, sample]).T, 'hamming') \
            .astype(np.int32)
    return differ / steps


def calc_MPEAR(pi, c):
    """Fritsch & Ickstadt (2009) eq. 13 (utils.py:133-145)."""
    same = 1 - pdist(np.stack([c, c]).T, 'hamming')
    I_sum = same.sum()
    pi_sum = pi.sum()
    index = (same * pi).sum()
    expected = (I_sum * pi_sum) / binom(c.size, 2)
    return (index - expected) / (.5 * (I_sum + pi_sum) - expected)


def get_MPEAR(assignments, dist=None):
    """utils.py:100-130"""
    if dist is None:
        dist = get_dist(assignments)
    sim = 1 - dist
    Z = linkage(dist, method='ward')
    big = [int((np.unique(a, return_counts=True)[1] > 2).sum())
        for a in assignments]
    avg = np.mean(big)
    n_range = np.arange(max(2, avg * 0.2),
        min(avg * 2.5, assignments.shape[1]), dtype=int)
    best, best_score = None, -np.inf
    for n in n_range:
        clusters = cut_tree(Z, n_clusters=n).flatten()
        score = calc_MPEAR(sim, clusters)
        if score > best_score:
            best, best_score = clusters, score
    return best


def mean_hierarchy_assignment(assignments, params_full, dist=None):
    """utils.py:148-192: MPEAR clustering + averaged cluster genotypes."""
    steps = assignments.shape[0]
    assign = get_MPEAR(assignments, dist)
    clusters = np.unique(assign)
    params = np.zeros((clusters.size, params_full.shape[2]))
    for i, cluster in enumerate(clusters):
        member = assign == cluster
        cells = np.nonzero(member)[0]
        other = np.nonzero(~member)[0]
        sub = assignments[:, cells]
        if cells.size == 1:
            together = np.ones(steps, dtype=bool)
        else:
            together = (sub == sub[:, :1]).all(axis=1)
        ids = np.array([np.argmax(np.bincount(row)) for row in sub])
        others = assignments[:, other]
        alone = np.array([ids[s] not in others[s] for s in range(steps)])
        if together.any():
            pick = together & alone if (together & alone).any() else together
            chosen = np.argwhere(pick).flatten()
            for s in chosen:
                present = np.append(np.unique(others[s]), ids[s])
                rank = np.argwhere(np.sort(present) == ids[s])[0][0]
                params[i] += params_full[s][rank]
            params[i] /= chosen.size
        else:
            for s, sample in enumerate(assignments):
                all_ids = np.unique(sample)
                cid, cnt = np.unique(sample[cells], return_counts=True)
                rows = np.argwhere(np.isin(all_ids, cid)).flatten()
                params[i] += np.dot(cnt, params_full[s][rows])
            params[i] /= steps * cells.size
    return assign, params[assign].T          # genotypes: mutations x cells


def concat_chain_results(results):
    """utils.py:206-223"""
    cat = {k: np.concatenate([r[k][r['burn_in']:] for r in results])
        for k in ('assignments', 'DP_alpha', 'ML', 'MAP', 'FN', 'FP')}
    params = [r['params'] for r in results]
    width = max(p.shape[1] for p in params)
    params = [np.pad(p, [(0, 0), (0, width - p.shape[1]), (0, 0)])
        for p in params]
    cat['params'] = np.concatenate(params)
    cat['burn_in'] = 0
    return cat


def latents_posterior(results, data):
    """Default (all chains pooled) path of utils.py:195-244."""
    res = concat_chain_results(results)
    assign, geno = mean_hierarchy_assignment(res['assignments'], res['params'])
    called = geno.T.round()
    FN_geno = (((called == 1) & (data == 0)).sum() + EPSILON) \
        / (called.sum() + EPSILON)
    FP_geno = (((called == 0) & (data == 1)).sum() + EPSILON) \
        / ((1 - called).sum() + EPSILON)
    avg = lambda v: (np.mean(v), np.std(v))
    return {'a': avg(res['DP_alpha']), 'assignment': assign,
        'genotypes': geno, 'FN': avg(res['FN']), 'FP': avg(res['FP']),
        'FN_geno': FN_geno, 'FP_geno': FP_geno}
