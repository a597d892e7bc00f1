"""Post-processing the driver and the CLI need.

Restated from /root/reference/libs/utils.py: the lugsail batch-means PSRF
(:427-467, used by the -ls termination mode and the run summary), the ML / MAP
point estimates (:248-282) and the posterior estimator (:90-244: mean
co-clustering distance, MPEAR-selected Ward clustering, averaged cluster
genotypes).  The O(samples x cells^2) co-clustering distance - the next
data-parallel kernel after the likelihood path (SURVEY.md section 8(f) rank
4) - and the MPEAR score of every candidate cut run on the GPU (bnpc_post:
the pair counts stay there, each candidate is one term of ONE pass over
them); Ward's linkage and the tree cuts are SciPy on the host.
Metrics (V-measure, ARI, Hamming) and tree helpers are out of scope.
"""
import numpy as np

EPSILON = np.finfo(np.float64).resolution


def _tau_lugsail(b, data, chain_mean):
    """Batch-means variance estimate with batch size b (utils.py:463-466)."""
    a = data.size // b
    batch_mean = np.mean(np.reshape(data[:a * b], (a, b)), axis=1)
    return (b / (a - 1)) * np.sum(np.square(batch_mean - chain_mean))


def get_lugsail_batch_means_est(data_in, steps=None):
    """Lugsail PSRF of Vats & Knudson (2018), eq. 5, over the ML traces of the
    chains: data_in = [(trace, burn_in), ...]  (utils.py:427-461)."""
    T_iL, s_i, n_i = [], [], []
    for trace, burn_in in data_in:
        data = trace[burn_in:steps]
        if data.size < 9:
            return np.inf
        n = data.size
        b = int(n ** 0.5)
        n_i.append(n)
        mean = np.mean(data)
        T_iL.append(2 * _tau_lugsail(b, data, mean)
            - _tau_lugsail(b // 3, data, mean))
        s_i.append(np.var(data, ddof=1))
    T_L, s, n = np.mean(T_iL), np.mean(s_i), np.round(np.mean(n_i))
    sigma_L = ((n - 1) * s + T_L) / n
    try:
        with np.errstate(divide='raise', invalid='raise'):
            return np.sqrt(sigma_L / s)
    except FloatingPointError:
        return np.inf


def point_estimate(result, est, data):
    """The sample with the best ML / MAP after burn-in (utils.py:262-282)."""
    burn_in = result['burn_in']
    k = int(np.argmax(result[est][burn_in:]))
    step = k + burn_in
    assignment = np.asarray(result['assignments'][step])
    clusters = np.unique(assignment)
    params = result['params'][k][np.arange(clusters.size)]
    row_of = {c: i for i, c in enumerate(clusters)}
    geno = params[[row_of[c] for c in assignment]]          # cells x muts
    called = geno.round()
    FN_geno = (((called == 1) & (data == 0)).sum() + EPSILON) \
        / (called.sum() + EPSILON)
    FP_geno = (((called == 0) & (data == 1)).sum() + EPSILON) \
        / ((1 - called).sum() + EPSILON)
    return {'step': step, 'a': result['DP_alpha'][step],
        'assignment': assignment.tolist(), 'genotypes': geno,
        'FN': result['FN'][step], 'FP': result['FP'][step],
        'FN_geno': FN_geno, 'FP_geno': FP_geno}


def best_chain(results, est):
    scores = [np.max(r[est][r['burn_in']:]) for r in results]
    return results[int(np.argmax(scores))]


# ---------------------------------------------------------------------------
# posterior estimator (utils.py:90-244)
# ---------------------------------------------------------------------------
def get_dist(assignments):
    """Mean posterior co-clustering distance of all cell pairs, condensed in
    pdist order (utils.py:90-97); exact integer counts from the GPU."""
    from bnpc_amd import _lib
    differ = _lib.codist(assignments)
    return differ / np.asarray(assignments).shape[0]


def _same_cluster(labels):
    """Condensed (pdist order) indicator of the pairs that share a label:
    what `1 - pdist(stack([labels, labels]).T, 'hamming')` holds
    (utils.py:137-138), built row by row from integer comparisons instead of
    a 2-column float distance."""
    labels = np.asarray(labels)
    n = labels.size
    same = np.empty(n * (n - 1) // 2, dtype=bool)
    at = 0
    for i in range(n - 1):
        np.equal(labels[i + 1:], labels[i], out=same[at:at + n - 1 - i])
        at += n - 1 - i
    return same


def calc_MPEAR(pi, labels):
    """Posterior expected adjusted Rand index of a clustering
    (Fritsch & Ickstadt 2009, eq. 13; utils.py:133-145).  Same elementwise
    products and the same NumPy reductions as the reference, so the score has
    the same bits."""
    from scipy.special import binom
    same = _same_cluster(labels)
    I_sum, pi_sum = float(same.sum()), pi.sum()
    expected = (I_sum * pi_sum) / binom(labels.size, 2)
    return ((same * pi).sum() - expected) \
        / (.5 * (I_sum + pi_sum) - expected)


def mpear_scores(same_differ, labels, differ_sum, S):
    """MPEAR (Fritsch & Ickstadt 2009, eq. 13; utils.py:133-145) of C
    candidate clusterings from exact integers: with pi = 1 - differ / S over
    the P pairs,
        I_sum  = pairs that share a label        (from the label counts)
        pi_sum = P - differ_sum / S
        index  = sum of pi over those pairs = I_sum - same_differ / S.
    labels: (C, N)."""
    from scipy.special import binom
    labels = np.asarray(labels)
    N = labels.shape[1]
    P = binom(N, 2)
    pi_sum = P - differ_sum / S
    scores = np.empty(labels.shape[0])
    for c, lab in enumerate(labels):
        n_k = np.bincount(lab).astype(np.float64)
        I_sum = float((n_k * (n_k - 1) / 2).sum())
        index = I_sum - same_differ[c] / S
        expected = (I_sum * pi_sum) / P
        scores[c] = (index - expected) / (.5 * (I_sum + pi_sum) - expected)
    return scores


def cut_tree_labels(tree, n_clusters):
    """`scipy.cluster.hierarchy.cut_tree(tree, n_clusters=...)` for several
    cluster counts: (cells, len(n_clusters)) labels.  SciPy replays all N - 1
    merges with an O(N) relabelling each (12 s at 50 000 cells).  What it
    does, restated:
      * the merges are replayed in the order of `_order_cluster_tree`: by
        height, and among equal heights in REVERSE order of a breadth-first
        walk from the root that visits right children first (every internal
        node is `insort_left`-ed as the walk meets it);
      * a merged cluster takes the smaller of the two labels and the labels
        above the larger one move down - which keeps the clusters numbered
        in the order of their smallest member at every step.
    So a cut into n clusters is: the first N - n merges of that order as a
    union-find whose roots are the smallest members, clusters ranked by
    root.  Same labels (tests: Ward / average / single trees with tied
    heights), one pass over the merges + O(N) per cut."""
    from collections import deque
    tree = np.asarray(tree)
    N = tree.shape[0] + 1
    left = tree[:, 0].astype(np.int64)
    right = tree[:, 1].astype(np.int64)
    height = tree[:, 2]
    wanted = [min(max(int(n), 1), N) for n in n_clusters]
    out = np.empty((N, len(wanted)), dtype=np.int64)
    # breadth-first visiting rank of the internal nodes (root first, right
    # child before left)
    visit = np.zeros(N - 1, dtype=np.int64)
    queue, seen = deque([2 * N - 2] if N > 1 else []), 0
    while queue:
        node = queue.popleft()
        if node >= N:
            visit[node - N] = seen
            seen += 1
            queue.append(int(right[node - N]))
            queue.append(int(left[node - N]))
    order = np.lexsort((-visit, height))        # by height, ties reversed
    # smallest member of every node
    low = np.arange(2 * N - 1, dtype=np.int64)
    for i in range(N - 1):
        low[N + i] = min(low[left[i]], low[right[i]])
    parent = np.arange(N, dtype=np.int64)       # union-find over the cells
    stops = {}
    for col, n in enumerate(wanted):
        stops.setdefault(N - n, []).append(col)

    def snapshot(cols):
        root = parent.copy()
        while True:                             # pointer jumping
            nxt = root[root]
            if np.array_equal(nxt, root):
                break
            root = nxt
        labels = np.unique(root, return_inverse=True)[1]
        for col in cols:
            out[:, col] = labels

    if 0 in stops:
        snapshot(stops[0])
    for done, i in enumerate(order, start=1):
        a, b = low[left[i]], low[right[i]]      # roots: smallest members
        parent[max(a, b)] = min(a, b)
        if done in stops:
            snapshot(stops[done])
    return out


def get_MPEAR(assignments, dist=None):
    """Ward tree on the mean distance, cut where MPEAR is largest
    (utils.py:100-130).  Product path (dist is None): the pair counts are
    made and kept on the device, the mean distance comes to the host once
    for the linkage, and ALL candidate cuts are scored in one device pass
    over the counts (bnpc_post_mpear) - the float64 similarity `1 - dist`
    and the reference's pass over it per candidate are never made.  With a
    given `dist` the scores are evaluated on the host (calc_MPEAR)."""
    from scipy.cluster.hierarchy import linkage
    from bnpc_amd import _lib
    assignments = np.asarray(assignments)
    post = None
    try:
        if dist is None:
            import os
            post = _lib.Posterior(assignments)
            tree = None
            if os.environ.get('BNPC_WARD_DEVICE', '1') != '0':   # ('plain': no graph)
                # the linkage on the device too: the distance vector (10 GB at
                # 50 000 cells) is never brought to the host
                try:
                    tree = post.ward()
                except _lib.DeviceMemoryError as err:
                    # the full distance matrix (8 N^2 bytes) does not fit the
                    # device: SciPy's own routine on the condensed vector -
                    # the reference's call, the same tree.  Any OTHER failure
                    # (a device fault, a chain that did not close) is raised.
                    print(f'[bnpc] Ward linkage on the device: {err}; '
                        'using scipy.cluster.hierarchy.linkage')
            if tree is None:
                dist = post.dist()
                tree = linkage(dist, method='ward')
        else:
            tree = linkage(dist, method='ward')
        sizable = [int((np.unique(a, return_counts=True)[1] > 2).sum())
            for a in assignments]
        avg = np.mean(sizable)
        candidates = np.arange(max(2, avg * 0.2),
            min(avg * 2.5, assignments.shape[1]), dtype=int)
        if candidates.size == 0:
            # no candidate cut: the reference's loop does not run and its
            # best_assignment stays None (utils.py:116-130)
            return None
        # every candidate cut of the tree (cut_tree's labels, without its
        # O(N^2) replay of the merges: 12 s at 50 000 cells)
        cuts = cut_tree_labels(tree, candidates)
        if post is not None and cuts.max() < 65534:
            dist = None
            labels = np.ascontiguousarray(cuts.T)
            scores = mpear_scores(post.mpear_sums(labels), labels,
                post.differ_sum, assignments.shape[0])
            return _first_maximum(cuts, scores)
        if dist is None:        # labels beyond uint16: the host scores
            dist = post.dist()
    finally:
        if post is not None:
            post.close()
    sim = 1 - dist
    scores = np.array([calc_MPEAR(sim, np.ascontiguousarray(cuts[:, col]))
        for col in range(candidates.size)])
    return _first_maximum(cuts, scores)


def _first_maximum(cuts, scores):
    """The cut the reference's loop keeps (utils.py:119-128: `if score >
    best: ...` from -inf): the FIRST of the largest scores; a NaN score never
    passes `>`, so it is skipped, and if nothing passes the result is None."""
    scores = np.asarray(scores, dtype=np.float64)
    usable = ~np.isnan(scores) & (scores > -np.inf)
    if not usable.any():
        return None
    best = int(np.argmax(np.where(usable, scores, -np.inf)))
    return np.ascontiguousarray(cuts[:, best])


def mean_hierarchy_assignment(assignments, params_full, dist=None):
    """utils.py:148-192: the MPEAR clustering and, per cluster, the mean of
    the sampled parameter vectors of the posterior samples in which the
    cluster's cells sit together (and alone, if such samples exist)."""
    steps = assignments.shape[0]
    assign = get_MPEAR(assignments, dist)
    clusters = np.unique(assign)
    params = np.zeros((clusters.size, params_full.shape[2]))
    for row, cluster in enumerate(clusters):
        member = assign == cluster
        cells = np.flatnonzero(member)
        sub = assignments[:, cells]
        others = assignments[:, np.flatnonzero(~member)]
        together = (sub == sub[:, :1]).all(axis=1)
        major = np.array([np.argmax(np.bincount(r)) for r in sub])
        alone = ~(others == major[:, None]).any(axis=1)
        if together.any():
            pick = together & alone
            if not pick.any():
                pick = together
            chosen = np.flatnonzero(pick)
            for s in chosen:
                present = np.append(np.unique(others[s]), major[s])
                rank = np.argwhere(np.sort(present) == major[s])[0][0]
                params[row] += params_full[s][rank]
            params[row] /= chosen.size
        else:
            for s, sample in enumerate(assignments):
                all_ids = np.unique(sample)
                ids, cnt = np.unique(sample[cells], return_counts=True)
                rows = np.flatnonzero(np.isin(all_ids, ids))
                params[row] += np.dot(cnt, params_full[s][rows])
            params[row] /= steps * cells.size
    return assign, params[assign].T


def concat_chain_results(results):
    """Pool the post-burn-in samples of all chains (utils.py:206-223)."""
    pooled = {k: np.concatenate([r[k][r['burn_in']:] for r in results])
        for k in ('assignments', 'DP_alpha', 'ML', 'MAP', 'FN', 'FP')}
    width = max(r['params'].shape[1] for r in results)
    pooled['params'] = np.concatenate([np.pad(r['params'],
        [(0, 0), (0, width - r['params'].shape[1]), (0, 0)])
        for r in results])
    pooled['burn_in'] = 0
    return pooled


def posterior_estimate(results, data):
    """`-e posterior` (the default estimator), chains pooled
    (utils.py:195-244)."""
    res = concat_chain_results(results)
    assign, geno = mean_hierarchy_assignment(res['assignments'],
        res['params'])
    called = geno.T.round()
    FN_geno = (((called == 1) & (data == 0)).sum() + EPSILON) \
        / (called.sum() + EPSILON)
    FP_geno = (((called == 0) & (data == 1)).sum() + EPSILON) \
        / ((1 - called).sum() + EPSILON)
    return {'a': (np.mean(res['DP_alpha']), np.std(res['DP_alpha'])),
        'assignment': assign.tolist(), 'genotypes': geno.T,
        'FN': (np.mean(res['FN']), np.std(res['FN'])),
        'FP': (np.mean(res['FP']), np.std(res['FP'])),
        'FN_geno': FN_geno, 'FP_geno': FP_geno}
