"""The little post-processing the driver and the CLI need.

Restated from /root/reference/libs/utils.py: the lugsail batch-means PSRF
(:427-467, used by the -ls termination mode and the run summary) and the ML /
MAP point estimates (:248-282).  The posterior (MPEAR) estimator, metrics and
tree helpers are out of scope.
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
