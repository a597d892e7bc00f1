#!/usr/bin/env python3
"""Phases of the posterior estimator's clustering step (postproc.get_MPEAR) on
synthetic posterior samples: N cells, S samples around C true clusters.
usage: posterior_bench.py N S [C]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from scipy.cluster.hierarchy import cut_tree, linkage  # noqa: E402

from bnpc_amd import _lib, postproc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
C = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.RandomState(0)
base = rng.randint(0, C, N)
a = np.tile(base, (S, 1)).astype(np.int32)
flip = rng.random_sample((S, N)) < 0.03
a[flip] = rng.randint(0, C, flip.sum())


def lap(label, t0):
    t1 = time.perf_counter()
    print(f'  {label:46s} {t1 - t0:8.3f} s', flush=True)
    return t1


print(f'N={N} S={S} C={C}: {N * (N - 1) // 2:.3e} pairs')
t0 = t_all = time.perf_counter()
post = _lib.Posterior(a)
t0 = lap('pair counts on the device (k_codist) + their sum', t0)
tree = post.ward()
t0 = lap('Ward linkage on the device (k_ward_chain|work) + relabel', t0)
scans, steps = post.ward_stats()
print(f'    ({scans} full row scans for {N - 1} merges and {steps} chain '
    'steps: the other steps took the row\'s cached nearest neighbour)')
if os.environ.get('WARD_CHECK'):
    dist = post.dist()
    t0 = lap('mean distance divided on the device -> host f64', t0)
    want = linkage(dist, method='ward')
    t0 = lap('Ward linkage (SciPy, host)', t0)
    print('  device linkage == SciPy linkage:', np.array_equal(tree, want))
    del dist
sizable = [int((np.unique(r, return_counts=True)[1] > 2).sum()) for r in a]
avg = np.mean(sizable)
cand = np.arange(max(2, avg * 0.2), min(avg * 2.5, N), dtype=int)
cuts = postproc.cut_tree_labels(tree, cand)
t0 = lap(f'tree cuts for {cand.size} candidates (cut_tree_labels)', t0)
labels = np.ascontiguousarray(cuts.T)
sums = post.mpear_sums(labels)
t0 = lap(f'MPEAR sums of {cand.size} candidates (k_mpear_sums)', t0)
scores = postproc.mpear_scores(sums, labels, post.differ_sum, S)
best = int(np.argmax(scores))
t0 = lap('scores from the integers (host)', t0)
print(f'  total {time.perf_counter() - t_all:.3f} s; best cut: '
    f'{cand[best]} clusters, MPEAR {scores[best]:.6f}')
post.close()
if len(sys.argv) > 4:       # the reference-order host evaluation, one candidate
    d = _lib.codist(a) / S
    t0 = time.perf_counter()
    s = postproc.calc_MPEAR(1 - d, labels[best])
    print(f'  host calc_MPEAR of ONE candidate: {time.perf_counter() - t0:.3f}'
        f' s (x {cand.size} candidates in the old path); score {s:.6f}')
