#!/usr/bin/env python3
"""A/B: does threading the MH proposal batch pay at K x M = 18 x 1000 ... ?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from bnpc_amd import model as P
import libs.CRP_learning_errors as dev
for N, M, K in ((5000, 1000, 18), (5000, 1000, 40), (10000, 2000, 20), (50000, 5000, 50)):
    data = bench.synth(0, N, M, 10, 0.2)
    m = bench.make_model(None, dev, data, True)
    np.random.seed(1)
    m.init(assign=list(np.random.RandomState(0).randint(0, K, N)))
    m.update_parameters()
    for thr, nthreads in ((1 << 30, 1), (1 << 13, 4), (1 << 13, 8), (1 << 13, 16)):
        P._THREAD_MIN_ELEMS = thr
        os.environ['BNPC_HOST_THREADS'] = str(nthreads)
        P._POOL.clear()
        m.update_parameters()
        t0 = time.perf_counter()
        for _ in range(30):
            m.update_parameters()
        print(f'N={N} M={M} K={K} threads={nthreads:2d}: {1e3 * (time.perf_counter() - t0) / 30:7.2f} ms per update_parameters')
    m.close()
