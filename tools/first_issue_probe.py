import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from bnpc_amd import _lib
N, M = 5000, 1000
data = bench.synth(0, N, M, 10, 0.2)
ctx = _lib.Context(data=data)
K = 3152
theta = np.clip(np.random.RandomState(1).uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
prior = np.zeros(K)
perm = np.random.permutation(N)
for rep in range(3):
    t0 = time.perf_counter()
    ll = ctx.ll_theta_pinned_sums_issue(0, theta, 0.01, 0.2, K + 394, prior)
    t1 = time.perf_counter()
    h = ctx.hints_in_order_issue(perm)
    t2 = time.perf_counter()
    ctx.hints_wait()
    t3 = time.perf_counter()
    ctx.matrix_wait()
    t4 = time.perf_counter()
    print(f'rep {rep}: sums_issue {1e3*(t1-t0):.2f} ms, hints_issue {1e3*(t2-t1):.2f}, hints_wait {1e3*(t3-t2):.2f}, matrix_wait {1e3*(t4-t3):.2f}')
# pieces: pageable upload alone
import ctypes as C
t0 = time.perf_counter(); ctx.ll_theta(0, theta[:64], 0.01, 0.2); t1 = time.perf_counter()
print(f'small ll_theta {1e3*(t1-t0):.2f} ms')
