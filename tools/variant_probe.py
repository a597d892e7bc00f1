#!/usr/bin/env python3
"""Probe kernel-selection thresholds at awkward shapes (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bnpc_amd import _lib
import bench
rng = np.random.RandomState(1)
for N, M, K in ((1000, 200, 632), (50000, 5000, 50), (10000, 2000, 200), (5000, 1000, 256)):
    data = bench.synth(0, N, M, 10, 0.2)
    ctx = _lib.Context(data=data)
    theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
    for name, env in (('default', {}), ('nosplit', {'BNPC_MSPLIT': '0'}),
            ('asm2>=256wg', {'BNPC_ASM2_MIN_WGS': '256'}),
            ('asm2>=256wg,nosplit', {'BNPC_ASM2_MIN_WGS': '256', 'BNPC_MSPLIT': '0'}),
            ('kw4', {'BNPC_KW': '4'}), ('kw4,nosplit', {'BNPC_KW': '4', 'BNPC_MSPLIT': '0'})):
        for k in ('BNPC_MSPLIT', 'BNPC_ASM2_MIN_WGS', 'BNPC_KW'):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx.reload_options()
        ctx.ll_theta(0, theta, 0.01, 0.2, fetch=False); ctx.sync(); ctx.bench_ll(2)
        t = min(ctx.bench_ll(5) for _ in range(3))
        print(f'N={N} M={M} K={K} {name:22s}: {t*1e3:8.1f} us  {N*K*M/(t*1e-3)/19.65e12*100:5.1f}% of peak')
    ctx.close()
