#!/usr/bin/env python3
"""A/B timing of the k_ll variants in ONE process (interleaved rounds).
usage: kernel_ab.py [N M K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bnpc_amd import _lib  # noqa: E402
import bench  # noqa: E402

N, M, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 \
    else (5000, 1000, 3152)
data = bench.synth(0, N, M, 10, 0.2)
rng = np.random.RandomState(1)
theta = np.clip(rng.uniform(size=(K, M)), 1e-5, 1 - 1e-5).astype(np.float32)
ctx = _lib.Context(data=data)
variants = {
    'cpp,noremap': dict(BNPC_LL_ASM='0', BNPC_XCD_REMAP='0'),
    'cpp,remap': dict(BNPC_LL_ASM='0', BNPC_XCD_REMAP='1'),
    'asm,noremap': dict(BNPC_LL_ASM='1', BNPC_XCD_REMAP='0'),
    'asm,remap': dict(BNPC_LL_ASM='1', BNPC_XCD_REMAP='1'),
    'asm2,noremap': dict(BNPC_LL_ASM='2', BNPC_XCD_REMAP='0'),
    'asm2,remap': dict(BNPC_LL_ASM='2', BNPC_XCD_REMAP='1'),
}
for kw in sys.argv[4:]:
    variants = {f'{k},kw{kw}': dict(v, BNPC_KW=kw) for k, v in variants.items()}
ref = None
times = {k: [] for k in variants}
for rnd in range(5):
    for name, env in variants.items():
        os.environ.update(env)
        ctx.reload_options()
        out = ctx.ll_theta(0, theta, 0.01, 0.2)
        if ref is None:
            ref = out
        assert np.array_equal(out, ref), name
        ctx.bench_ll(1)
        times[name].append(ctx.bench_ll(5))
for name, t in times.items():
    ms = min(t)
    print(f'{name:16s} min {ms:8.4f} ms  median {sorted(t)[len(t) // 2]:8.4f} ms'
        f'  -> {N * K * M / ms / 1e9:8.2f} T elem-evals/s '
        f'({N * K * M / ms / 1e9 / 19650 * 100:5.1f}% of 2-add FP64 peak)')
