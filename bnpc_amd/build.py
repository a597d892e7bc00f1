"""Build libbnpc_hip.so in-tree with hipcc for gfx950 (cross-compiles without
a GPU).  `python -m bnpc_amd.build` or __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SOURCES = [os.path.join(PKG, 'csrc', 'bnpc_kernels.hip'),
    os.path.join(PKG, 'csrc', 'bnpc_sweeps.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_moves.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_step.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_hostmath.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_mt.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_ingest.cpp'),
    os.path.join(PKG, 'csrc', 'bnpc_codist.hip')]
HEADERS = [os.path.join(ROOT, 'include', 'bnpc_hip.h'),
    os.path.join(PKG, 'csrc', 'bnpc_internal.h')]
TARGET = os.path.join(PKG, 'libbnpc_hip.so')
# BNPC_SANITIZE=thread|address,undefined builds the HOST side instrumented
# into build/libbnpc_hip.<sanitizer>.so (git- AND gpurun-ignored: CPU runs
# only, the GPU pool has no sanitizer support); load it with BNPC_LIB=<path>
# and the matching runtime in LD_PRELOAD (tools/run_sanitized.sh).
SANITIZE = os.environ.get('BNPC_SANITIZE', '')
if SANITIZE:
    os.makedirs(os.path.join(ROOT, 'build'), exist_ok=True)
    TARGET = os.path.join(ROOT, 'build',
        f'libbnpc_hip.{SANITIZE.replace(",", "_")}.so')


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'),
            '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def source_digest():
    """sha256 (first 16 hex digits) over the library's sources and headers:
    what a measurement file must carry to be taken for the code that is
    running (bench.py refuses PMC passes of another build)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(SOURCES + HEADERS):
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(TARGET):
        return True
    t = os.path.getmtime(TARGET)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return TARGET
    cmd = [find_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17',
        # the reference rounds every product and sum separately (NumPy):
        # no FMA contraction in tables or in the native sweeps
        '-ffp-contract=off',
        '-fPIC', '-shared',
        '-I' + os.path.join(ROOT, 'include'),
        '-I' + os.path.join(PKG, 'csrc'),
        '-o', TARGET] + SOURCES
    if SANITIZE:
        cmd[2:3] = ['-O1', '-g', '-fno-omit-frame-pointer',
            f'-fsanitize={SANITIZE}', '-shared-libsan']
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    return TARGET


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
