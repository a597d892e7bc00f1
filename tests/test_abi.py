"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/bnpc_hip.h declares; the ctypes table covers the header."""
import ctypes
import os
import re

import pytest

from bnpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, 'include', 'bnpc_hip.h')) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bnpc_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert len(syms) >= 20
    for must in ('bnpc_create', 'bnpc_ll_theta', 'bnpc_ll_tables',
            'bnpc_colcounts', 'bnpc_ll_total', 'bnpc_gibbs_sweep'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), \
        'run `python -m bnpc_amd.build` first (hipcc, no GPU needed)'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f'{name} declared but not exported'


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()
    lib = _lib.load()
    assert lib.bnpc_abi_version() == _lib.ABI_VERSION == 12
    assert lib.bnpc_last_error() is not None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_bad_arguments_return_errors_not_crashes():
    lib = _lib.load()
    n = ctypes.c_int64(0)
    assert lib.bnpc_view_size(None, 0, ctypes.byref(n)) != 0
    assert b'NULL' in lib.bnpc_last_error()
    with pytest.raises(RuntimeError, match='libbnpc_hip'):
        _lib.check(lib.bnpc_sync(None), 'sync')


def test_header_is_plain_c(tmp_path):
    """include/bnpc_hip.h is a C header: it must compile with a C compiler."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'use_header.c'
    src.write_text('#include "bnpc_hip.h"\n'
        'int main(void) { bnpc_ctx *c = 0; bnpc_gibbs_state s; bnpc_mt19937 r;'
        ' (void)c; (void)s; (void)r; return bnpc_abi_version ? 0 : 1; }\n')
    res = subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic',
        '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), str(src)],
        capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """The ctypes mirrors of the header's structures (bnpc_amd/_lib.py) have
    the size and the field offsets a C compiler gives them."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    pairs = (('bnpc_chain', _lib.ChainState), ('bnpc_gibbs_state',
        _lib.GibbsState), ('bnpc_move_state', _lib.MoveState),
        ('bnpc_mh_args', _lib.MHArgs), ('bnpc_accept_args', _lib.LogAArgs),
        ('bnpc_mt19937', _lib.MT19937), ('bnpc_legacy_gauss',
        _lib.LegacyGauss))
    lines = ['#include <stdio.h>', '#include <stddef.h>',
        '#include "bnpc_hip.h"', 'int main(void) {']
    want = []
    for cname, mirror in pairs:
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        want.append(ctypes.sizeof(mirror))
        for field in mirror._fields_:
            lines.append(
                f'printf("%zu\\n", offsetof({cname}, {field[0]}));')
            want.append(getattr(mirror, field[0]).offset)
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    res = subprocess.run([gcc, '-std=c99', '-I', os.path.join(ROOT, 'include'),
        str(src), '-o', str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True,
        text=True, check=True).stdout.split()]
    assert got == want
    assert _lib.TOP2.itemsize == 64
