"""TEST INFRASTRUCTURE - sequential NaN-skipping sums of the CPU oracle.

The reference sums with Bottleneck (``bn.nansum`` at libs/CRP.py:202,204,234,
363,366 and libs/CRP_learning_errors.py:63; ``bn.nanargmax`` at
libs/CRP.py:90,105).  Bottleneck==1.3.5 (requirements.txt:1) is a third-party
wheel absent from /root/reference; its published algorithm for float64 is a
single accumulator walked in index order that skips NaN, which is what
oracle/seqsum.c and the NumPy form below restate.  The order matters: it is the
order the HIP kernel reproduces bit for bit.
"""
import ctypes
import os

import numpy as np

_DP = ctypes.POINTER(ctypes.c_double)


def _load_helper():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build',
        'liboracle_seqsum.so')
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
    except OSError:
        return None
    lib.bnpc_oracle_nansum.restype = ctypes.c_double
    lib.bnpc_oracle_nansum.argtypes = [_DP, ctypes.c_long]
    for name in ('bnpc_oracle_nansum_axis1', 'bnpc_oracle_nansum_axis0'):
        fn = getattr(lib, name)
        fn.restype = None
        fn.argtypes = [_DP, ctypes.c_long, ctypes.c_long, _DP]
    return lib


_LIB = _load_helper()


def _seqsum_np(v, axis=None):
    """Pure-NumPy form with the same order (cumsum is a sequential scan)."""
    v = np.asarray(v, dtype=np.float64)
    clean = np.where(np.isnan(v), 0.0, v)
    if axis is None:
        flat = clean.ravel()
        return float(np.cumsum(flat)[-1]) if flat.size else 0.0
    if clean.shape[axis] == 0:
        return np.zeros(np.delete(clean.shape, axis).astype(int))
    return np.take(np.cumsum(clean, axis=axis), -1, axis=axis)


def seqsum(v, axis=None):
    """Sum in index order, NaN skipped, float64 accumulator.  Integer input
    (the reference also feeds counts through nansum) is summed exactly."""
    v = np.asarray(v)
    if v.dtype.kind in 'iub':
        return v.sum(axis=axis)
    use_c = _LIB is not None and v.ndim <= 2 \
        and (axis is None or v.ndim == 2)
    if not use_c:
        if v.ndim == 1 and axis == 0:
            axis = None
        return _seqsum_np(v, axis)
    v = np.ascontiguousarray(v, dtype=np.float64)
    src = v.ctypes.data_as(_DP)
    if axis is None:
        return float(_LIB.bnpc_oracle_nansum(src, v.size))
    rows, cols = v.shape
    along_rows = axis in (1, -1)
    out = np.empty(rows if along_rows else cols)
    fn = _LIB.bnpc_oracle_nansum_axis1 if along_rows \
        else _LIB.bnpc_oracle_nansum_axis0
    fn(src, rows, cols, out.ctypes.data_as(_DP))
    return out


def first_nanargmax(v):
    """bottleneck.nanargmax on a vector: position of the FIRST maximum."""
    return int(np.nanargmax(v))
