"""Addresses of the scalar kernels SciPy and NumPy evaluate themselves.

The native batch of the parameter moves (csrc/bnpc_hostmath.cpp) must produce
the float64 bits of ``scipy.stats.truncnorm`` / ``beta`` and of ``np.log`` /
``np.exp`` - the proposal is cast to float32 and a 1-ulp difference eventually
changes a cast, i.e. the trajectory.  Instead of restating cephes / the SIMD
loops, the library is handed the functions themselves:

  * ``scipy.special.cython_special.__pyx_capi__`` exports the C entry points
    of the special functions (``double f(double, int skip_dispatch)``) as
    PyCapsules - the same code the ufuncs loop over;
  * a NumPy ufunc object lists its type-specific inner loops
    (``functions[i]`` for ``types[i*nargs:(i+1)*nargs]``); the first 'd->d'
    entry is the one NumPy dispatches float64 arrays to on this CPU.

Both are read here with ctypes, once per process.  Nothing is assumed to work:
`table()` returns None when a capsule or a loop cannot be found, and the caller
(bnpc_amd.model) bit-compares a whole native batch against the SciPy-level
evaluation before it uses the table at all.
"""
import ctypes as C
import os

import numpy as np

_NPY_DOUBLE = 12


class HostKernels(C.Structure):
    """bnpc_host_kernels (include/bnpc_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in (
        'ndtr', 'log_ndtr', 'ndtri_exp', 'sc_log1p', 'xlogy', 'xlog1py',
        'betaln', 'np_log', 'np_exp', 'np_log1p', 'np_expm1', 'np_log_data',
        'np_exp_data', 'np_log1p_data', 'np_expm1_data')] + [
        ('norm_pdf_logC', C.c_double), ('left_ok', C.c_int),
        ('gammaln', C.c_void_p)]


class _UFuncHead(C.Structure):
    """Leading fields of PyUFuncObject (numpy/ufuncobject.h; unchanged
    between NumPy 1.x and 2.x)."""
    _fields_ = [('ob_refcnt', C.c_ssize_t), ('ob_type', C.c_void_p),
        ('nin', C.c_int), ('nout', C.c_int), ('nargs', C.c_int),
        ('identity', C.c_int), ('functions', C.POINTER(C.c_void_p)),
        ('data', C.POINTER(C.c_void_p)), ('ntypes', C.c_int),
        ('reserved1', C.c_int), ('name', C.c_char_p),
        ('types', C.POINTER(C.c_char))]


def _ufunc_loop_dd(ufunc):
    """(function address, data) of the first float64 -> float64 inner loop."""
    head = _UFuncHead.from_address(id(ufunc))
    if head.nin != 1 or head.nout != 1 or head.nargs != 2 \
            or not head.name or head.name.decode() != ufunc.__name__ \
            or not 0 < head.ntypes < 64 or head.ntypes != len(ufunc.types):
        raise LookupError(f'unexpected layout of ufunc {ufunc.__name__}')
    for i in range(head.ntypes):
        if ord(head.types[2 * i]) == _NPY_DOUBLE \
                and ord(head.types[2 * i + 1]) == _NPY_DOUBLE:
            if ufunc.types[i] != 'd->d':
                raise LookupError('type table does not match ufunc.types')
            data = head.data[i] if head.data else None
            return head.functions[i], data
    raise LookupError(f'no d->d loop in {ufunc.__name__}')


def _capsule_pointer(capi, name, signature):
    cap = capi[name]
    api = C.pythonapi
    api.PyCapsule_GetName.restype = C.c_char_p
    api.PyCapsule_GetName.argtypes = [C.py_object]
    api.PyCapsule_GetPointer.restype = C.c_void_p
    api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    sig = api.PyCapsule_GetName(cap)
    if sig is None or sig.decode() != signature:
        raise LookupError(f'{name}: unexpected signature {sig!r}')
    ptr = api.PyCapsule_GetPointer(cap, sig)
    if not ptr:
        raise LookupError(f'{name}: NULL capsule')
    return ptr


def _gammaln_pointer(capi):
    """scipy.special.gammaln's C entry point if it returns the ufunc's bits
    over the arguments a move can hand it (cluster sizes), else None: the
    native moves (bnpc_sm_move) are then not used."""
    try:
        from scipy.special import gammaln
        ptr = _capsule_pointer(capi, 'gammaln', _F1)
        fn = C.CFUNCTYPE(C.c_double, C.c_double, C.c_int)(ptr)
        probe = np.concatenate([np.arange(1, 4097), 2 ** np.arange(13, 40),
            3 ** np.arange(8, 24) + 1]).astype(np.float64)
        want = gammaln(probe)
        got = np.array([fn(float(x), 0) for x in probe])
        return ptr if np.array_equal(want, got) else None
    except Exception:
        return None


_F1 = 'double (double, int __pyx_skip_dispatch)'
_F2 = 'double (double, double, int __pyx_skip_dispatch)'
_cache = {}


def table():
    """The kernel table of this process (a HostKernels instance, kept alive
    here), or None if it cannot be assembled."""
    pid = os.getpid()
    if _cache.get('pid') == pid:
        return _cache['table']
    _cache.clear()
    _cache['pid'] = pid
    _cache['table'] = None
    try:
        import scipy.special.cython_special as cs
        from scipy.stats import _continuous_distns as cd
        from bnpc_amd import fastdist
        capi = cs.__pyx_capi__
        t = HostKernels()
        for field, name, sig in (
                ('ndtr', '__pyx_fuse_1ndtr', _F1),
                ('log_ndtr', '__pyx_fuse_1log_ndtr', _F1),
                ('ndtri_exp', 'ndtri_exp', _F1),
                ('sc_log1p', '__pyx_fuse_1log1p', _F1),
                ('xlogy', '__pyx_fuse_1xlogy', _F2),
                ('xlog1py', '__pyx_fuse_1xlog1py', _F2),
                ('betaln', 'betaln', _F2)):
            setattr(t, field, _capsule_pointer(capi, name, sig))
        for field, uf in (('np_log', np.log), ('np_exp', np.exp),
                ('np_log1p', np.log1p), ('np_expm1', np.expm1)):
            fn, data = _ufunc_loop_dd(uf)
            setattr(t, field, fn)
            setattr(t, field + '_data', data)
        t.norm_pdf_logC = float(cd._norm_pdf_logC)
        # the composed ("shared mass") forms are what the native batch
        # restates; without them there is nothing to compare against
        if not (fastdist.selfcheck() and fastdist._state['shared']):
            return None
        t.left_ok = int(bool(fastdist._state['left']))
        t.gammaln = _gammaln_pointer(capi)
        _cache['table'] = t
    except Exception:
        _cache['table'] = None
    return _cache['table']


def probe():
    """Which of the two private interfaces `table()` rests on can be read in
    this process (bench.py reports them: a box with another NumPy / SciPy
    drops to the SciPy-level path silently otherwise): the C entry points of
    scipy.special.cython_special, the float64 inner loops of NumPy's ufuncs."""
    out = {'scipy_capi': False, 'ufunc_loops': False}
    try:
        import scipy.special.cython_special as cs
        capi = cs.__pyx_capi__
        for name, sig in (('__pyx_fuse_1ndtr', _F1),
                ('__pyx_fuse_1log_ndtr', _F1), ('ndtri_exp', _F1),
                ('__pyx_fuse_1log1p', _F1), ('__pyx_fuse_1xlogy', _F2),
                ('__pyx_fuse_1xlog1py', _F2), ('betaln', _F2)):
            _capsule_pointer(capi, name, sig)
        out['scipy_capi'] = True
    except Exception:       # noqa: BLE001
        pass
    try:
        for uf in (np.log, np.exp, np.log1p, np.expm1):
            _ufunc_loop_dd(uf)
        out['ufunc_loops'] = True
    except Exception:       # noqa: BLE001
        pass
    return out
