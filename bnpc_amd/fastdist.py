"""Wrapper-free calls into SciPy's truncated-normal and Beta distributions.

The MH proposal of the reference (libs/CRP.py:328-357, 371-376) calls
``truncnorm.rvs``, ``truncnorm.logpdf`` (x2) and ``beta.logpdf`` (x2) once per
cluster per step.  On M ~ 1000 elements most of their time is SciPy's generic
argument machinery (broadcasting, argsreduce, place), not arithmetic.  The
functions below evaluate the SAME SciPy kernels (``truncnorm._ppf``,
``truncnorm._logpdf``, ``beta._logpdf``) directly, with the public wrappers'
standardisation and support handling restated, so the values are bit-identical
to the public API - which `selfcheck()` verifies once per process on vectors
that cover the left / right / central branches; if anything differs (another
SciPy version) the public API is used instead.  Arrays of any shape are
accepted, which lets the caller evaluate all clusters of a step at once.
"""
import numpy as np
from scipy.stats import beta as _beta, truncnorm as _truncnorm
from scipy.stats import gamma as _gamma

try:        # SciPy's own building blocks of truncnorm (private module)
    from scipy.stats import _continuous_distns as _cd
    import scipy.special as _sc
    _HAVE_PARTS = all(hasattr(_cd, n) for n in
        ('_log_gauss_mass', '_log_sum', '_norm_logcdf', '_norm_logpdf')) \
        and hasattr(_sc, 'ndtri_exp')
except Exception:           # pragma: no cover
    _HAVE_PARTS = False

_state = {'checked': False, 'fast': False, 'shared': False, 'left': False}


# -- direct forms ------------------------------------------------------------
def _tn_ppf_direct(q, a, b, loc, scale):
    return _truncnorm._ppf(q, a, b) * scale + loc


def _tn_logpdf_direct(x, a, b, loc, scale):
    xs = np.asarray((x - loc) / scale, dtype=np.float64)
    out = _truncnorm._logpdf(xs, a, b) - np.log(scale)
    outside = ~((a <= xs) & (xs <= b))
    if outside.any():
        out = np.array(out, dtype=np.float64)
        out[outside] = -np.inf
    return out


def _beta_logpdf_direct(x, p, q):
    xs = np.asarray(x, dtype=np.float64)
    out = _beta._logpdf(xs, p, q)
    outside = ~((0 < xs) & (xs < 1))
    if outside.any():
        out = np.array(out, dtype=np.float64)
        out[outside] = -np.inf
    return out


def _gamma_logpdf_direct(x, a, loc, scale):
    xs = np.asarray((x - loc) / scale, dtype=np.float64)
    if not np.all(xs > 0):
        return _gamma.logpdf(x, a, loc=loc, scale=scale)
    return _gamma._logpdf(xs, a) - np.log(scale)


# -- composed forms: the Gaussian mass of (a, b) is evaluated ONCE and shared
#    by the proposal's ppf and its forward log-density (scipy's _ppf and
#    _logpdf each recompute it) --------------------------------------------
def _tn_mass(a, b):
    """scipy's _log_gauss_mass(a, b), per element:

      central (a <= 0 < b; a proposal around a value strictly inside the
               bounds): log1p(-ndtr(a) - ndtr(-b)), evaluated directly without
               the per-case masks and the complex scratch array;
      left    (b <= 0; a profile entry sitting ON the upper clip, where the
               float32 difference TMAX - theta is exactly 0): SciPy takes
               logsumexp([logcdf(b), logcdf(a) + pi*1j]) through complex
               arithmetic - restated natively (_lib.log_diff_pi) and used only
               if the self-check found it bit-identical;
      anything else goes to SciPy."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape or a.ndim == 0:
        return _cd._log_gauss_mass(a, b)
    central = (a <= 0) & (b > 0)
    if central.all():
        return _sc.log1p(-_sc.ndtr(a) - _sc.ndtr(-b))
    if not _state['left'] or (a > 0).any():
        return _cd._log_gauss_mass(a, b)
    # the central form on everything (cheaper than gathering the central
    # elements), then the left-of-zero elements are overwritten
    with np.errstate(all='ignore'):
        out = _sc.log1p(-_sc.ndtr(a) - _sc.ndtr(-b))
    where = np.flatnonzero(~central)
    flat = out.reshape(-1)
    flat[where] = _mass_left(a.reshape(-1)[where], b.reshape(-1)[where])
    return flat.reshape(a.shape)


def _mass_left(a, b):
    from bnpc_amd import _lib
    if not b.any():         # every interval ends exactly at 0
        log_p = np.full(b.shape, _cd._norm_logcdf(0.0))
    else:
        log_p = _cd._norm_logcdf(b)
    return _lib.log_diff_pi(log_p, _cd._norm_logcdf(a))


def _log_sum(p, q):
    """scipy's _log_sum(p, q) = logsumexp([p, q], axis=0) for two finite real
    arrays, with the arithmetic of scipy.special._logsumexp spelled out (max
    removed from the sum, log1p of the remaining exponential, log of the
    multiplicity of the max) - the same ufuncs in the same order, without the
    array-API plumbing.  Non-finite input goes to SciPy."""
    if not (np.isfinite(p).all() and np.isfinite(q).all()):
        return _cd._log_sum(p, q)
    top = np.maximum(p, q)
    tie = p == q
    s = np.where(tie, 0.0, np.exp(np.minimum(p, q) - top))
    m = np.where(tie, 2.0, 1.0)
    return np.log1p(s) + np.log(m) + top


def _tn_ppf_shared(q, a, b, loc, scale, lgm):
    """truncnorm._ppf with the mass given: scipy's left-tail form where
    a < 0, its mirrored right-tail form elsewhere (a == 0: a profile entry on
    the lower clip).  The left form is evaluated on every element - cheaper
    than gathering - and the few others are overwritten through an index."""
    q, a, b, lgm = np.broadcast_arrays(q, a, b, lgm)
    other = np.flatnonzero(a >= 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        if other.size < a.size:
            out = _sc.ndtri_exp(_log_sum(_cd._norm_logcdf(a),
                np.log(q) + lgm))
        else:
            out = np.empty(a.shape)
        if other.size:
            qr, br, mr = (v.reshape(-1)[other] for v in (q, b, lgm))
            out = np.ascontiguousarray(out)
            out.reshape(-1)[other] = -_sc.ndtri_exp(_log_sum(
                _cd._norm_logcdf(-br), np.log1p(-qr) + mr))
    return out * scale + loc


def _tn_logpdf_shared(x, a, b, loc, scale, lgm):
    xs = np.asarray((x - loc) / scale, dtype=np.float64)
    out = _cd._norm_logpdf(xs) - lgm - np.log(scale)
    outside = ~((a <= xs) & (xs <= b))
    if outside.any():
        out = np.array(out, dtype=np.float64)
        out[outside] = -np.inf
    return out


# -- public forms ------------------------------------------------------------
def _tn_ppf_public(q, a, b, loc, scale):
    return _truncnorm.ppf(q, a, b, loc=loc, scale=scale)


def _tn_logpdf_public(x, a, b, loc, scale):
    return _truncnorm.logpdf(x, a, b, loc=loc, scale=scale)


def _beta_logpdf_public(x, p, q):
    return _beta.logpdf(x, p, q)


def _left_case_is_exact(rng):
    """Bit-compare the native left-of-zero mass with SciPy's on intervals
    ending at 0 (the case that occurs), ending below 0, tiny, wide and
    degenerate ones."""
    try:
        n = 4096
        width = np.concatenate([rng.uniform(1e-9, 12, size=n // 2),
            10.0 ** rng.uniform(-12, 1.2, size=n // 2)])
        b = np.where(rng.uniform(size=n) < .6, 0.0,
            -(10.0 ** rng.uniform(-9, 1, size=n)))
        a = b - width
        a[:4] = [-40.0, -1e3, -1e-300, -5e-324]
        b[:4] = 0.0
        with np.errstate(all='ignore'):
            want = _cd._log_gauss_mass(a, b)
            got = _mass_left(a, b)
        return bool(np.array_equal(want, got, equal_nan=True))
    except Exception:
        return False


def selfcheck():
    """Bit-compare the direct forms with SciPy's public API."""
    if _state['checked']:
        return _state['fast']
    _state['checked'] = True
    try:
        rng = np.random.RandomState(12345)
        M = 257
        old = np.clip(rng.uniform(size=M), 1e-5, 1 - 1e-5).astype(np.float32)
        old[:8] = np.float32(1e-5)          # a == 0: right branch of ppf
        old[8:16] = np.float32(1 - 1e-5)    # b == 0: left branch of the mass
        old[16:24] = np.float32(0.5)
        std = rng.choice(np.array([0.1, 0.25, 0.5]), size=M)
        a = (1e-5 - old) / std
        b = ((1 - 1e-5) - old) / std
        U = rng.uniform(size=M)
        state = np.random.get_state()
        np.random.seed(99)
        ref_rvs = _truncnorm.rvs(a, b, loc=old, scale=std, size=M)
        np.random.seed(99)
        got_rvs = _tn_ppf_direct(np.random.uniform(size=M), a, b, old, std)
        np.random.set_state(state)
        ok = np.array_equal(ref_rvs, got_rvs)
        new = got_rvs.astype(np.float32)
        new[30:34] = np.float32(0.0)        # outside the support -> -inf
        for x, lo, sc_ in ((new, old, std), (old, new, std)):
            aa = (1e-5 - lo) / sc_
            bb = ((1 - 1e-5) - lo) / sc_
            with np.errstate(all='ignore'):
                ok &= np.array_equal(_tn_logpdf_public(x, aa, bb, lo, sc_),
                    _tn_logpdf_direct(x, aa, bb, lo, sc_))
        a0 = (0 - old) / std
        b0 = (1 - old) / std
        ok &= np.array_equal(_tn_logpdf_public(new, a0, b0, old, std),
            _tn_logpdf_direct(new, a0, b0, old, std))
        ok &= np.array_equal(_tn_ppf_public(U, a, b, old, std),
            _tn_ppf_direct(U, a, b, old, std))
        xs = new.copy()
        xs[30:34] = np.float32(0.3)
        for p, q in ((.25, .25), (.75, 2.0), (3.0, .5)):
            ok &= np.array_equal(_beta_logpdf_public(xs, p, q),
                _beta_logpdf_direct(xs, p, q))
            ok &= np.array_equal(
                _beta_logpdf_public(xs.reshape(1, -1), p, q),
                _beta_logpdf_direct(xs.reshape(1, -1), p, q))
        # scalar arguments (error-rate and concentration priors)
        for x, lo, sc_ in ((0.013, 0.01, 0.01), (0.19, 0.2, 0.1),
                (0.0004, 0.001, 0.0005)):
            aa, bb = (0 - lo) / sc_, (1 - lo) / sc_
            ok &= bool(_tn_logpdf_public(x, aa, bb, lo, sc_)
                == _tn_logpdf_direct(x, aa, bb, lo, sc_))
        # a scalar draw (error-rate proposals, CRP_learning_errors.py:81-84):
        # truncnorm.rvs with size=None = ppf of ONE uniform of the stream
        for lo, sc_ in ((0.01, 0.005), (0.2, 0.15), (0.0007, 0.00075)):
            aa, bb = (0 - lo) / sc_, (1 - lo) / sc_
            state = np.random.get_state()
            np.random.seed(4242)
            ref = _truncnorm.rvs(aa, bb, loc=lo, scale=sc_)
            after_ref = np.random.random()
            np.random.seed(4242)
            got = _tn_ppf_direct(np.random.uniform(), aa, bb, lo, sc_)
            after_got = np.random.random()
            np.random.set_state(state)
            ok &= bool(ref == got) and after_ref == after_got \
                and np.ndim(got) == 0
        for x, aa, lo in ((70.7, 70.71, 1), (3.2, 2.5, 0.5), (31.0, 31.6, 1)):
            ok &= bool(_gamma.logpdf(x, aa, loc=lo, scale=1)
                == _gamma_logpdf_direct(x, aa, lo, 1))
        _state['fast'] = bool(ok)
        if ok and _HAVE_PARTS:
            _state['left'] = _left_case_is_exact(rng)
            with np.errstate(all='ignore'):
                lgm = _tn_mass(a, b)
                shared = np.array_equal(
                    _tn_ppf_shared(U, a, b, old, std, lgm),
                    _tn_ppf_public(U, a, b, old, std))
                shared &= np.array_equal(
                    _tn_logpdf_shared(new, a, b, old, std, lgm),
                    _tn_logpdf_public(new, a, b, old, std))
            _state['shared'] = bool(shared)
    except Exception:
        _state['fast'] = False
        _state['shared'] = False
    return _state['fast']


def tn_rvs_from_uniform(U, a, b, loc, scale):
    """truncnorm.rvs(a, b, loc, scale, size) given its uniforms U
    (= random_state.uniform(size=size), scipy/stats/_distn_infrastructure.py
    rv_continuous._rvs): ppf(U) * scale + loc."""
    if selfcheck():
        return _tn_ppf_direct(U, a, b, loc, scale)
    return _truncnorm._ppf(U, a, b) * scale + loc


def tn_rvs_scalar(a, b, loc, scale):
    """truncnorm.rvs(a, b, loc=loc, scale=scale) for scalar arguments: one
    uniform from the global stream, then the ppf kernel - without the public
    wrapper's argument machinery when the self-check allows."""
    if selfcheck():
        return np.float64(_tn_ppf_direct(np.random.uniform(), a, b, loc, scale))
    return _truncnorm.rvs(a, b, loc=loc, scale=scale)


def tn_logpdf(x, a, b, loc, scale):
    if selfcheck():
        if _state['shared'] and np.ndim(a) > 0:
            return _tn_logpdf_shared(x, a, b, loc, scale, _tn_mass(a, b))
        return _tn_logpdf_direct(x, a, b, loc, scale)
    return _tn_logpdf_public(x, a, b, loc, scale)


def beta_logpdf(x, p, q):
    if selfcheck():
        return _beta_logpdf_direct(x, p, q)
    return _beta_logpdf_public(x, p, q)


def gamma_logpdf(x, a, loc=0, scale=1):
    if selfcheck():
        return _gamma_logpdf_direct(x, a, loc, scale)
    return _gamma.logpdf(x, a, loc=loc, scale=scale)


def tn_propose(U, a, b, loc, scale):
    """The proposal draw and its forward log-density in one go:
    x = truncnorm.rvs(...) from its uniforms U, and a function giving
    truncnorm.logpdf(y, a, b, loc, scale) for that same (a, b) - the Gaussian
    mass of the interval is computed once.  Bit-identical to the public API
    (self-checked); falls back to two independent evaluations otherwise."""
    if selfcheck() and _state['shared']:
        lgm = _tn_mass(a, b)
        x = _tn_ppf_shared(U, a, b, loc, scale, lgm)
        return x, (lambda y: _tn_logpdf_shared(y, a, b, loc, scale, lgm))
    x = tn_rvs_from_uniform(U, a, b, loc, scale)
    return x, (lambda y: tn_logpdf(y, a, b, loc, scale))
