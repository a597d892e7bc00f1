"""bnpc_amd.fastdist: wrapper-free SciPy kernels must be BIT-identical to the
public scipy.stats API (they carry the MH proposals of the trajectory)."""
import numpy as np
import pytest
from scipy.stats import beta, truncnorm

from bnpc_amd import fastdist as F

TMIN, TMAX = 1e-5, 1 - 1e-5


def test_selfcheck_passes_on_this_scipy():
    assert F.selfcheck() is True
    assert F._state['shared'] is True


@pytest.mark.parametrize('shape', [(1000,), (7, 333), (1, 64)])
def test_bit_identical_to_public_api(shape):
    rng = np.random.RandomState(1)
    old = np.clip(rng.uniform(size=shape), TMIN, TMAX).astype(np.float32)
    old.flat[:3] = np.float32(TMIN)
    old.flat[3:6] = np.float32(TMAX)
    std = rng.choice(np.array([0.1, 0.25, 0.5]), size=shape)
    a, b = (TMIN - old) / std, (TMAX - old) / std
    U = rng.uniform(size=shape)
    want = truncnorm.ppf(U, a, b, loc=old, scale=std)
    assert np.array_equal(F.tn_rvs_from_uniform(U, a, b, old, std), want)
    draw, fwd = F.tn_propose(U, a, b, old, std)
    assert np.array_equal(draw, want)
    new = want.astype(np.float32)
    assert np.array_equal(fwd(new),
        truncnorm.logpdf(new, a, b, loc=old, scale=std))
    a_rev, b_rev = (TMIN - new) / std, (TMAX - new) / std
    assert np.array_equal(F.tn_logpdf(old, a_rev, b_rev, new, std),
        truncnorm.logpdf(old, a_rev, b_rev, loc=new, scale=std))
    for p, q in ((.25, .25), (.75, 2.)):
        assert np.array_equal(F.beta_logpdf(new, p, q),
            beta(p, q).logpdf(new))


def test_mass_of_intervals_on_the_clips_is_bit_identical():
    """Profile entries ON the upper clip give intervals (a, 0] (SciPy's
    complex-logsumexp 'left' case, restated natively), entries on the lower
    clip give [0, b): all three cases mixed in one array."""
    from scipy.stats import _continuous_distns as cd
    assert F.selfcheck() and F._state['left'] is True
    rng = np.random.RandomState(3)
    for shape in ((4000,), (9, 501)):
        old = np.clip(rng.uniform(size=shape), TMIN, TMAX).astype(np.float32)
        where = rng.uniform(size=shape)
        old[where < .3] = np.float32(TMAX)
        old[where > .8] = np.float32(TMIN)
        std = rng.choice(np.array([0.1, 0.25, 0.5]), size=shape)
        a, b = (TMIN - old) / std, (TMAX - old) / std
        assert (b == 0).any() and (a == 0).any()
        assert np.array_equal(F._tn_mass(a, b), cd._log_gauss_mass(a, b))
        U = rng.uniform(size=shape)
        draw, fwd = F.tn_propose(U, a, b, old, std)
        assert np.array_equal(draw,
            truncnorm.ppf(U, a, b, loc=old, scale=std))
        new = draw.astype(np.float32)
        assert np.array_equal(fwd(new),
            truncnorm.logpdf(new, a, b, loc=old, scale=std))
    # an interval right of zero is not ours: SciPy's own path, same bits
    a = np.array([-1.0, 0.5, -2.0])
    b = np.array([0.0, 1.5, 1.0])
    assert np.array_equal(F._tn_mass(a, b), cd._log_gauss_mass(a, b))


def test_rvs_consumes_the_global_stream_like_scipy():
    rng = np.random.RandomState(2)
    old = np.clip(rng.uniform(size=500), TMIN, TMAX).astype(np.float32)
    std = rng.choice(np.array([0.1, 0.25, 0.5]), size=500)
    a, b = (TMIN - old) / std, (TMAX - old) / std
    np.random.seed(5)
    ref = truncnorm.rvs(a, b, loc=old, scale=std, size=500)
    ref_next = np.random.random()
    np.random.seed(5)
    got = F.tn_rvs_from_uniform(np.random.uniform(size=500), a, b, old, std)
    assert np.array_equal(ref, got) and np.random.random() == ref_next


def test_fallback_when_private_kernels_disagree(monkeypatch):
    monkeypatch.setitem(F._state, 'checked', True)
    monkeypatch.setitem(F._state, 'fast', False)
    monkeypatch.setitem(F._state, 'shared', False)
    x = np.array([0.2, 0.7], dtype=np.float32)
    assert np.array_equal(F.beta_logpdf(x, .25, .25),
        beta(.25, .25).logpdf(x))
    old = np.array([0.4, 0.6], dtype=np.float32)
    std = np.array([0.1, 0.5])
    a, b = (TMIN - old) / std, (TMAX - old) / std
    draw, fwd = F.tn_propose(np.array([0.3, 0.9]), a, b, old, std)
    assert np.array_equal(fwd(x), truncnorm.logpdf(x, a, b, loc=old,
        scale=std))
