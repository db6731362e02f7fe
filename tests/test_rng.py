"""CPU: the counter-based random state (brutus_amd/rng.py), the specification
the device stream is checked against on the GPU."""
import numpy as np

from brutus_amd.rng import (PhiloxRandomState, philox4x32, philox_normal,
                            philox_uniform)


def test_philox_known_answers():
    """Random123 known-answer vectors for Philox4x32-10 (the round function and
    key schedule are shared with the 7-round variant used here)."""
    o = philox4x32([0], [0], [0], [0], 0, 0, rounds=10)
    assert [int(x[0]) for x in o] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    o = philox4x32([f], [f], [f], [f], f, f, rounds=10)
    assert [int(x[0]) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    o = philox4x32([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344],
                   0xa4093822, 0x299f31d0, rounds=10)
    assert [int(x[0]) for x in o] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_streams_are_random_access_and_reproducible():
    rs = PhiloxRandomState(7)
    z = rs.normal(size=(40, 3, 5))
    u = rs.random_sample(17)
    assert (rs.n_normal, rs.n_uniform) == (600, 17)
    again = PhiloxRandomState(7, n_normal=100).normal(size=50)
    assert np.array_equal(again, z.ravel()[100:150])
    assert np.array_equal(philox_uniform(7, np.arange(17)), u)
    assert np.array_equal(philox_normal(7, np.array([599, 0, 3])), z.ravel()[[599, 0, 3]])
    assert not np.array_equal(PhiloxRandomState(8).normal(size=10), z.ravel()[:10])
    # scalar calls consume one deviate each
    r2 = PhiloxRandomState(7)
    assert r2.normal() == z.ravel()[0] and r2.normal() == z.ravel()[1]


def test_distributions():
    from scipy import stats
    rs = PhiloxRandomState(2024)
    z = rs.normal(size=400000)
    u = rs.random_sample(200000)
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 0.01
    assert 0. <= u.min() and u.max() < 1.


def test_choice_semantics_match_numpy_legacy():
    """`choice(a, size, p)` == searchsorted(cumsum(p)/sum, uniforms, 'right')."""
    p = np.array([0.1, 0.2, 0.3, 0.15, 0.25])
    rs = PhiloxRandomState(3)
    idx = rs.choice(5, size=1000, p=p)
    u = philox_uniform(3, np.arange(1000))
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    assert np.array_equal(idx, np.searchsorted(cdf, u, side="right"))
    one = rs.choice(5, p=p)
    assert one == np.searchsorted(cdf, philox_uniform(3, np.array([1000]))[0], side="right")
    assert rs.n_uniform == 1001
    counts = np.bincount(idx, minlength=5) / 1000.
    assert np.max(np.abs(counts - p)) < 0.06
