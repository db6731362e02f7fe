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


def test_ziggurat_normals_in_equal_probability_bins_and_in_the_tail():
    """2 * 10^6 normals: chi-square over 400 equal-probability bins (the layer edges of the
    ziggurat fall inside the bins, so a wrong wedge or rectangle shows), the mass beyond
    R = X[1] (the tail loop) and the moments."""
    from scipy import stats
    from brutus_amd.rng import ZIG_X
    n = 2000000
    z = philox_normal(99, np.arange(n, dtype=np.uint64))
    edges = stats.norm.ppf(np.linspace(0., 1., 401))
    obs = np.histogram(z, bins=edges)[0]
    chi2 = np.sum((obs - n / 400.) ** 2 / (n / 400.))
    assert stats.chi2.sf(chi2, 399) > 1e-3, chi2
    R = ZIG_X[1]
    ntail, ptail = np.sum(np.abs(z) > R), 2 * stats.norm.sf(R)
    assert abs(ntail - n * ptail) < 5 * np.sqrt(n * ptail), (ntail, n * ptail)
    assert abs(np.mean(z)) < 4 / np.sqrt(n) and abs(np.mean(z ** 2) - 1) < 4 * np.sqrt(2. / n)
    assert abs(np.mean(z ** 4) - 3) < 4 * np.sqrt(96. / n)
    # the wedges: layers' outer parts, |z| in [X[i+1], X[i]) -- compare a band of them
    for lo, hi in ((ZIG_X[3], ZIG_X[2]), (ZIG_X[600], ZIG_X[590])):
        p = 2 * (stats.norm.cdf(hi) - stats.norm.cdf(lo))
        k = np.sum((np.abs(z) >= lo) & (np.abs(z) < hi))
        assert abs(k - n * p) < 5 * np.sqrt(n * p) + 1, (lo, hi, k, n * p)


def test_ziggurat_table_is_closed_and_equals_the_librarys():
    """Equal-area layers under exp(-x^2/2), and the very same numbers in the HIP library
    (host copy of csrc/zig_table.inc; loading the library needs no GPU)."""
    import ctypes as C
    from brutus_amd import _lib
    from brutus_amd.rng import ZIG_N, ZIG_X, ZIG_Y
    assert ZIG_X.shape == ZIG_Y.shape == (ZIG_N + 1,) and ZIG_X[-1] == 0. and ZIG_Y[-1] == 1.
    assert np.all(np.diff(ZIG_X) < 0) and np.all(np.diff(ZIG_Y) > 0)
    assert np.allclose(ZIG_Y[1:], np.exp(-0.5 * ZIG_X[1:] ** 2), rtol=1e-14, atol=0)
    v = ZIG_X[1:-1] * (ZIG_Y[2:] - ZIG_Y[1:-1])
    assert np.ptp(v) < 1e-15
    from scipy import special
    R = ZIG_X[1]
    v0 = R * ZIG_Y[1] + np.sqrt(np.pi / 2) * special.erfc(R / np.sqrt(2))      # base strip + tail
    assert abs(v0 - v.mean()) < 1e-15 and abs(ZIG_X[0] - v0 / ZIG_Y[1]) < 1e-12
    x, y = np.empty(ZIG_N + 1), np.empty(ZIG_N + 1)
    _lib.check(_lib.lib().brutus_debug_zig_table(x.ctypes.data_as(C.c_void_p),
                                                 y.ctypes.data_as(C.c_void_p), ZIG_N + 1))
    assert np.array_equal(x, ZIG_X) and np.array_equal(y, ZIG_Y)


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
