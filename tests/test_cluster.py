"""cluster.isochrone_loglike: oracle vs reference golden (CPU) and HIP vs both
(GPU).  Float tolerance 1e-9 relative on per-object mixture log-likelihoods."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, make_cluster_data, relerr

CASES = (("a", 200, 6, 1), ("b", 300, 8, 2))
THETA = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])


def _run(fn, tag, nobj, nb, seed):
    iso, phot, err, par, perr = make_cluster_data(nobj, nb, seed)
    out = {}
    for dp in (True, False):
        out["dp%d" % dp] = fn(THETA, iso, phot.copy(), err.copy(),
                              parallax=par.copy(), parallax_err=perr.copy(),
                              dim_prior=dp, return_lnls=True)
    theta2 = np.concatenate([THETA, np.linspace(0.97, 1.03, nb - 1), [0.5]])
    out["free"] = fn(theta2, iso, phot.copy(), err.copy(),
                     offsets=[1.0] + [None] * (nb - 1),
                     corr_params=[None, 0., 0., 1.], return_lnls=True)
    return out


def _check(out, z, tag, tol):
    for key in ("dp1", "dp0", "free"):
        tot, mix = out[key]
        assert relerr(z["%s_%s_mix" % (tag, key)], mix) < tol, (tag, key)
        assert abs(tot - float(z["%s_%s_tot" % (tag, key)])) < tol * abs(tot) * 10


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_oracle_matches_reference(case):
    from oracle import brutus_oracle as O
    z = np.load(os.path.join(GOLDEN, "cluster.npz"))
    _check(_run(O.isochrone_loglike, *case), z, case[0], 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_hip_matches_reference(case):
    from brutus_amd import cluster
    z = np.load(os.path.join(GOLDEN, "cluster.npz"))
    _check(_run(cluster.isochrone_loglike, *case), z, case[0], 1e-9)


@pytest.mark.gpu
def test_hip_config5_size_vs_oracle():
    """BASELINE configs[4] at its own size -- 5 000 objects, 12 bands, the full 15 x 2 000
    isochrone table -- against the oracle (reference cluster.py:336-414).  The per-object
    mixture log-likelihoods are independent of each other (the catalogue only meets in the
    final sum), so the oracle, whose (Ncmd, Nobj, Nb) temporaries are 1 GB per slice at this
    size, is evaluated in object chunks of 250 and its chunk totals are added up."""
    from brutus_amd import cluster
    from oracle import brutus_oracle as O
    iso, phot, err, par, perr = make_cluster_data(5000, 12, 5)
    a = cluster.isochrone_loglike(THETA, iso, phot, err, parallax=par,
                                  parallax_err=perr, return_lnls=True)
    assert np.isfinite(a[0]) and a[1].shape == (5000,)
    tot, mix = 0., []
    for lo in range(0, 5000, 250):
        sl = slice(lo, lo + 250)
        c = O.isochrone_loglike(THETA, iso, phot[sl], err[sl], parallax=par[sl],
                                parallax_err=perr[sl], return_lnls=True)
        tot += c[0]
        mix.append(c[1])
    mix = np.concatenate(mix)
    assert relerr(mix, a[1]) < 1e-9
    assert abs(a[0] - tot) < 1e-9 * abs(tot)
    # without the dimensionality prior the outlier term spans the whole catalogue
    # (cluster.py:304-322), so there a 250-object catalogue of its own is compared
    sl = slice(1000, 1250)
    kw = dict(parallax=par[sl], parallax_err=perr[sl], return_lnls=True, dim_prior=False)
    b = cluster.isochrone_loglike(THETA, iso, phot[sl], err[sl], **kw)
    c = O.isochrone_loglike(THETA, iso, phot[sl], err[sl], **kw)
    assert relerr(c[1], b[1]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("nbands", [5, 16, 20])
def test_hip_band_counts_full_table_vs_oracle(nbands):
    """Band counts that need padding (5 -> 8, 20 -> 24) and the default
    15 x 2000 isochrone table, which no longer fits one LDS stage at >= 16
    bands: the kernel walks it in sub-slices."""
    from brutus_amd import cluster
    from oracle import brutus_oracle as O
    iso, phot, err, par, perr = make_cluster_data(60, nbands, 7)
    a = cluster.isochrone_loglike(THETA, iso, phot, err, parallax=par,
                                  parallax_err=perr, return_lnls=True)
    c = O.isochrone_loglike(THETA, iso, phot, err, parallax=par,
                            parallax_err=perr, return_lnls=True)
    assert relerr(c[1], a[1]) < 1e-9
    assert abs(a[0] - c[0]) < 1e-8 * abs(c[0])


def test_errors_match_reference_messages():
    from brutus_amd import cluster
    iso, phot, err, par, perr = make_cluster_data(20, 6, 3)
    with pytest.raises(ValueError, match="photometry must be provided"):
        cluster.isochrone_loglike(THETA, iso, None, err)
    with pytest.raises(ValueError, match="degeneracy"):
        cluster.isochrone_loglike(THETA, iso, phot, err, offsets='free')
    with pytest.raises(ValueError, match="parallax errors"):
        cluster.isochrone_loglike(THETA, iso, phot, err, parallax=par)
    bad = phot.copy()
    bad[3] = np.nan
    with pytest.raises(ValueError, match="no valid data"):
        cluster.isochrone_loglike(THETA, iso, bad, err)


@pytest.mark.gpu
def test_batched_plugin_hook_and_caches_change_nothing():
    """The optional `get_seds_grid` hook, the per-dataset cache and the point-table cache
    are pure accelerations: a plug-in without the hook, `cache=False`, a first call and a
    repeated call all give the same per-object values (<= 1e-12; the hook returns the very
    magnitudes `get_seds` does)."""
    from brutus_amd import cluster, synth

    class NoHook(object):                      # the reference's plug-in surface only
        def __init__(self, iso):
            self.iso = iso

        def get_seds(self, **kw):
            return self.iso.get_seds(**kw)

    iso = synth.TableIsochrone(nbands=12, neep=2000)
    phot, err, par, perr = synth.make_cluster(iso, 700, seed=3)
    kw = dict(parallax=par, parallax_err=perr, return_lnls=True)
    cluster.clear_caches()
    ref = cluster.isochrone_loglike(THETA, NoHook(iso), phot, err, cache=False, **kw)
    first = cluster.isochrone_loglike(THETA, iso, phot, err, **kw)
    again = cluster.isochrone_loglike(THETA, iso, phot, err, **kw)
    th2 = THETA + np.array([0.02, -0.03, 0.05, 0., 12., 0.01])
    moved = cluster.isochrone_loglike(th2, iso, phot, err, **kw)
    moved_ref = cluster.isochrone_loglike(th2, NoHook(iso), phot, err, cache=False, **kw)
    for a in (first, again):
        assert relerr(ref[1], a[1]) < 1e-12 and abs(a[0] - ref[0]) <= 1e-12 * abs(ref[0])
    assert relerr(moved_ref[1], moved[1]) < 1e-12
    assert abs(moved[0] - ref[0]) > 1e-3          # a different theta is a different answer
    # offsets: the cached device copies of the photometry are bypassed, not reused
    nb = 12
    theta3 = np.concatenate([THETA, np.linspace(0.97, 1.03, nb - 1)])
    off = cluster.isochrone_loglike(theta3, iso, phot, err, offsets=[1.0] + [None] * (nb - 1), **kw)
    off_ref = cluster.isochrone_loglike(theta3, NoHook(iso), phot, err, cache=False,
                                        offsets=[1.0] + [None] * (nb - 1), **kw)
    assert relerr(off_ref[1], off[1]) < 1e-12
    # a catalogue edited in place is a new catalogue
    phot2 = phot.copy()
    a = cluster.isochrone_loglike(THETA, iso, phot2, err, **kw)
    phot2[5, 3] *= 1.5
    b = cluster.isochrone_loglike(THETA, iso, phot2, err, **kw)
    assert a[1][5] != b[1][5] and np.array_equal(np.delete(a[1], 5), np.delete(b[1], 5))
    # a plug-in changed in place announces it through `cache_token`
    class Shifted(NoHook):
        shift = 0.

        def get_seds(self, **kw):
            seds, p1, p2 = self.iso.get_seds(**kw)
            return seds + self.shift, p1, p2
    plug = Shifted(iso)
    v0 = cluster.isochrone_loglike(THETA, plug, phot, err, **kw)
    plug.shift, plug.cache_token = 0.05, 1
    v1 = cluster.isochrone_loglike(THETA, plug, phot, err, **kw)
    v1_ref = cluster.isochrone_loglike(THETA, plug, phot, err, cache=False, **kw)
    assert abs(v1[0] - v0[0]) > 1e-3 and relerr(v1_ref[1], v1[1]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("smf", [(0.,), (0., 1.), (0., 0.3, 0.6, 1.), None])
def test_hip_short_mass_fraction_grids_vs_oracle(smf):
    """One, two, four and the default fifteen secondary-mass-fraction slices (fewer slices
    than plug-in groups; a single slice has the weight ln 1) and a short EEP grid against the
    oracle, with the fake isochrone of the golden cases (per-slice `get_seds` only)."""
    from brutus_amd import cluster
    from oracle import brutus_oracle as O
    iso, phot, err, par, perr = make_cluster_data(300, 8, 3)
    kw = dict(parallax=par, parallax_err=perr, return_lnls=True, smf_grid=smf,
              eep_grid=np.linspace(202., 808., 257))
    a = cluster.isochrone_loglike(THETA, iso, phot, err, cache=False, **kw)
    c = O.isochrone_loglike(THETA, iso, phot, err, **kw)
    assert relerr(c[1], a[1]) < 1e-9 and abs(a[0] - c[0]) < 1e-9 * abs(c[0])


@pytest.mark.gpu
def test_pipelined_groups_change_nothing(monkeypatch):
    """The point table is built and summed a few mass-fraction slices at a time
    (`brutus_cluster_lnl_part` per group, `brutus_cluster_lnl_merge` at the end) while the
    plug-in works on the next group: 1, 2, 3, 4 and 15 groups, with and without the batched
    hook, and the one-go sum of the cached table all give the same per-object values."""
    from brutus_amd import cluster, synth

    class NoHook(object):
        def __init__(self, iso):
            self.iso = iso

        def get_seds(self, **kw):
            return self.iso.get_seds(**kw)

    iso = synth.TableIsochrone(nbands=12, neep=2000)
    phot, err, par, perr = synth.make_cluster(iso, 900, seed=4)
    kw = dict(parallax=par, parallax_err=perr, return_lnls=True)
    monkeypatch.setenv("BRUTUS_CLUSTER_PIPELINE", "1")
    ref = cluster.isochrone_loglike(THETA, iso, phot, err, cache=False, **kw)
    for groups in (2, 3, 4, 15):
        monkeypatch.setenv("BRUTUS_CLUSTER_PIPELINE", str(groups))
        for plug in (iso, NoHook(iso)):
            got = cluster.isochrone_loglike(THETA, plug, phot, err, cache=False, **kw)
            assert relerr(ref[1], got[1]) < 1e-12, (groups, type(plug).__name__)
        cluster.clear_caches()
        first = cluster.isochrone_loglike(THETA, iso, phot, err, **kw)      # in pieces
        again = cluster.isochrone_loglike(THETA, iso, phot, err, **kw)      # cached table, one go
        assert relerr(ref[1], first[1]) < 1e-12 and relerr(ref[1], again[1]) < 1e-12
    # an isochrone none of whose points survives the mass bound: -inf for every object
    dead = cluster.isochrone_loglike(THETA, iso, phot, err, mini_bound=1e9, cache=False, **kw)
    none = cluster.isochrone_loglike(THETA, iso, phot, err, mini_bound=1e9, cache=False,
                                     **dict(kw, return_lnls=False))
    assert np.all(np.isfinite(dead[1])) and dead[0] == none      # (the outlier term alone)


@pytest.mark.gpu
def test_cluster_lnl_in_parts_equals_one_call():
    """C ABI: `brutus_cluster_lnl_part` over three uneven pieces of the point list (one of
    them empty) + `brutus_cluster_lnl_merge` against one `brutus_cluster_lnl` call."""
    import torch
    from brutus_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(8)
    nobj, nb, npts = 700, 12, 5000
    dev = torch.device("cuda:0")
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    flux = rng.uniform(0.5, 2., size=(npts, nb))
    flux[rng.uniform(size=flux.shape) < 0.01] = np.nan
    lnw = rng.normal(size=npts)
    phot = rng.uniform(0.5, 2., size=(nobj, nb))
    ivar = rng.uniform(100., 900., size=(nobj, nb)) * (rng.uniform(size=(nobj, nb)) > 0.1)
    t = [up(flux), up(lnw), up(phot), up(ivar), up(rng.uniform(0., 3., nobj)),
         up(rng.normal(size=nobj)), up(rng.randint(1, 14, nobj), np.int32)]
    ws = torch.empty(L.brutus_cluster_workspace_bytes(nobj), dtype=torch.uint8, device=dev)
    nchunk = L.brutus_cluster_chunks()
    assert nchunk == 256
    for dim_prior in (1, 0):
        one = torch.empty(nobj, dtype=torch.float64, device=dev)
        _lib.check(L.brutus_cluster_lnl(nobj, nb, npts, *[x.data_ptr() for x in t], dim_prior,
                                        ws.data_ptr(), ws.numel(), one.data_ptr(), None))
        one = one.cpu().numpy()
        parts = torch.empty(nobj, dtype=torch.float64, device=dev)
        for (a, b), (c0, c1) in zip(((0, 1700), (1700, 1700), (1700, npts)),
                                    ((0, 100), (100, 101), (101, nchunk))):
            _lib.check(L.brutus_cluster_lnl_part(
                nobj, nb, b - a, t[0][a:].data_ptr() if b > a else None,
                t[1][a:].data_ptr() if b > a else None, *[x.data_ptr() for x in t[2:]],
                dim_prior, ws.data_ptr(), ws.numel(), c0, c1 - c0, None))
        _lib.check(L.brutus_cluster_lnl_merge(nobj, nchunk, ws.data_ptr(), ws.numel(),
                                              parts.data_ptr(), None))
        torch.cuda.synchronize()
        assert relerr(one, parts.cpu().numpy()) < 1e-13
    with pytest.raises(ValueError, match="chunk range"):
        _lib.check(L.brutus_cluster_lnl_part(
            nobj, nb, 10, t[0].data_ptr(), t[1].data_ptr(), *[x.data_ptr() for x in t[2:]], 1,
            ws.data_ptr(), ws.numel(), 250, 10, None))


@pytest.mark.gpu
@pytest.mark.parametrize("dim_prior", [1, 0])
def test_cluster_lnl_edge_points_and_objects(dim_prior):
    """C ABI, `brutus_cluster_lnl` against the reference's block (cluster.py:379-407) restated
    in numpy / scipy, on what the online sum treats specially: points of weight -inf, points
    without a finite band that the caller did NOT mark, points with some NaN bands, an object
    that sits exactly on a point (chi2 = 0) with 1, 2, 3 and 4 measurements, objects 1e4
    sigma from every point (every e^(-chi2/2) underflows on its own), 33 measurements
    (the largest power), and a point list that is not a multiple of the kernel's step."""
    import torch
    from scipy.special import logsumexp
    from scipy.stats import chi2 as chisquare
    from brutus_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(21)
    nobj, nb, npts = 300, 32, 1003
    flux = rng.uniform(0.5, 2., size=(npts, nb))
    lnw = rng.normal(size=npts)
    lnw[rng.choice(npts, 60, replace=False)] = -np.inf
    flux[rng.choice(npts, 40, replace=False)] = np.nan                   # no finite band at all
    flux[rng.uniform(size=flux.shape) < 0.02] = np.nan
    phot = flux[rng.randint(0, npts, nobj)].copy()
    phot[~np.isfinite(phot)] = 1.
    phot *= 1. + 0.03 * rng.normal(size=phot.shape)
    ivar = np.full((nobj, nb), 1. / 0.03 ** 2)
    nuse = rng.randint(4, nb + 1, nobj)
    nuse[:8] = (1, 2, 3, 4, 1, 2, 3, 4)
    nuse[8:12] = nb
    for o in range(nobj):
        ivar[o, nuse[o]:] = 0.
    good = np.flatnonzero(np.all(np.isfinite(flux), axis=1) & np.isfinite(lnw))
    phot[:4] = flux[good[:4]]                                            # chi2 = 0 at one point
    phot[4:8] = flux[good[4:8]]
    phot[12:20] *= 300.                                                  # 1e4 sigma off everything
    chi2_p = rng.uniform(0., 3., nobj) * (rng.uniform(size=nobj) < 0.6)
    chi2_p[:4] = 0.
    ndim = nuse + (chi2_p > 0.)
    ndim[8] = 33
    lnorm = rng.normal(size=nobj)
    # the reference's block: nansum over bands, log-pdf, non-finite -> -inf, logsumexp
    with np.errstate(all="ignore"):
        chi2 = np.nansum((phot[None] - flux[:, None]) ** 2 * ivar[None], axis=2) + chi2_p
        lnl_c = (chisquare.logpdf(chi2, ndim) if dim_prior else -0.5 * (chi2 + lnorm))
        lnl_c[~np.isfinite(lnl_c)] = -np.inf
        w = np.where(np.any(np.isfinite(flux), axis=1), lnw, -np.inf)
        want = logsumexp(lnl_c + w[:, None], axis=0)
    dev = torch.device("cuda:0")
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    t = [up(flux), up(lnw), up(phot), up(ivar), up(chi2_p), up(lnorm), up(ndim, np.int32)]
    ws = torch.empty(L.brutus_cluster_workspace_bytes(nobj), dtype=torch.uint8, device=dev)
    out = torch.empty(nobj, dtype=torch.float64, device=dev)
    _lib.check(L.brutus_cluster_lnl(nobj, nb, npts, *[x.data_ptr() for x in t], dim_prior,
                                    ws.data_ptr(), ws.numel(), out.data_ptr(), None))
    got = out.cpu().numpy()
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got)) and fin.sum() > nobj - 8
    assert np.all(got[~fin] == want[~fin])
    assert relerr(want[fin], got[fin]) < 1e-10, np.argmax(np.abs(want[fin] - got[fin]))
    assert want[12:20].max() < -1e6                  # (those objects did exercise the underflow)


@pytest.mark.gpu
def test_cluster_mix_is_numpy_logaddexp_and_a_fixed_order_sum():
    """C ABI, `brutus_cluster_mix`: numpy's `logaddexp` per object -- equal arguments, -inf on
    either or both sides, +inf, NaN -- and a total that has the same bits on every run."""
    import torch
    from brutus_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(2)
    n = 5003
    a = rng.normal(size=n) * 30. - 40.
    b = rng.normal(size=n) * 3. - 20.
    a[:8] = (-np.inf, -np.inf, 0., np.inf, np.nan, -800., 5., -np.inf)
    b[:8] = (-np.inf, -3., 0., 1., 2., -20., np.nan, np.inf)
    b[8:16] = a[8:16]
    ln_fin, ln_fout = np.log(0.9), np.log(0.1)
    with np.errstate(all="ignore"):
        want = np.logaddexp(a + ln_fin, b + ln_fout)
    dev = torch.device("cuda:0")
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    outs = []
    for _ in range(3):
        out = torch.zeros(n + 1, dtype=torch.float64, device=dev)
        _lib.check(L.brutus_cluster_mix(n, ta.data_ptr(), tb.data_ptr(), float(ln_fin),
                                        float(ln_fout), out.data_ptr(), out[n:].data_ptr(), None))
        outs.append(out.cpu().numpy())
    got = outs[0]
    assert np.array_equal(np.isnan(want), np.isnan(got[:n]))
    ok = np.isfinite(want)
    assert np.array_equal(want[~ok & ~np.isnan(want)], got[:n][~ok & ~np.isnan(want)])
    assert np.max(np.abs(want[ok] - got[:n][ok]) / np.maximum(1., np.abs(want[ok]))) < 4e-16
    assert np.isnan(got[n])                                   # (a NaN term: numpy's sum too)
    fin = np.isfinite(want)
    out = torch.zeros(fin.sum() + 1, dtype=torch.float64, device=dev)
    m = int(fin.sum())
    fa, fb = torch.from_numpy(a[fin]).to(dev), torch.from_numpy(b[fin]).to(dev)
    tots = []
    for _ in range(3):
        _lib.check(L.brutus_cluster_mix(m, fa.data_ptr(), fb.data_ptr(), float(ln_fin),
                                        float(ln_fout), out.data_ptr(), out[m:].data_ptr(), None))
        tots.append(float(out[m].item()))
    assert tots[0] == tots[1] == tots[2]
    assert abs(tots[0] - np.sum(want[fin])) < 1e-13 * abs(np.sum(want[fin]))
    assert all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:])


def test_plugin_groups_grow_and_cover_every_slice():
    """`_group_bounds`: consecutive, non-empty, growing groups for any slice / group count."""
    from brutus_amd.cluster import _group_bounds
    assert _group_bounds(15, 3) == [0, 3, 8, 15] and _group_bounds(15, 1) == [0, 15]
    for nsmf in (1, 2, 3, 7, 15, 16, 100, 300):
        for ng in (1, 2, 3, 4, 8, 15, 64, 1000):
            b = _group_bounds(nsmf, ng)
            sizes = np.diff(b)
            assert b[0] == 0 and b[-1] == nsmf and np.all(sizes >= 1), (nsmf, ng, b)
            assert len(sizes) == min(nsmf, ng, 64)
            assert np.all(np.diff(sizes) >= -1)              # (growing, up to rounding)


def test_cache_keys_follow_content_not_identity():
    """The cluster caches are keyed by address, layout and a digest of the CONTENT of the
    catalogue arrays (CPU-only check of the helpers): an in-place edit, a copy, a view with
    other strides and None all give distinct / equal keys as they should; the LRU keeps the
    most recently used entries."""
    import collections
    from brutus_amd import cluster
    a = np.arange(60, dtype=float).reshape(5, 12)
    k0 = cluster._fingerprint(a)
    assert cluster._fingerprint(a) == k0 and cluster._fingerprint(None) is None
    b = a.copy()
    assert cluster._fingerprint(b) != k0 and cluster._fingerprint(b)[3] == k0[3]   # other address, same digest
    a[2, 3] += 1.
    assert cluster._fingerprint(a) != k0                                           # same address, new content
    assert cluster._fingerprint(a[:, ::2])[2] != cluster._fingerprint(a)[2]
    lru = collections.OrderedDict()
    for k in range(4):
        cluster._lru_put(lru, k, str(k), 3)
    assert list(lru) == [1, 2, 3] and cluster._lru_get(lru, 1) == "1" and list(lru) == [2, 3, 1]
    cluster._lru_put(lru, 9, "9", 3)
    assert list(lru) == [3, 1, 9] and cluster._lru_get(lru, 2) is None


@pytest.mark.gpu
@pytest.mark.parametrize("dim_prior", [1, 0])
def test_cluster_magnitude_path_counts_a_band_by_its_magnitude(dim_prior):
    """C ABI: the sum that takes the plug-in's MAGNITUDES (`brutus_cluster_lnl_part_mags`) and
    the flux table (`brutus_cluster_points_grid` + `brutus_cluster_lnl_part`) must agree on
    which points exist -- a point counts if any band's magnitude is finite (reference
    cluster.py:358-364, `np.any(np.isfinite(cmd_sed), axis=1)`), so a point whose bands are
    all +inf (flux exactly 0, a finite number) is dropped, one with a -inf band gives -inf,
    NaN bands fall out of the band sum -- and on the value, which is the reference block's."""
    import torch
    from scipy.special import logsumexp
    from scipy.stats import chi2 as chisquare
    from brutus_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(3)
    nobj, nb, neep, nsmf = 64, 8, 160, 3
    nrow = neep * nsmf
    mags = rng.uniform(14., 16., size=(nrow, nb))
    mags[rng.choice(nrow, 25, replace=False)] = np.inf               # every band +inf: no such point
    mags[rng.choice(nrow, 10, replace=False)] = np.nan
    some = rng.uniform(size=mags.shape)
    mags[some < 0.02] = np.inf                                       # a band at +inf: flux 0, counted in chi2
    mags[(some > 0.02) & (some < 0.03)] = np.nan
    mags[7, 2] = -np.inf                                             # infinite flux: chi2 = inf
    lnw_eep = rng.normal(size=neep)
    lnw_eep[rng.choice(neep, 9, replace=False)] = -np.inf
    lnw_smf = rng.normal(size=nsmf)
    src = np.sort(rng.choice(nrow, 400, replace=False)).astype(np.int32)
    src = np.union1d(src, [7]).astype(np.int32)
    npts = src.size
    with np.errstate(all="ignore"):
        flux = 10. ** (-0.4 * mags[src])
    phot = 10. ** (-0.4 * rng.uniform(14., 16., size=(nobj, nb)))
    ivar = 1. / (0.05 * phot) ** 2
    chi2_p = rng.uniform(0., 2., nobj)
    lnorm = rng.normal(size=nobj)
    ndim = np.full(nobj, nb + 1, dtype=np.int32)
    with np.errstate(all="ignore"):
        chi2 = np.nansum((phot[None] - flux[:, None]) ** 2 * ivar[None], axis=2) + chi2_p
        lnl_c = (chisquare.logpdf(chi2, ndim) if dim_prior else -0.5 * (chi2 + lnorm))
        lnl_c[~np.isfinite(lnl_c)] = -np.inf
        w = lnw_eep[src % neep] + lnw_smf[src // neep]
        w = np.where(np.any(np.isfinite(mags[src]), axis=1), w, -np.inf)
        want = logsumexp(lnl_c + w[:, None], axis=0)
    dev = torch.device("cuda:0")
    up = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    t_src, t_mags, t_eep, t_smf = up(src, np.int32), up(mags), up(lnw_eep), up(lnw_smf)
    obj = [up(phot), up(ivar), up(chi2_p), up(lnorm), up(ndim, np.int32)]
    ws = torch.empty(L.brutus_cluster_workspace_bytes(nobj), dtype=torch.uint8, device=dev)
    nchunk = L.brutus_cluster_chunks()
    out = {}
    for path in ("mags", "flux"):
        res = torch.empty(nobj, dtype=torch.float64, device=dev)
        if path == "mags":
            _lib.check(L.brutus_cluster_lnl_part_mags(
                nobj, nb, npts, neep, t_src.data_ptr(), t_mags.data_ptr(), t_eep.data_ptr(),
                t_smf.data_ptr(), *[x.data_ptr() for x in obj], dim_prior, ws.data_ptr(), ws.numel(),
                0, nchunk, None))
        else:
            t_flux = torch.empty(npts * nb, dtype=torch.float64, device=dev)
            t_lnw = torch.empty(npts, dtype=torch.float64, device=dev)
            _lib.check(L.brutus_cluster_points_grid(npts, nb, neep, t_src.data_ptr(), t_mags.data_ptr(),
                                                    t_eep.data_ptr(), t_smf.data_ptr(),
                                                    t_flux.data_ptr(), t_lnw.data_ptr(), None))
            _lib.check(L.brutus_cluster_lnl_part(
                nobj, nb, npts, t_flux.data_ptr(), t_lnw.data_ptr(), *[x.data_ptr() for x in obj],
                dim_prior, ws.data_ptr(), ws.numel(), 0, nchunk, None))
        _lib.check(L.brutus_cluster_lnl_merge(nobj, nchunk, ws.data_ptr(), ws.numel(),
                                              res.data_ptr(), None))
        torch.cuda.synchronize()
        out[path] = res.cpu().numpy()
    assert np.all(np.isfinite(want))
    assert relerr(want, out["flux"]) < 1e-10
    assert relerr(want, out["mags"]) < 1e-10
    assert relerr(out["flux"], out["mags"]) < 1e-13
