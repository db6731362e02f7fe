"""GPU: `lnpost` + resampling on the device (brutus_post_batch) against the
reference semantics.  The random stream is brutus_amd/rng.PhiloxRandomState, a
valid `rstate` object for the reference/oracle, so the device result is compared
with the ORACLE run on the same `rstate` -- indices bit-exact."""
import os

import numpy as np
import pytest

from helpers import galprior, relerr

pytestmark = pytest.mark.gpu


def test_device_rng_matches_specification():
    import torch
    from brutus_amd import _lib
    from brutus_amd.rng import philox_normal, philox_uniform
    L = _lib.lib()
    from brutus_amd.rng import ZIG_X
    slow = 0
    for seed, start in ((0, 0), (12345, 7), (2 ** 63 + 11, 2 ** 33 + 5)):
        n = 400001
        z = torch.empty(n, dtype=torch.float64, device="cuda")
        u = torch.empty(n, dtype=torch.float64, device="cuda")
        _lib.check(L.brutus_debug_rng(seed, start, n, z.data_ptr(), u.data_ptr(), None))
        torch.cuda.synchronize()
        idx = np.arange(start, start + n, dtype=np.uint64)
        assert np.array_equal(u.cpu().numpy(), philox_uniform(seed, idx))
        zr = philox_normal(seed, idx)
        # the rectangle case of the ziggurat (99.57 %) is a table look-up and one
        # multiplication: bit-equal.  Wedge and tail take exp / ln (ocml vs numpy, last
        # bits), and a wrong accept / reject decision there would move a deviate by O(1)
        zd = z.cpu().numpy()
        err = np.abs(zd - zr) / np.abs(zr)
        print("normals: max rel err %.2e, bit-equal %.5f, beyond R %d"
              % (err.max(), np.mean(zd == zr), np.sum(np.abs(zr) > ZIG_X[1])))
        assert err.max() < 1e-15
        assert np.mean(zd == zr) > 0.999
        slow += int(np.sum(np.abs(zr) > ZIG_X[1]))
    assert slow > 20          # the tail loop was exercised (2 Phi(-R) = 5.4e-5 per normal)


def _post_params(**kw):
    from brutus_amd import _lib
    from brutus_amd.galprior import device_params
    pp = _lib.PostParams()
    for k, v in device_params(**kw).items():
        if isinstance(v, tuple):
            getattr(pp, k)[:] = list(v)
        else:
            setattr(pp, k, v)
    pp.has_feh = pp.has_loga = 1
    return pp


def _lnp_err(ref, got):
    """|difference| of two ln-prior vectors relative to max(|ref|, 1) (a ln prior crosses
    zero: its last-bit errors are absolute), same -inf pattern required."""
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(ref[~fin], got[~fin])
    return float(np.max(np.abs(ref[fin] - got[fin]) / np.maximum(np.abs(ref[fin]), 1.)))


def test_device_galprior_matches_host():
    import torch
    from brutus_amd import _lib
    from brutus_amd.galprior import gal_lnprior
    L = _lib.lib()
    rng = np.random.RandomState(3)
    n = 5000
    d = 10. ** rng.uniform(-2, 1.5, n)
    lab = np.zeros(n, dtype=[("feh", "f8"), ("loga", "f8")])
    lab["feh"] = rng.uniform(-3, 0.6, n)
    lab["loga"] = rng.uniform(7.5, 10.2, n)
    # both Galactocentric frames: the reference's astropy route (default) and the simple one
    for frame in ("astropy", "simple"):
        for coord in ((204.7, -19.2), (0., 90.), (33., 2.), (0.02, -0.01)):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
            out = torch.empty(n, dtype=torch.float64, device="cuda")
            td, tc, tf, tl = t(d), t(np.array(coord)), t(lab["feh"]), t(lab["loga"])
            _lib.check(L.brutus_debug_galprior(_post_params(frame=frame), n, td.data_ptr(),
                                               tc.data_ptr(), tf.data_ptr(), tl.data_ptr(),
                                               out.data_ptr(), None))
            torch.cuda.synchronize()
            ref = gal_lnprior(d, coord, labels=lab, frame=frame)
            assert relerr(ref, out.cpu().numpy()) < 1e-12, (frame, coord)
            # ... and in the form the Monte Carlo sample loop evaluates it (per-object
            # constant block, R^2(d) as a quadratic, table-driven halo power law), plus the
            # plain form it falls back to when the parameters do not admit the table
            # (eta = 40: the series' dropped term is too large for the table form, the
            # library must take the plain form by itself -- and a tiny Rs_halo likewise)
            for kw in (dict(), dict(eta_halo=40.), dict(Rs_halo=0.03)):
                out.fill_(0.)
                _lib.check(L.brutus_debug_galprior_mc(_post_params(frame=frame, **kw), n, td.data_ptr(),
                                                      tc.data_ptr(), tf.data_ptr(), tl.data_ptr(),
                                                      out.data_ptr(), None))
                refk = gal_lnprior(d, coord, labels=lab, frame=frame, **kw)
                assert _lnp_err(refk, out.cpu().numpy()) < 1e-12, (frame, coord, kw)


def test_sightline_table_matches_closed_form():
    """The per-item sightline table of the Monte Carlo kernels (post_kernels.hpp: degree-7 fits of
    the three density components in s = 1 / d^2, sixteen intervals per octave) against the closed
    form it replaces, on 10^6 distances per sightline: sorted distances (every workgroup's 256
    distances inside its window: the table must serve all of them within 2 kpc), the neighbourhood of the
    Z = 0 crossing where |Z| has its kink, and unsorted distances over six decades (most of them
    outside the window of their workgroup: closed form)."""
    import torch
    from brutus_amd import _lib
    from brutus_amd.galprior import _frame
    L = _lib.lib()
    rng = np.random.RandomState(11)
    n = 1 << 20
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    feh, loga = rng.uniform(-3, 0.6, n), rng.uniform(7.5, 10.2, n)
    tf, tl = t(feh), t(loga)

    def both(d, coord):
        td, tc = t(d), t(np.array(coord))
        ref = torch.empty(n, dtype=torch.float64, device="cuda")
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        used = torch.zeros(n, dtype=torch.int32, device="cuda")
        _lib.check(L.brutus_debug_galprior_mc(_post_params(), n, td.data_ptr(), tc.data_ptr(), tf.data_ptr(),
                                              tl.data_ptr(), ref.data_ptr(), None))
        _lib.check(L.brutus_debug_galprior_sl(_post_params(), n, td.data_ptr(), tc.data_ptr(), tf.data_ptr(),
                                              tl.data_ptr(), out.data_ptr(), used.data_ptr(), None))
        torch.cuda.synchronize()
        return ref.cpu().numpy(), out.cpu().numpy(), used.cpu().numpy().astype(bool)

    def gap(ref, out):
        # (-inf where the age prior excludes the model: in both or in neither)
        fin = np.isfinite(ref)
        assert np.array_equal(fin, np.isfinite(out)) and np.array_equal(ref[~fin], out[~fin]) and fin.any()
        return float(np.max(np.abs(out[fin] - ref[fin])))

    M, off = _frame("astropy", 8.2, 0.025)
    worst = 0.
    for coord in ((204.7, -19.2), (0., 90.), (0., -90.), (33., 2.), (0.02, -0.01), (0., -0.17), (180., -5.),
                  (90., -30.)):
        d = np.sort(10. ** rng.uniform(-2.5, 2.3, n))
        ref, out, used = both(d, coord)
        # (far above the plane an interval spans several scale heights of the disks: the builder
        # certifies every fit against the closed form and leaves those intervals to it)
        assert used[d < 2.].all() and used.mean() > 0.5, (coord, used.mean())
        worst = max(worst, gap(ref, out))
        # the crossing of the plane Z = 0, if this sightline has one: a million distances within
        # +-2 % of it (one or two intervals of the table around the kink)
        ell, b = np.deg2rad(coord)
        uz = (M @ np.array([np.cos(b) * np.cos(ell), np.cos(b) * np.sin(ell), np.sin(b)]))[2]
        if uz != 0. and -off[2] / uz > 1e-3:
            dk = -off[2] / uz
            ref, out, used = both(np.sort(dk * (1. + rng.uniform(-0.02, 0.02, n))), coord)
            assert used.all() or dk > 2., coord
            worst = max(worst, gap(ref, out))
        # unsorted: the window of a workgroup covers a factor 22 below its nearest distance
        ref, out, used = both(10. ** rng.uniform(-3, 3, n), coord)
        assert 0.05 < used.mean() < 0.9, used.mean()
        assert gap(ref[~used], out[~used]) < 1e-13, coord      # (the same closed form, inlined elsewhere)
        worst = max(worst, gap(ref, out))
    # |ln prior (table) - ln prior (closed form)|: every component within ~1e-11 of itself (SL_TOL at
    # the point where the interpolation errs most), whatever the label weights make of them
    assert worst < 1e-10, worst
    # parameters that do not admit the halo table: refused (the kernels then run without either table)
    td, tc = t(np.ones(n)), t(np.array((10., 10.)))
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    used = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert L.brutus_debug_galprior_sl(_post_params(eta_halo=40.), n, td.data_ptr(), tc.data_ptr(), tf.data_ptr(),
                                      tl.data_ptr(), out.data_ptr(), used.data_ptr(), None) != 0


def _setup(nmodel=6000, nstar=9, seed=31):
    from brutus_amd import fitting, synth
    from oracle import brutus_oracle as O
    models, labels, lmask = synth.make_mist_like_grid(nmodel, 8, seed=seed)
    st = synth.make_stars(models, nstar, seed=seed + 1)
    st["mask"][1, 2] = False
    BF = fitting.BruteForce(models, labels, lmask)
    lnprior = O.static_lnprior(labels, lmask)
    return BF, models, labels, st, lnprior


NAMES = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds dreds "
         "logwts").split()


def _compare(dev, ref, tag):
    assert np.array_equal(dev[0], ref[0]), "%s: resampled indices" % (tag,)
    for n, a, b in zip(NAMES[1:], ref[1:], dev[1:]):
        if n in ("reds", "dreds"):
            # a drawn Av / Rv is a0 + L z: a sum of O(1) terms that may land near zero, so the
            # error is measured against the size of the terms (the array's largest value)
            a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
            err = float(np.max(np.abs(a - b)) / np.max(np.abs(a))) if a.size else 0.
        else:
            err = relerr(a, b)
        assert err < 1e-8, (tag, n, err)


def test_device_lnpost_shared_stream_vs_oracle():
    """One sequential PhiloxRandomState over all objects, batches of 4."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup()
    BF.batch_size = 4
    rs = PhiloxRandomState(2024)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60,
                       rstate=rs))
    ro = PhiloxRandomState(2024)
    for i in range(len(dev)):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=20, Ndraws=60)
        _compare(dev[i], ref, i)
    # the device advanced the caller's rstate exactly like the host would have
    assert (rs.n_normal, rs.n_uniform) == (ro.n_normal, ro.n_uniform)


@pytest.mark.parametrize("stream", ["philox", "numpy"])
def test_fit_writes_whole_batches_like_rows(tmp_path, stream):
    """`fit()` takes the device stage's results a batch at a time (`_RowBlock`: arrays in the file's
    layout) instead of 13-tuples per object: the file must equal, dataset by dataset, the one
    written row by row from `_fit`'s tuples -- ragged last batch, an object without parallax,
    `running_io` on and off."""
    from brutus_amd import h5io
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    BF, models, labels, st, lnprior = _setup()
    n = st["flux"].shape[0]
    BF.batch_size = 5
    mk = (lambda: PhiloxRandomState(5)) if stream == "philox" else (lambda: np.random.RandomState(5))
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], Nmc_prior=20,
              lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60)
    for rio, dar in ((True, True), (False, True), (True, False)):
        path = os.path.join(str(tmp_path), "blk_%s_%d_%d" % (stream, rio, dar))
        BF.fit(st["flux"], st["err"], st["mask"], np.arange(n), path, rstate=mk(), verbose=False,
               running_io=rio, save_dar_draws=dar, **kw)
        assert BF._yield_row_blocks is False
        ref = h5io.ResultsFile(path + "_rows.h5", n, 60, np.arange(n), dar, running_io=rio)
        (d, e, m, _, coords, lnp_, lng, lnd, avg, wt, _) = BF._setup(
            st["flux"], st["err"], st["mask"], np.arange(n), parallax=st["parallax"],
            parallax_err=st["parallax_err"], data_coords=st["coords"], lngalprior=gal_lnprior)
        for i, row in enumerate(BF._fit(d, e, m, parallax=st["parallax"], parallax_err=st["parallax_err"],
                                        lnprior=lnp_, lngalprior=lng, lndustprior=lnd, av_gauss=avg,
                                        wt_thresh=wt, data_coords=coords, Nmc_prior=20, Ndraws=60,
                                        return_distreds=dar, rstate=mk())):
            assert isinstance(row, tuple) and len(row) == (13 if dar else 9)
            ref.write_row(i, row)
        ref.close()
        for k in h5io.list_datasets(path + ".h5"):
            a, b = h5io.read_dataset(path + ".h5", k), h5io.read_dataset(path + "_rows.h5", k)
            assert a.dtype == b.dtype and np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), (k, rio)


@pytest.mark.parametrize("nfilt", [40, 64])
def test_more_than_32_bands_fit_vs_oracle(nfilt):
    """33 - 64 bands unmasked at once: `_fit` takes the full-grid pipeline, cuts on the device and
    hands dense records to the same device `lnpost` as any other band count -- against the oracle
    driven by the same counter-based stream: resampled indices bit-exact, two batches (the second
    ragged), a masked pair of bands, a negative flux."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    models, labels, lmask = synth.make_grid(6000, nfilt, seed=40 + nfilt)
    st = synth.make_stars(models, 5, seed=41)
    st["mask"][1, [3, nfilt - 2]] = False
    st["flux"][2, 5] = -abs(st["flux"][2, 5])
    BF = fitting.BruteForce(models, labels, lmask)
    lnprior = O.static_lnprior(labels, lmask)
    BF.batch_size = 3
    rs = PhiloxRandomState(9)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60, rstate=rs))
    ro = PhiloxRandomState(9)
    for i in range(len(dev)):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=20, Ndraws=60)
        _compare(dev[i], ref, i)
    assert (rs.n_normal, rs.n_uniform) == (ro.n_normal, ro.n_uniform)


@pytest.mark.parametrize("lims", [dict(rvlim=(3.32, 3.32), rv_gauss=(3.32, 1e-6)),
                                  dict(rvlim=(3.32, 3.32), rv_gauss=(3.32, 1e-6), avlim=(-30., 50.))])
def test_device_lnpost_unseen_normal_runs_vs_oracle(lims):
    """k_post_mc generates only the runs of normals the integrand can see: with Rv pinned every
    record is in the Rv bounds whatever its normals are (third run skipped), with Av limits far
    from every record's Av as well (second run skipped too).  Same stream positions, same sums:
    the oracle draws all three runs."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup()
    BF.batch_size = 4
    rs = PhiloxRandomState(31)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60,
                       rstate=rs, **lims))
    ro = PhiloxRandomState(31)
    for i in range(len(dev)):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=20, Ndraws=60, **lims)
        _compare(dev[i], ref, (sorted(lims), i))
    assert (rs.n_normal, rs.n_uniform) == (ro.n_normal, ro.n_uniform)


def test_randomised_fits_vs_oracle():
    """Thirty random cases of tools/fuzz_lnpost.py (grid size, bands, stars, S/N, parallaxes, masks,
    Nmc_prior 7 - 70, Ndraws, Av / Rv limits and priors incl. the ones that let k_post_mc skip runs
    of normals, counter-based and numpy streams): `_fit` against the oracle with the same stream.
    (675 cases of nine other seeds ran clean in round 5: profiles/r05_fuzz.txt.)"""
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import fuzz_lnpost as F
    rng = np.random.RandomState(21)
    for c in range(30):
        models, labels, lmask, st, kw, stream, desc = F.case(rng)
        try:
            F.check(models, labels, lmask, st, kw, stream, 500 + c)
        except AssertionError as e:
            raise AssertionError("case %d %s: %s" % (c, desc, e))


def _steep_halo_hook():
    """The built-in prior with a halo too steep for the table form of its power law (the
    library must run the plain form: `k_post_mc<false>` / `k_post_mc_arr<false>`)."""
    from brutus_amd.galprior import device_params, gal_lnprior

    def steep(dists, coord, labels=None, **kw):
        return gal_lnprior(dists, coord, labels=labels, eta_halo=40., **kw)
    steep.broadcasts_labels = True
    steep.device_params = lambda **kw: device_params(**dict(dict(eta_halo=40.), **kw))
    return steep


@pytest.mark.parametrize("stream", ["philox", "numpy"])
def test_device_lnpost_plain_halo_form_vs_oracle(stream):
    """As test_device_lnpost_shared_stream_vs_oracle with prior parameters that do not admit
    the halo table, through both Monte Carlo kernels (counter-based and numpy streams)."""
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    hook = _steep_halo_hook()
    BF, models, labels, st, lnprior = _setup()
    BF.batch_size = 4
    mk = (lambda: PhiloxRandomState(77)) if stream == "philox" else (lambda: np.random.RandomState(77))
    rs = mk()
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=hook, data_coords=st["coords"], Ndraws=60, rstate=rs))
    ro = mk()
    for i in range(len(dev)):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, hook, Nmc_prior=20, Ndraws=60)
        _compare(dev[i], ref, (stream, i))


def test_full_size_fit_vs_oracle():
    """The whole per-object result at the bench's size and defaults (750k x 12,
    Nmc_prior=50, Ndraws=250, ~10^5 models per object through the Monte Carlo
    integral): fused scan + device `lnpost` against the oracle -- C `loglike`
    followed by the numpy `lnpost` / resampling -- driven by the same
    PhiloxRandomState."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    from oracle import c_oracle
    models, labels, lmask = synth.make_mist_like_grid(750000, 12)
    st = synth.make_stars(models, 2, seed=2)
    lnprior = O.static_lnprior(labels, lmask)
    BF = fitting.BruteForce(models, labels, lmask)
    BF.batch_size = 2
    rs = PhiloxRandomState(77)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], rstate=rs,
                       Nmc_prior=50, Ndraws=250))      # fit()'s defaults (_fit's differ)
    ro = PhiloxRandomState(77)
    py_loglike = O.loglike
    O.loglike = lambda *a, return_vals=True, **k: c_oracle.loglike(*a, **k)
    try:
        for i in range(2):
            ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                             labels, st["coords"][i], st["parallax"][i],
                             st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=50,
                             Ndraws=250)
            assert np.array_equal(dev[i][0], ref[0]), "resampled indices, object %d" % i
            for n, a, b in zip(NAMES[1:], ref[1:], dev[i][1:]):
                assert relerr(a, b) < 1e-6, (i, n, relerr(a, b))
    finally:
        O.loglike = py_loglike
    assert (rs.n_normal, rs.n_uniform) == (ro.n_normal, ro.n_uniform)


def test_device_lnpost_per_object_and_host_agree():
    """seed0 + 'philox': per-object streams; the device path, the host path with
    the same rstate objects and the oracle all agree, for any batching."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nstar=7, seed=41)
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], Nmc_prior=15,
              lnprior=lnprior, lngalprior=gal_lnprior, data_coords=st["coords"],
              Ndraws=40, seed0=500, rstate_per_object="philox")
    BF.batch_size = 3
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], **kw))
    BF.batch_size = 7
    dev2 = list(BF._fit(st["flux"], st["err"], st["mask"], **kw))
    BF.device_lnpost = False
    host = list(BF._fit(st["flux"], st["err"], st["mask"], **kw))
    for i in range(7):
        assert np.array_equal(dev[i][0], dev2[i][0])
        _compare(dev[i], host[i], "host %d" % i)
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], PhiloxRandomState(500 + i), gal_lnprior,
                         Nmc_prior=15, Ndraws=40)
        _compare(dev[i], ref, "oracle %d" % i)


def test_device_lnpost_nsel_max_clip():
    """A tiny mem_lim makes Nsel_max smaller than the second-cut selection:
    those objects are clipped to the Nsel_max best, best first, by a device
    sort (fitting.py:1029-1036) and still match the oracle."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nstar=4, seed=51)
    mem_lim = 20 * 4e-4 * 300          # Nsel_max = 300
    rs = PhiloxRandomState(9)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=30,
                       rstate=rs, mem_lim=mem_lim))
    ro = PhiloxRandomState(9)
    for i in range(4):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=20, Ndraws=30,
                         mem_lim=mem_lim)
        _compare(dev[i], ref, i)


def test_device_lnpost_vs_reference_golden():
    """tests/golden/fit_philox.npz: the UPSTREAM `_fit` run with
    `rstate=PhiloxRandomState(31337)` and `lngalprior=brutus_amd.galprior.
    gal_lnprior` (tools/gen_golden.py).  The device lnpost must reproduce it:
    indices bit-exact, floats <=1e-5, and leave the rstate where the reference
    left it."""
    import os
    from helpers import GOLDEN
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    z = np.load(os.path.join(GOLDEN, "fit_philox.npz"))
    models, labels, lmask = synth.make_mist_like_grid(int(z["grid_nmodel"]),
                                                      int(z["grid_nfilt"]),
                                                      seed=int(z["grid_seed"]))
    BF = fitting.BruteForce(models, labels, lmask)
    BF.batch_size = 3
    rs = PhiloxRandomState(int(z["seed"]))
    outs = list(BF._fit(z["flux"], z["err"], z["mask"], parallax=z["parallax"],
                        parallax_err=z["parallax_err"], Nmc_prior=25,
                        lnprior=z["lnprior"], lngalprior=gal_lnprior,
                        data_coords=z["coords"], rstate=rs, Ndraws=80))
    for i, out in enumerate(outs):
        assert np.array_equal(out[0], z["sidxs"][i]), i
        for n, got in zip(NAMES[1:], out[1:]):
            assert relerr(z[n][i], got) < 1e-5, (i, n, relerr(z[n][i], got))
    assert (rs.n_normal, rs.n_uniform) == (int(z["n_normal"]), int(z["n_uniform"]))


@pytest.mark.parametrize("stream", ["philox", "numpy"])
def test_lnprior_ext_on_the_device_vs_oracle(stream):
    """`lnprior_ext` with the built-in priors: the constraints are added to lnlike over the whole
    grid ON THE DEVICE (full-grid pipeline), the cut and `lnpost` follow there -- against the oracle
    assembled the way the reference's star loop does (fitting.py:1995-2012), same stream: resampled
    indices bit-exact.  Two label columns, a NaN mean and a zero width (both skipped), two batches."""
    from scipy.special import logsumexp
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nmodel=5000, nstar=6, seed=53)
    BF.batch_size = 4
    ext = {"feh": np.array([[-0.3, 0.2], [np.nan, 0.2], [0.1, 0.0], [-1.0, 0.5], [0.2, 0.3], [-0.5, 0.1]]),
           "loga": np.array([[9.5, 0.3], [9.0, 0.2], [np.nan, 1.0], [9.8, 0.05], [8.7, 0.4], [9.2, 0.0]])}
    mk = (lambda: PhiloxRandomState(5)) if stream == "philox" else (lambda: np.random.RandomState(5))
    rs = mk()
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=15, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=30,
                       lnprior_ext=ext, rstate=rs))
    ro = mk()
    for i in range(6):
        par, perr = st["parallax"][i], st["parallax_err"][i]
        res = list(O.loglike(st["flux"][i], st["err"][i], st["mask"][i], models,
                             av_gauss=(0., 1e6), parallax=par, parallax_err=perr,
                             return_vals=True))
        for k in ("feh", "loga"):
            mean, std = ext[k][i]
            if np.isfinite(mean) and std > 0:
                res[0] = res[0] - 0.5 * ((labels[k] - mean) ** 2 * (1. / std ** 2)
                                         + np.log(2. * np.pi * std ** 2))
        sel, cov, lnp, dists, reds, dreds, logwts = O.lnpost(
            tuple(res), parallax=par, parallax_err=perr, coord=st["coords"][i],
            Nmc_prior=15, lnprior=lnprior, wt_thresh=1e-3, lngalprior=gal_lnprior,
            lndustprior=None, dlabels=labels, avlim=(0., 20.), rvlim=(1., 8.), rstate=ro,
            apply_av_prior=False, mem_lim=8000.)
        wt = np.exp(lnp - logsumexp(lnp))
        wt /= wt.sum()
        idxs = ro.choice(len(sel), size=30, p=wt)
        assert np.array_equal(dev[i][0], sel[idxs]), i
        assert relerr(lnp[idxs], dev[i][6]) < 1e-8
        # (the oracle's stream has to take the second resampling stage too, like the device did)
        for j, idx in enumerate(idxs):
            w = np.exp(logwts[idx] - logsumexp(logwts[idx]))
            w /= w.sum()
            ro.choice(15, p=w)


def test_lnprior_ext_vs_oracle():
    """External per-object Gaussian label constraints (fitting.py:1993-2009): the
    full-grid device outputs + host cut, against the oracle pieces assembled the
    way the reference's star loop does."""
    from scipy.special import logsumexp
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nmodel=4000, nstar=4, seed=51)
    ext = {"feh": np.array([[-0.3, 0.2], [np.nan, 0.2], [0.1, 0.0], [-1.0, 0.5]])}
    with pytest.raises(ValueError, match="do not match"):
        list(BF._fit(st["flux"], st["err"], st["mask"], lnprior=lnprior, lngalprior=galprior,
                     data_coords=st["coords"], lnprior_ext={"nope": ext["feh"]},
                     rstate=np.random.RandomState(1)))
    rs = np.random.RandomState(5)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=15, lnprior=lnprior,
                       lngalprior=galprior, data_coords=st["coords"], Ndraws=30,
                       lnprior_ext=ext, rstate=rs))
    ro = np.random.RandomState(5)
    for i in range(4):
        par, perr = st["parallax"][i], st["parallax_err"][i]
        res = list(O.loglike(st["flux"][i], st["err"][i], st["mask"][i], models,
                             av_gauss=(0., 1e6), parallax=par, parallax_err=perr,
                             return_vals=True))
        mean, std = ext["feh"][i]
        if np.isfinite(mean) and std > 0:
            res[0] = res[0] - 0.5 * ((labels["feh"] - mean) ** 2 / std ** 2
                                     + np.log(2. * np.pi * std ** 2))
        sel, cov, lnp, dists, reds, dreds, logwts = O.lnpost(
            tuple(res), parallax=par, parallax_err=perr, coord=st["coords"][i],
            Nmc_prior=15, lnprior=lnprior, wt_thresh=1e-3, lngalprior=galprior,
            lndustprior=None, dlabels=labels, avlim=(0., 20.), rvlim=(1., 8.), rstate=ro,
            apply_av_prior=False, mem_lim=8000.)
        wt = np.exp(lnp - logsumexp(lnp))
        wt /= wt.sum()
        idxs = ro.choice(len(sel), size=30, p=wt)
        assert np.array_equal(dev[i][0], sel[idxs]), i
        assert relerr(lnp[idxs], dev[i][6]) < 1e-8
        assert abs(dev[i][7] - logsumexp(lnp)) < 1e-8 * abs(logsumexp(lnp))
        # second resampling stage consumes the stream too
        wts = np.exp(logwts[idxs] - logsumexp(logwts[idxs], axis=1)[:, None])
        for j in range(30):
            ro.choice(15, p=wts[j] / wts[j].sum())


def test_fit_options_device_equals_host(tmp_path):
    """`save_dar_draws=False` + `running_io=False` through the public `fit()`: the
    device `lnpost` mode writes the file the host stage writes for the same
    PhiloxRandomState, and the sample datasets are absent."""
    import os
    from brutus_amd import h5io
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    BF, models, labels, st, lnprior = _setup(nmodel=5000, nstar=7, seed=61)
    BF.batch_size = 3
    files = []
    for mode in (True, False):
        BF.device_lnpost = mode
        path = os.path.join(str(tmp_path), "dev" if mode else "host")
        BF.fit(st["flux"], st["err"], st["mask"], np.arange(7), path,
               parallax=st["parallax"], parallax_err=st["parallax_err"],
               data_coords=st["coords"], lngalprior=gal_lnprior, Nmc_prior=12, Ndraws=25,
               save_dar_draws=False, running_io=False, rstate=PhiloxRandomState(9),
               verbose=False)
        files.append(path + ".h5")
    BF.device_lnpost = True
    names = set(h5io.list_datasets(files[0]))
    assert names == set(h5io.list_datasets(files[1]))
    assert not any(n.startswith("samps_") for n in names)
    assert np.array_equal(h5io.read_dataset(files[0], "model_idx"),
                          h5io.read_dataset(files[1], "model_idx"))
    for n in ("ml_scale", "ml_av", "ml_rv", "ml_cov_sar", "obj_log_post", "obj_log_evid",
              "obj_chi2min", "obj_Nbands"):
        a, b = h5io.read_dataset(files[0], n), h5io.read_dataset(files[1], n)
        assert np.allclose(a, b, rtol=2e-6, atol=0), n


def test_user_dust_prior_hook_vs_oracle():
    """A user `lndustprior(dists, coord, avs, dustfile=)` hook with `av_gauss=None`
    (fitting.py:1010, 1085, 1396-1398): host stage on device records, and
    `logl_dim_prior=False` + custom `avlim`, against the oracle."""
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nmodel=4000, nstar=5, seed=71)
    BF.batch_size = 2

    def dust(dists, coord, avs, dustfile=None):
        return -0.5 * ((avs - 0.4 * np.log1p(dists)) / 0.6) ** 2

    kw = dict(Nmc_prior=12, Ndraws=25, avlim=(0., 6.))
    rs = np.random.RandomState(3)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], lnprior=lnprior, lngalprior=galprior,
                       lndustprior=dust, data_coords=st["coords"], logl_dim_prior=False,
                       rstate=rs, **kw))
    ro = np.random.RandomState(3)
    for i in range(5):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior, labels,
                         st["coords"][i], st["parallax"][i], st["parallax_err"][i], ro,
                         galprior, lndustprior=dust, dim_prior=False, **kw)
        _compare(dev[i], ref, i)


def test_device_psd_repair_on_crafted_records():
    """The crafted non-positive-definite set of tests/golden/psd.npz (the reference runs
    its repair loop, fitting.py:1039-1065, on 128 of the 160 matrices; the host stage is
    pinned to it in tests/test_priors_golden.py) fed to the DEVICE post stage as first-cut
    records: same repaired covariances, same resampled indices as the oracle driven by the
    same PhiloxRandomState."""
    import os
    import torch
    from helpers import GOLDEN
    from brutus_amd import _lib, fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    z = np.load(os.path.join(GOLDEN, "psd.npz"))
    n = len(z["scale"])
    lab = np.zeros(n, dtype=[('feh', 'f8'), ('loga', 'f8')])
    lab['feh'] = z["feh"]
    lab['loga'] = 9.5
    models, _, _ = synth.make_grid(n, 6, seed=1)          # only to own an engine
    eng = fitting._Engine(fitting.DeviceGrid(models), max_batch=2)
    dev = eng.grid.device
    # parallax S/N < 4: the reference's first cut then rests on lnlike alone (pdf.py:209),
    # so all 160 crafted models are first-cut records on both sides; the Monte Carlo stage
    # still applies the parallax likelihood
    coord, par, perr = np.array([[50., 20.]]), np.array([1.2]), np.array([0.5])
    Nmc, Ndraws = 25, 80
    # records of one object: every model selected by the first cut, ascending order
    vals = np.empty((_lib.NVALS, n))
    vals[0], vals[1], vals[2], vals[3], vals[4] = z["lnl"], z["chi2"], z["scale"], z["av"], z["rv"]
    ic = z["icov"]
    for k, (a, b) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        vals[5 + k] = ic[:, a, b]
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    # indexed records (include/brutus_amd.h): the values sit in shuffled columns of a
    # buffer that also holds columns no record refers to
    cap = n + 37
    slot = np.random.RandomState(3).permutation(cap)[:n]
    spread = np.full((_lib.NVALS, cap), np.nan)
    spread[:, slot] = vals
    idx_pad = np.zeros(cap, dtype=np.int64)
    idx_pad[:n] = np.arange(n)
    slot_pad = np.zeros(cap, dtype=np.int64)
    slot_pad[:n] = slot
    rec = fitting.Records(up(idx_pad, torch.int32), up(slot_pad, torch.int32),
                          up(spread, torch.float64), up(np.array([0, n]), torch.int64))
    statics = (up(z["lnprior"], torch.float64), up(lab['feh'], torch.float64),
               up(lab['loga'], torch.float64))
    pp = _lib.PostParams()
    pp.nmc, pp.ndraws, pp.return_distreds = Nmc, Ndraws, 1
    pp.has_feh = pp.has_loga = 1
    pp.wt_thresh = 1e-3
    pp.avlim[:] = [0., 20.]
    pp.rvlim[:] = [1., 8.]
    pp.nsel_max = 400000
    pp.per_object, pp.object0, pp.seed = 0, 0, 99
    pp.normal_base = pp.uniform_base = 0
    for k, val in gal_lnprior.device_params().items():
        if isinstance(val, tuple):
            getattr(pp, k)[:] = list(val)
        else:
            setattr(pp, k, val)
    out_idx, out_vals, star_out, flags, nbase = eng.post_batch_device(
        rec, 1, statics, coord, par, perr, pp)
    res = (z["lnl"].copy(), 8, z["chi2"].copy(), z["scale"].copy(), z["av"].copy(),
           z["rv"].copy(), z["icov"].copy())
    ref = O.finish_star(res, z["lnprior"].copy(), lab, coord[0], par[0], perr[0],
                        PhiloxRandomState(99), gal_lnprior, Nmc_prior=Nmc, Ndraws=Ndraws)
    assert flags[0] == 0
    v = out_vals[0]
    assert np.array_equal(out_idx[0].astype(np.int64), ref[0])
    # at least a third of the drawn models went through the repair loop
    assert np.mean(z["not_psd_before"][ref[0]]) > 0.3
    assert relerr(ref[4], v[:, 3:12].reshape(-1, 3, 3)) < 1e-8       # repaired covariances
    assert relerr(ref[6], v[:, 12]) < 1e-8                            # lnprob
    assert abs(ref[7] - star_out[0, 0]) < 1e-8 * abs(ref[7])          # log evidence
    for a, b in ((ref[9], v[:, 13]), (ref[10], v[:, 14]), (ref[11], v[:, 15]),
                 (ref[12], v[:, 16])):
        assert relerr(a, b) < 1e-8


def test_device_lnpost_with_los_dust_prior_vs_oracle():
    """`fit(dustfile=LOSTable(...))`: the built-in line-of-sight dust prior (reference
    pdf.py:752-840, host version pinned by tests/golden/dust.npz) evaluated on the device at
    the MLE point and for every Monte Carlo sample; indices bit-exact against the oracle
    with the same hook and stream, Philox and numpy streams, one sightline without
    coverage."""
    from brutus_amd import fitting
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.pdf import LOSTable, dust_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup()
    n = len(st["flux"])
    rng = np.random.RandomState(4)
    dist = np.concatenate([[0.05], 10. ** np.linspace(-1., 1.5, 30)])
    mean = np.cumsum(rng.uniform(0., 0.15, size=(n, dist.size)), axis=1)
    err = 0.05 + 0.1 * rng.uniform(size=(n, dist.size))
    mean[2, 7] = np.nan                          # no coverage on that sightline: flat prior
    tab = LOSTable(st["coords"][:, 0], st["coords"][:, 1], dist, mean, err)
    hook = lambda d, c, a, dustfile=None: dust_lnprior(d, c, a, dustfile=tab)
    calls = {"n": 0}
    orig = fitting._Engine.post_batch_device

    def spy(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    fitting._Engine.post_batch_device = spy
    orig_begin = fitting._Engine.post_numpy_begin

    def spy_begin(self, *a, **k):       # the two-phase form of the numpy-stream call
        calls["n"] += 1
        return orig_begin(self, *a, **k)
    fitting._Engine.post_numpy_begin = spy_begin
    try:
        for mk in (lambda: PhiloxRandomState(11), lambda: np.random.RandomState(11)):
            BF.batch_size = 5
            rs, ro = mk(), mk()
            before = calls["n"]
            dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                               parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                               lngalprior=gal_lnprior, dustfile=tab, data_coords=st["coords"],
                               Ndraws=60, rstate=rs))
            assert calls["n"] - before == 2          # the device stage ran (2 batches)
            for i in range(n):
                ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                                 labels, st["coords"][i], st["parallax"][i],
                                 st["parallax_err"][i], ro, gal_lnprior, lndustprior=hook,
                                 Nmc_prior=20, Ndraws=60)
                _compare(dev[i], ref, ("dust", i))
    finally:
        fitting._Engine.post_batch_device = orig
        fitting._Engine.post_numpy_begin = orig_begin


def test_device_lnpost_more_than_64_samples_per_model_both_streams():
    """Nmc_prior = 70 > 64: the numpy stream leaves the 8 records x 8 sample groups kernel
    (`k_post_mc_arr`, nmc <= 64) for the lane-per-record one reading the normals from
    memory, and the counter-based stream walks 105 Philox calls per record; odd and even
    run alignments (70 * 3 = 210 normals per record, batches of 4)."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _setup(nstar=6, seed=61)
    BF.batch_size = 4
    for mk in (lambda: PhiloxRandomState(70), lambda: np.random.RandomState(70)):
        rs, ro = mk(), mk()
        dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                           parallax_err=st["parallax_err"], Nmc_prior=70, lnprior=lnprior,
                           lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=40,
                           rstate=rs))
        for i in range(len(dev)):
            ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                             labels, st["coords"][i], st["parallax"][i],
                             st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=70, Ndraws=40)
            _compare(dev[i], ref, ("nmc70", i))
        if isinstance(rs, PhiloxRandomState):
            assert (rs.n_normal, rs.n_uniform) == (ro.n_normal, ro.n_uniform)
        else:
            assert rs.get_state()[2] == ro.get_state()[2]
            assert np.array_equal(rs.get_state()[1], ro.get_state()[1])
