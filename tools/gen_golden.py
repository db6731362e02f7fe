#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the UPSTREAM
reference (imported from /root/reference under tools/ref_shim.py).

Run in the build container only (the reference does not travel to the GPU
box):   PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Every file holds inputs and the reference's outputs for those inputs -- data,
no reference source.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so these are the pins for the CPU restatement in
oracle/ and, through it, for the HIP path.

Grid coefficients are float32 values promoted to float64 before they are
handed to the reference: that reproduces numba's arithmetic (f64 on f32-rounded
values), which the pure-Python shim would otherwise not (NEP 50).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import ref_shim  # noqa: E402
from brutus_amd import synth  # noqa: E402

F, U, P, C = ref_shim.import_reference()
OUT = os.path.join(ROOT, "tests", "golden")


def galprior(dists, coord, labels=None):
    """Analytic stand-in for the Galactic prior hook (the reference default
    needs astropy).  Same definition lives in tests/helpers.py."""
    with np.errstate(all="ignore"):
        lp = 2. * np.log(dists) - dists / 2. + 0.01 * np.cos(np.deg2rad(coord[1]))
    if labels is not None:
        lp = lp + 0.1 * labels['feh']
    return lp


def star_from_model(models, idx, av, rv, dist, frac_err, rng, noise=True):
    c = models[idx].astype(np.float64)
    sed = c[:, 0] + av * (c[:, 1] + rv * c[:, 2])
    f = 10. ** (-0.4 * sed) / dist ** 2
    e = frac_err * f
    if noise:
        f = f + rng.normal(size=f.shape) * e
    return f, e


def loglike_cases():
    rng = np.random.RandomState(42)
    cases = []

    def add(name, nmodel, nfilt, gseed, av, rv, dist, ferr, par=None,
            perr=None, mask_band=None, neg_band=None, **kw):
        models, _, _ = synth.make_grid(nmodel, nfilt, seed=gseed)
        idx = rng.randint(nmodel)
        f, e = star_from_model(models, idx, av, rv, dist, ferr, rng)
        m = np.ones(nfilt, dtype=bool)
        if mask_band is not None:
            m[mask_band] = False
            f[mask_band] = np.nan  # masked bands may hold garbage
        if neg_band is not None:
            f[neg_band] = -0.3 * abs(f[neg_band])
        cases.append(dict(name=name, models=models, flux=f, err=e, mask=m,
                          parallax=par, parallax_err=perr, kw=kw))

    add("hisnr_par_12", 2048, 12, 11, 1.2, 3.3, 1.0, 0.02, par=1.02, perr=0.05)
    add("losnr_nopar_12", 2048, 12, 12, 0.7, 3.1, 2.0, 0.15,
        par=np.nan, perr=np.nan)
    add("masked_band_8", 512, 8, 13, 0.4, 3.4, 0.5, 0.03, par=2.1, perr=0.3,
        mask_band=2)
    add("neg_flux_8", 512, 8, 14, 2.0, 3.0, 3.0, 0.2, par=np.nan, perr=np.nan,
        neg_band=0)
    add("av_zero_clamp_6", 512, 6, 15, 0.0, 3.32, 1.5, 0.04, par=0.6, perr=0.2)
    add("av_only_6", 512, 6, 16, 1.0, 3.32, 0.8, 0.03, par=np.nan,
        perr=np.nan, rvlim=(3.32, 3.32))
    add("no_dim_prior_8", 2048, 8, 17, 1.7, 3.5, 0.3, 0.05, par=3.5, perr=0.4,
        dim_prior=False)
    add("rv_hi_clamp_12", 512, 12, 18, 2.2, 7.9, 1.0, 0.02, par=np.nan,
        perr=np.nan, rv_gauss=(3.32, 5.0))
    add("rv_lo_clamp_12", 512, 12, 19, 2.2, 1.0, 1.0, 0.02, par=1.0, perr=0.1,
        rv_gauss=(3.32, 5.0))
    add("av_max_clamp_8", 512, 8, 20, 2.0, 3.3, 1.0, 0.03, par=np.nan,
        perr=np.nan, avlim=(0., 1.))
    add("av_prior_6", 512, 6, 21, 0.6, 3.3, 1.0, 0.05, par=1.0, perr=0.3,
        av_gauss=(0.5, 0.2))
    add("par_none_8", 512, 8, 22, 0.9, 3.3, 1.0, 0.05, par=None, perr=None)
    add("tight_tol_8", 512, 8, 23, 1.4, 3.6, 1.2, 0.04, par=0.8, perr=0.1,
        ltol=3e-3, ltol_subthresh=1e-2, init_thresh=5e-3)
    return cases


def run_loglike_case(c):
    calls = {"flux": 0, "nsel": None}
    orig = F._optimize_fit_flux

    def wrapped(*a, **k):
        calls["flux"] += 1
        if calls["nsel"] is None:
            calls["nsel"] = a[2].shape[0]
        return orig(*a, **k)

    F._optimize_fit_flux = wrapped
    try:
        out = F.loglike(c["flux"].copy(), c["err"].copy(), c["mask"].copy(),
                        c["models"].astype(np.float64),
                        parallax=c["parallax"], parallax_err=c["parallax_err"],
                        return_vals=True, **c["kw"])
    finally:
        F._optimize_fit_flux = orig
    lnl, Ndim, chi2, scale, av, rv, icov = out
    return dict(lnl=lnl, Ndim=int(Ndim), chi2=chi2, scale=scale, av=av, rv=rv,
                icov=icov, K2=calls["flux"], nsel=calls["nsel"])


def gen_loglike():
    for c in loglike_cases():
        r = run_loglike_case(c)
        kw = c["kw"]
        np.savez_compressed(
            os.path.join(OUT, "loglike_%s.npz" % c["name"]),
            models=c["models"], flux=c["flux"], err=c["err"], mask=c["mask"],
            parallax=np.array(np.nan if c["parallax"] is None else c["parallax"]),
            parallax_err=np.array(np.nan if c["parallax_err"] is None
                                  else c["parallax_err"]),
            parallax_is_none=np.array(c["parallax"] is None),
            kw_keys=np.array(sorted(kw.keys())),
            kw_vals=np.array([np.atleast_1d(np.asarray(kw[k], dtype=float))
                              .tolist() + [np.nan] * (2 - np.size(kw[k]))
                              for k in sorted(kw.keys())]).reshape(-1, 2),
            **r)
        print("loglike", c["name"], "Ndim", r["Ndim"], "K2", r["K2"], "nsel",
              r["nsel"])


def gen_fit():
    """Per-star `_fit` yields (reference fitting.py:1980-2065) with one
    `RandomState(1000 + i)` per star, so results do not depend on star order."""
    models, labels, lmask = synth.make_grid(4000, 8, seed=31)
    st = synth.make_stars(models, 12, seed=7)
    # a few hand-made edge cases
    st['mask'][1, 3] = False
    st['flux'][2, 5] = -abs(st['flux'][2, 5])
    st['parallax'][3] = np.nan
    st['parallax_err'][3] = np.nan
    st['mask'][4, [0, 1, 2]] = False   # 5 of 8 bands
    BF = F.BruteForce(models.astype(np.float64), labels, lmask)
    sp = BF._setup(st['flux'].copy(), st['err'].copy(), st['mask'].copy(),
                   None, data_coords=st['coords'], lngalprior=galprior,
                   parallax=st['parallax'], parallax_err=st['parallax_err'])
    lnprior = sp[5]
    mask_after_setup = np.array(sp[2])
    res = {}
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(st['flux'])):
        sl = slice(i, i + 1)
        gen = BF._fit(st['flux'][sl].copy(), st['err'][sl].copy(),
                      st['mask'][sl].copy(), parallax=st['parallax'][sl],
                      parallax_err=st['parallax_err'][sl], Nmc_prior=50,
                      lnprior=lnprior.copy(), lngalprior=galprior,
                      data_coords=st['coords'][sl],
                      rstate=np.random.RandomState(1000 + i), Ndraws=250)
        r = next(gen)
        for n, v in zip(names, r):
            res.setdefault(n, []).append(np.asarray(v))
        print("fit star", i, "Ndim", r[5], "levid", r[7], "chi2min", r[8])
    np.savez_compressed(
        os.path.join(OUT, "fit_synth.npz"), grid_nmodel=4000, grid_nfilt=8,
        grid_seed=31, flux=st['flux'], err=st['err'], mask=st['mask'],
        mask_after_setup=mask_after_setup, parallax=st['parallax'],
        parallax_err=st['parallax_err'], coords=st['coords'], lnprior=lnprior,
        seed0=1000, **{k: np.array(v) for k, v in res.items()})


def gen_loglike_init():
    """`loglike` with per-model `av_init` / `rv_init` arrays (reference
    fitting.py:697-707): a different starting point changes the number of sweeps and,
    through the sweep count, the converged values of every model."""
    rng = np.random.RandomState(77)
    models, _, _ = synth.make_grid(1024, 8, seed=24)
    f, e = star_from_model(models, rng.randint(1024), 1.1, 3.4, 1.3, 0.04, rng)
    m = np.ones(8, dtype=bool)
    av_init = rng.uniform(0., 2.5, 1024)
    rv_init = rng.uniform(2.2, 4.6, 1024)
    out = {}
    for tag, kw in (("both", dict(av_init=av_init.copy(), rv_init=rv_init.copy())),
                    ("av", dict(av_init=av_init.copy()))):
        r = F.loglike(f.copy(), e.copy(), m.copy(), models.astype(np.float64), parallax=0.8,
                      parallax_err=0.1, return_vals=True, **kw)
        for n, v in zip("lnl Ndim chi2 scale av rv icov".split(), r):
            out["%s_%s" % (tag, n)] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "init_loglike.npz"), models=models, flux=f, err=e,
                        mask=m, parallax=0.8, parallax_err=0.1, av_init=av_init,
                        rv_init=rv_init, **out)
    print("loglike with av_init / rv_init done")


def gen_fit_cdf():
    """`_fit` with `wt_thresh=None`: CDF thresholding (reference fitting.py:992-998,
    1017-1022) -- ascending sort, so the most probable models are dropped and the rest is
    handed on in sort order.  Two objects with the default `mem_lim`, two with
    one small enough for the `Nsel_max` clip (fitting.py:1029-1036) to act."""
    models, labels, lmask = synth.make_grid(1500, 6, seed=33)
    st = synth.make_stars(models, 4, seed=9)
    st['parallax'][1] = np.nan
    st['parallax_err'][1] = np.nan
    BF = F.BruteForce(models.astype(np.float64), labels, lmask)
    sp = BF._setup(st['flux'].copy(), st['err'].copy(), st['mask'].copy(), None,
                   data_coords=st['coords'], lngalprior=galprior, parallax=st['parallax'],
                   parallax_err=st['parallax_err'])
    lnprior = sp[5]
    res = {}
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    mem = [8000., 8000., 4., 4.]
    for i in range(4):
        sl = slice(i, i + 1)
        gen = BF._fit(st['flux'][sl].copy(), st['err'][sl].copy(), st['mask'][sl].copy(),
                      parallax=st['parallax'][sl], parallax_err=st['parallax_err'][sl],
                      Nmc_prior=12, lnprior=lnprior.copy(), lngalprior=galprior,
                      data_coords=st['coords'][sl], wt_thresh=None, cdf_thresh=2e-3,
                      rstate=np.random.RandomState(500 + i), Ndraws=40, mem_lim=mem[i])
        r = next(gen)
        for n, v in zip(names, r):
            res.setdefault(n, []).append(np.asarray(v))
        print("cdf fit star", i, "levid", r[7], "chi2min", r[8], "distinct models",
              len(set(r[0].tolist())))
    np.savez_compressed(
        os.path.join(OUT, "fit_cdf.npz"), grid_nmodel=1500, grid_nfilt=6, grid_seed=33,
        flux=st['flux'], err=st['err'], mask=st['mask'], parallax=st['parallax'],
        parallax_err=st['parallax_err'], coords=st['coords'], lnprior=lnprior, seed0=500,
        mem_lim=np.array(mem), **{k: np.array(v) for k, v in res.items()})


def gen_helpers():
    rng = np.random.RandomState(5)
    A = rng.normal(size=(64, 3, 3))
    A = np.einsum('nij,nkj->nik', A, A) + 0.1 * np.eye(3)
    x = np.concatenate([[-1., 0.], rng.uniform(0.01, 400., 62)])
    mean = rng.normal(size=(7, 3))
    cov = A[:7]
    mvn = U.sample_multivariate_normal(mean, cov, size=11,
                                       rstate=np.random.RandomState(9))
    mgrid = np.array([0.05, 0.08, 0.0800001, 0.3, 0.5, 0.5000001, 1.0, 3.0])
    scales = rng.uniform(0.1, 4., 50)
    serrs = rng.uniform(0.01, 0.5, 50)
    flux = rng.uniform(-1e-9, 1e-8, size=(5, 6))
    ferr = rng.uniform(1e-11, 1e-9, size=(5, 6))
    with np.errstate(all="ignore"):
        mag, magerr = U.magnitude(flux, ferr)
    np.savez_compressed(
        os.path.join(OUT, "helpers.npz"),
        inv3_in=A, inv3_out=U._inverse3(A),
        chi2_x=x, chi2_df5=U._chisquare_logpdf(x.copy(), 5),
        chi2_df9=U._chisquare_logpdf(x.copy(), 9),
        mvn_mean=mean, mvn_cov=cov, mvn_out=mvn,
        imf_m=mgrid, imf_out=P.imf_lnprior(mgrid),
        sp_scales=scales, sp_serrs=serrs,
        sp_hi=P.scale_parallax_lnprior(scales, serrs, 1.0, 0.1),
        sp_lo=P.scale_parallax_lnprior(scales, serrs, 1.0, 0.3),
        sp_nan=P.scale_parallax_lnprior(scales, serrs, np.nan, 0.3),
        pl_out=P.parallax_lnprior(np.sqrt(scales), 1.1, 0.2),
        pl_nan=P.parallax_lnprior(np.sqrt(scales), np.nan, 0.2),
        p2s_hi=np.array(P.parallax_to_scale(1.0, 0.1)),
        p2s_lo=np.array(P.parallax_to_scale(1.0, 0.3)),
        mag_flux=flux, mag_ferr=ferr, mag_out=mag, magerr_out=magerr)
    print("helpers done")


def gen_setup():
    """`BruteForce._setup` (reference fitting.py:1144-1424): band masking by
    mag/magerr limits, photometric offsets, static prior, error for <4 bands."""
    models, labels, lmask = synth.make_grid(2048, 6, seed=41)
    st = synth.make_stars(models, 8, seed=11)
    flux, err, mask = st['flux'].copy(), st['err'].copy(), st['mask'].copy()
    flux[0, 1] = 10. ** (-0.4 * 51.)        # mag > mag_max
    err[1, 2] = 0.5 * flux[1, 2]            # magerr > merr_max
    flux[2, 3] = np.nan                     # non-finite flux
    err[3, 4] = 0.                          # non-positive error
    offs = np.array([1.0, 1.02, 0.97, 1.0, 1.05, 0.99])
    BF = F.BruteForce(models.astype(np.float64), labels, lmask)
    out = BF._setup(flux.copy(), err.copy(), mask.copy(), None,
                    phot_offsets=offs, data_coords=st['coords'],
                    lngalprior=galprior, parallax=st['parallax'],
                    parallax_err=st['parallax_err'])
    bad_mask = mask.copy()
    bad_mask[5, :3] = False                 # 3 valid bands -> ValueError
    try:
        BF._setup(flux.copy(), err.copy(), bad_mask, None,
                  data_coords=st['coords'], lngalprior=galprior)
        raised = False
    except ValueError:
        raised = True
    np.savez_compressed(
        os.path.join(OUT, "setup.npz"), grid_nmodel=2048, grid_nfilt=6,
        grid_seed=41, flux=flux, err=err, mask=mask, offsets=offs,
        out_flux=out[0], out_err=out[1], out_mask=out[2], lnprior=out[5],
        av_gauss=np.array(out[8], dtype=float), wt_thresh=out[9],
        raised_3band=raised)
    print("setup done; 3-band ValueError raised:", raised)


def gen_galprior_pieces():
    """Astropy-free pieces of `pdf.gal_lnprior` (reference pdf.py:263-473)."""
    rng = np.random.RandomState(3)
    R = rng.uniform(0., 20., 200)
    Z = rng.uniform(-5., 5., 200)
    feh = rng.uniform(-3., 0.6, 200)
    age = rng.uniform(-0.5, 14.5, 200)
    np.savez_compressed(
        os.path.join(OUT, "galprior_pieces.npz"), R=R, Z=Z, feh=feh, age=age,
        disk_thin=P.logn_disk(R, Z), disk_thick=P.logn_disk(R, Z, R_scale=2.0, Z_scale=0.9),
        halo=P.logn_halo(R, Z),
        feh_thin=P.logp_feh(feh), feh_halo=P.logp_feh(feh, feh_mean=-1.6, feh_sigma=0.5),
        age_thin=P.logp_age_from_feh(age.copy(), feh_mean=-0.2),
        age_thick=P.logp_age_from_feh(age.copy(), feh_mean=-0.7),
        age_halo=P.logp_age_from_feh(age.copy(), feh_mean=-1.6))
    print("galprior pieces done")


def gen_cluster():
    """`cluster.isochrone_loglike` (reference cluster.py:23-419) with the fake
    isochrone of tests/helpers.py."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_cluster_data
    res = {}
    for tag, nobj, nb, seed in (("a", 200, 6, 1), ("b", 300, 8, 2)):
        iso, phot, err, par, perr = make_cluster_data(nobj, nb, seed)
        theta = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])
        for dp in (True, False):
            tot, mix = C.isochrone_loglike(theta, iso, phot.copy(), err.copy(),
                                           parallax=par.copy(), parallax_err=perr.copy(),
                                           dim_prior=dp, return_lnls=True)
            res["%s_dp%d_tot" % (tag, dp)] = tot
            res["%s_dp%d_mix" % (tag, dp)] = mix
        # free offsets + correction parameters, no parallax
        theta2 = np.concatenate([theta, np.linspace(0.97, 1.03, nb - 1), [0.5]])
        tot, mix = C.isochrone_loglike(theta2, iso, phot.copy(), err.copy(),
                                       offsets=[1.0] + [None] * (nb - 1),
                                       corr_params=[None, 0., 0., 1.],
                                       return_lnls=True)
        res["%s_free_tot" % tag] = tot
        res["%s_free_mix" % tag] = mix
        print("cluster", tag, res["%s_dp1_tot" % tag], res["%s_dp0_tot" % tag], tot)
    np.savez_compressed(os.path.join(OUT, "cluster.npz"), **res)


def gen_orion():
    """`_fit` yields for 20 objects of the reference's real-data demo
    catalogue (demos/Orion_l204.7_b-19.2.h5: PS grizy + 2MASS JHKs magnitudes,
    missing bands flagged mag = -999 / err = inf, Gaia parallaxes in arcsec),
    converted like the notebook does (Overview 3, cell "convert to flux"),
    against a 10k-model synthetic grid.  The 20 catalogue rows are stored in
    the fixture (input data), the reference outputs beside them."""
    from brutus_amd import h5io
    cat = h5io.read_dataset(os.path.join(ref_shim.REFERENCE_ROOT, "demos",
                                         "Orion_l204.7_b-19.2.h5"),
                            "/photometry/pixel 0-0")
    nb = np.sum(np.isfinite(cat["err"]) & (cat["mag"] > -900), axis=1)
    pick = np.concatenate([np.where(nb == 8)[0][:8], np.where(nb == 7)[0][:4],
                           np.where((nb >= 4) & (nb <= 6))[0][:8]])[:20]
    cat = cat[pick]
    mag, magerr = cat["mag"].astype(np.float64), cat["err"].astype(np.float64)
    mask = np.isfinite(magerr) & (mag > -900)
    with np.errstate(all="ignore"):
        flux = np.where(mask, 10. ** (-0.4 * mag), np.nan)
        err = np.where(mask, flux * magerr * 0.4 * np.log(10.), np.nan)
    par = cat["parallax"].astype(np.float64) * 1e3
    perr = cat["parallax_error"].astype(np.float64) * 1e3
    bad = ~(np.isfinite(par) & np.isfinite(perr) & (perr > 0) & (perr < 1e3)) | (par == 0)
    par[bad], perr[bad] = np.nan, np.nan
    coords = np.c_[cat["l"], cat["b"]]
    models, labels, lmask = synth.make_mist_like_grid(10000, 8, seed=77)
    # put the grid at the catalogue's brightness: shift to apparent mags ~ 14-20 at 0.4 kpc
    BF = F.BruteForce(models.astype(np.float64), labels, lmask)
    sp = BF._setup(flux.copy(), err.copy(), mask.copy(), None, data_coords=coords,
                   lngalprior=galprior, parallax=par, parallax_err=perr,
                   merr_max=1.0)
    lnprior, mask2 = sp[5], np.array(sp[2])
    res = {}
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(flux)):
        sl = slice(i, i + 1)
        r = next(BF._fit(flux[sl].copy(), err[sl].copy(), mask2[sl].copy(),
                         parallax=par[sl], parallax_err=perr[sl], Nmc_prior=30,
                         lnprior=lnprior.copy(), lngalprior=galprior,
                         data_coords=coords[sl],
                         rstate=np.random.RandomState(2000 + i), Ndraws=100))
        for n, v in zip(names, r):
            res.setdefault(n, []).append(np.asarray(v))
        print("orion star", i, "bands", int(mask2[i].sum()), "par", par[i], "levid", r[7])
    np.savez_compressed(os.path.join(OUT, "fit_orion20.npz"), grid_nmodel=10000,
                        grid_nfilt=8, grid_seed=77, flux=flux, err=err, mask=mask2,
                        parallax=par, parallax_err=perr, coords=coords,
                        lnprior=lnprior, seed0=2000,
                        **{k: np.array(v) for k, v in res.items()})


def gen_fit_philox():
    """The REFERENCE's `_fit` driven by `brutus_amd.rng.PhiloxRandomState` (a
    valid `rstate` object: it has the `normal` / `choice` methods the reference
    calls) and by `brutus_amd.galprior.gal_lnprior` as the `lngalprior` hook:
    the pin for the device-side lnpost (brutus_post_batch).  One shared
    sequential stream over all objects."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    models, labels, lmask = synth.make_mist_like_grid(3000, 8, seed=61)
    st = synth.make_stars(models, 8, seed=62)
    st['mask'][2, 1] = False
    st['parallax'][5] = np.nan
    st['parallax_err'][5] = np.nan
    BF = F.BruteForce(models.astype(np.float64), labels, lmask)
    sp = BF._setup(st['flux'].copy(), st['err'].copy(), st['mask'].copy(), None,
                   data_coords=st['coords'], lngalprior=gal_lnprior,
                   parallax=st['parallax'], parallax_err=st['parallax_err'])
    lnprior = sp[5]
    rs = PhiloxRandomState(31337)
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    res = {}
    gen = BF._fit(st['flux'].copy(), st['err'].copy(), st['mask'].copy(),
                  parallax=st['parallax'], parallax_err=st['parallax_err'],
                  Nmc_prior=25, lnprior=lnprior.copy(), lngalprior=gal_lnprior,
                  data_coords=st['coords'], rstate=rs, Ndraws=80)
    for i, r in enumerate(gen):
        for n, v in zip(names, r):
            res.setdefault(n, []).append(np.asarray(v))
        print("philox fit star", i, "levid", r[7])
    np.savez_compressed(os.path.join(OUT, "fit_philox.npz"), grid_nmodel=3000,
                        grid_nfilt=8, grid_seed=61, flux=st['flux'], err=st['err'],
                        mask=st['mask'], parallax=st['parallax'],
                        parallax_err=st['parallax_err'], coords=st['coords'],
                        lnprior=lnprior, seed=31337, n_normal=rs.n_normal,
                        n_uniform=rs.n_uniform,
                        **{k: np.array(v) for k, v in res.items()})


def gen_ps1():
    """`ps1_MrLF_lnprior` (reference pdf.py:111-141) on a grid that leaves the table on
    both sides (linear extrapolation), and `_setup`'s static prior for a Bayestar-style
    grid whose labels carry `Mr` but no `mini` (reference fitting.py:1334-1346)."""
    Mr = np.concatenate([np.linspace(-6., 24., 121), [-2., 16., 4.9999, 5.0001]])
    out = P.ps1_MrLF_lnprior(Mr)
    models, labels, lmask = synth.make_grid(1024, 6, seed=43)
    ltype = np.dtype([('Mr', 'f8'), ('feh', 'f8'), ('agewt', 'f8')])
    lab = np.zeros(len(labels), dtype=ltype)
    lab['Mr'] = np.round(np.random.RandomState(3).uniform(-3., 17., len(labels)) / 0.25) * 0.25
    lab['feh'] = labels['feh']
    lab['agewt'] = labels['agewt']
    mtype = np.dtype([(n, '?') for n in ltype.names])
    lm = np.zeros(1, dtype=mtype)
    lm['Mr'] = True
    lm['feh'] = True
    st = synth.make_stars(models, 4, seed=12)
    BF = F.BruteForce(models.astype(np.float64), lab, lm)
    res = BF._setup(st['flux'].copy(), st['err'].copy(), st['mask'].copy(), None,
                    data_coords=st['coords'], lngalprior=galprior)
    np.savez_compressed(os.path.join(OUT, "ps1.npz"), Mr=Mr, lnp=out, grid_seed=43,
                        lab_Mr=lab['Mr'], lab_feh=lab['feh'], lab_agewt=lab['agewt'],
                        setup_lnprior=res[5])
    print("ps1 done")


class _FakeLOS(object):
    """Stand-in for dust.Bayestar (dust.py:184-299): `query(coord)` returns
    `(av_dist, av_mean, av_err)` of a sightline."""

    def __init__(self, dist, mean, err):
        self.args = (dist, mean, err)

    def query(self, coord):
        return self.args


def gen_dust():
    """`dust_lnprior` (reference pdf.py:752-840) with the Bayestar object replaced by a
    table (the module-level `bayestar` global is what the reference queries), for the
    two call shapes of `lnpost` ((Nsel,) and (Nmc, Nsel)) and a sightline without
    coverage."""
    rng = np.random.RandomState(17)
    dist = np.concatenate([[0.063], 10. ** np.linspace(-1., 1.8, 40)])
    mean = np.cumsum(rng.uniform(0., 0.12, dist.size))
    err = 0.05 + 0.1 * rng.uniform(size=dist.size)
    d1 = 10. ** rng.uniform(-1.5, 2., 200)
    a1 = rng.uniform(0., 4., 200)
    d2 = 10. ** rng.uniform(-1.5, 2., (7, 50))
    a2 = rng.uniform(0., 4., (7, 50))
    P.bayestar = _FakeLOS(dist, mean, err)
    o1 = P.dust_lnprior(d1, (120., 10.), a1)
    o2 = P.dust_lnprior(d2, (120., 10.), a2)
    o3 = P.dust_lnprior(d1, (120., 10.), a1, offset=0.1, scale=0.9, smooth=1.5, scatter=0.1)
    bad = mean.copy()
    bad[5] = np.nan
    P.bayestar = _FakeLOS(dist, bad, err)
    o4 = P.dust_lnprior(d1, (120., 10.), a1)
    np.savez_compressed(os.path.join(OUT, "dust.npz"), dist=dist, mean=mean, err=err,
                        d1=d1, a1=a1, d2=d2, a2=a2, o1=o1, o2=o2, o3=o3, o4=o4)
    print("dust done")


def gen_bin_pdfs():
    """`pdf.bin_pdfs_distred` (pdf.py:843-1113) on saved draws and on regenerated ones, every
    distance type, E(B-V), CDF, both forms of `smooth` / `bins` / `span`."""
    rng = np.random.RandomState(77)
    nobj, ns = 5, 40
    dists = 10. ** rng.normal(0.2, 0.15, size=(nobj, ns))
    reds = np.abs(rng.normal(1.2, 0.5, size=(nobj, ns)))
    dreds = rng.normal(3.3, 0.2, size=(nobj, ns))
    scales = 1. / dists ** 2
    covs = np.zeros((nobj, ns, 3, 3))
    for i in range(nobj):
        for k in range(ns):
            A = rng.normal(size=(3, 3)) * np.array([0.05 * scales[i, k], 0.1, 0.05])[:, None]
            covs[i, k] = A @ A.T + np.diag([1e-6 * scales[i, k] ** 2, 1e-4, 1e-4])
    par = 1. / np.median(dists, axis=1) + rng.normal(size=nobj) * 0.05
    perr = np.full(nobj, 0.05)
    par[1] = np.nan
    perr[3] = np.nan
    coord = np.stack([rng.uniform(0, 360, nobj), rng.uniform(-60, 60, nobj)], axis=1)
    prior = lambda d, c: galprior(d, c)
    cases = [
        ("dm", dict()),
        ("par_cdf", dict(dist_type="parallax", cdf=True, bins=24)),
        ("scale_ebv", dict(dist_type="scale", ebv=True, bins=(30, 12), smooth=(2., 0.05))),
        ("dist_span", dict(dist_type="distance", span=((0., 4.), (0.3, 6.)), bins=(28, 16), smooth=1.5)),
    ]
    res = dict(dists=dists, reds=reds, dreds=dreds, scales=scales, covs=covs, parallaxes=par,
               parallax_errors=perr, coord=coord, names=np.array([c[0] for c in cases]))
    for name, kw in cases:
        kw = dict(kw)
        kw.setdefault("bins", (36, 18))
        b, xe, ye = P.bin_pdfs_distred((dists.copy(), reds.copy(), dreds.copy()), parallaxes=par,
                                       parallax_errors=perr, **kw)
        res["saved_%s" % name], res["saved_%s_x" % name], res["saved_%s_y" % name] = b, xe, ye
        b, xe, ye = P.bin_pdfs_distred((scales.copy(), reds.copy(), dreds.copy(), covs.copy()),
                                       lndistprior=prior, coord=coord, parallaxes=par,
                                       parallax_errors=perr, Nr=12, rstate=np.random.RandomState(5),
                                       **kw)
        res["regen_%s" % name], res["regen_%s_x" % name], res["regen_%s_y" % name] = b, xe, ye
    np.savez_compressed(os.path.join(OUT, "bin_pdfs.npz"), **res)
    print("wrote bin_pdfs.npz")


def gen_utils_misc():
    """The small helpers of `brutus.utils.__all__` (utils.py:43-127, 179-347, 718-762, 978-1086)."""
    rng = np.random.RandomState(12)
    A = rng.normal(size=(7, 3, 3))
    B = rng.normal(size=(7, 3, 3))
    spd = A @ np.transpose(A, (0, 2, 1)) + 0.1 * np.eye(3)
    notpd = spd.copy()
    notpd[:, 0, 0] = -1.
    x = np.linspace(-4., 6., 41)
    coeffs = rng.normal(size=(9, 5, 3)).astype(np.float32).astype(np.float64)
    av, rv = rng.uniform(0, 2, 9), rng.uniform(2, 5, 9)
    samp, wts = rng.normal(size=200), rng.uniform(size=200)
    q = np.array([0.025, 0.16, 0.5, 0.84, 0.975])
    phot, err = 10. ** rng.normal(-8, 1, size=(6, 4)), 10. ** rng.normal(-9.5, 0.3, size=(6, 4))
    phot[0, 0] = -1e-9
    lm, le = U.luptitude(phot, err, skynoise=2e-9, zeropoints=3.)
    res = dict(
        A=A, B=B, adj=U._adjoint3(A), invT=U._inverse_transpose3(A), dot=U._dot3(A, B),
        spd=spd, notpd=notpd,
        psd=np.array([U._isPSD(m) for m in spd] + [U._isPSD(m) for m in notpd]),
        x=x, tn_pdf=U._truncnorm_pdf(x.copy(), -1.5, 2.5, loc=1., scale=1.7),
        tn_logpdf=U._truncnorm_logpdf(x.copy(), -1.5, 2.5, loc=1., scale=1.7),
        tn_scalar=np.array([U._truncnorm_pdf(0.3, -1.5, 2.5, 1., 1.7), U._truncnorm_pdf(9., -1.5, 2.5, 1., 1.7),
                            U._truncnorm_logpdf(0.3, -1.5, 2.5, 1., 1.7), U._truncnorm_logpdf(9., -1.5, 2.5, 1., 1.7)]),
        coeffs=coeffs, av=av, rv=rv, samp=samp, wts=wts, q=q,
        quant=np.asarray(U.quantile(samp, q)), quant_w=np.asarray(U.quantile(samp, q, weights=wts)),
        phot=phot, err=err, lup=lm, lup_err=le,
        add=U.add_mag(np.linspace(10, 20, 7), np.linspace(21, 9, 7), f1=0.7, f2=1.3))
    for k, rf in (("mag", False), ("flux", True)):
        sd, rvc, drv = U._get_seds(coeffs, av, rv, return_flux=rf)
        res["seds_" + k], res["rvecs_" + k], res["drvecs_" + k] = sd, rvc, drv
    ip, ie = U.inv_luptitude(lm, le, skynoise=2e-9, zeropoints=3.)
    res["ilup"], res["ilup_err"] = ip, ie
    np.savez_compressed(os.path.join(OUT, "utils_misc.npz"), **res)
    print("wrote utils_misc.npz")


def gen_psd():
    """`lnpost` on a crafted set of precision matrices whose inverse is NOT positive
    definite, so that the repair loop of reference fitting.py:1039-1065 runs on purpose:
    a non-positive variance in each position, pairs of them, all three, and matrices
    with positive diagonal that are indefinite; several need more than one doubling of
    the regulariser.  Well-conditioned matrices in between must come out untouched."""
    rng = np.random.RandomState(77)
    n = 160
    A = rng.normal(size=(n, 3, 3))
    icov = np.einsum('nij,nkj->nik', A, A) + 0.3 * np.eye(3)        # SPD
    scale = 10. ** rng.uniform(-1.5, 0.5, n)
    # per-model units: (s, Av, Rv) precisions of very different magnitude
    u = np.stack([1. / scale, np.full(n, 8.), np.full(n, 5.)], axis=1)
    icov = icov * u[:, :, None] * u[:, None, :]
    kinds = np.zeros(n, dtype=int)

    def flip(k, signs, boost=1.):
        """make the covariance (= inverse) have eigen-directions of negative variance"""
        w, V = np.linalg.eigh(icov[k])
        w = w * np.asarray(signs) * boost
        icov[k] = (V * w) @ V.T

    spec = [(1, (-1, 1, 1)), (2, (1, -1, 1)), (3, (1, 1, -1)), (4, (-1, -1, 1)),
            (5, (-1, 1, -1)), (6, (1, -1, -1)), (7, (-1, -1, -1))]
    for j in range(0, 105):
        kind, signs = spec[j % 7]
        flip(j, signs, boost=10. ** rng.uniform(-2., 2.))
        kinds[j] = kind
    # positive diagonal of the covariance but indefinite: strong off-diagonals
    for j in range(105, 125):
        d = np.sqrt(np.diag(icov[j]))
        c = np.diag(np.diag(icov[j]))
        c[0, 1] = c[1, 0] = 1.4 * d[0] * d[1] * rng.choice([-1, 1])
        c[1, 2] = c[2, 1] = 0.2 * d[1] * d[2]
        icov[j] = c
        kinds[j] = 8
    # diag entries of the precision exactly zero / negative
    icov[125, 0, 0] = 0.
    icov[126, 1, 1] = -icov[126, 1, 1]
    icov[127, 2, 2] = 0.
    kinds[125:128] = 9
    icov = 0.5 * (icov + np.transpose(icov, (0, 2, 1)))
    av = rng.uniform(0.2, 3., n)
    rv = rng.uniform(2., 5., n)
    lnl = rng.uniform(-3., 0., n)          # all inside both cuts
    chi2 = rng.uniform(3., 9., n)
    lnprior = rng.uniform(-0.5, 0.5, n)
    labels = np.zeros(n, dtype=[('feh', 'f8')])
    labels['feh'] = rng.uniform(-1., 0.3, n)
    res = (lnl.copy(), 8, chi2.copy(), scale.copy(), av.copy(), rv.copy(), icov.copy())
    out = F.lnpost(res, parallax=None, parallax_err=None, coord=(50., 20.), Nmc_prior=30,
                   lnprior=lnprior.copy(), lngalprior=galprior, lndustprior=None,
                   dlabels=labels, rstate=np.random.RandomState(123), apply_av_prior=False)
    sel, cov, lnp, dist_mc, a_mc, r_mc, lnp_mc = out
    with np.errstate(all="ignore"):
        bad0 = ~np.all(np.linalg.eigvals(U._inverse3(icov.copy())) > 0, axis=1)
    np.savez_compressed(os.path.join(OUT, "psd.npz"), icov=icov, scale=scale, av=av, rv=rv,
                        lnl=lnl, chi2=chi2, lnprior=lnprior, feh=labels['feh'], kinds=kinds,
                        not_psd_before=bad0, sel=sel, cov=cov, lnp=lnp, dist_mc=dist_mc,
                        a_mc=a_mc, r_mc=r_mc, lnp_mc=lnp_mc)
    print("psd done: %d of %d matrices needed repair, %d kept by lnpost"
          % (bad0.sum(), n, len(sel)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["loglike", "fit", "helpers", "setup", "galprior",
                             "cluster"]
    if "binpdfs" in which:
        gen_bin_pdfs()
    if "utilsmisc" in which:
        gen_utils_misc()
    if "init" in which:
        gen_loglike_init()
    if "cdf" in which:
        gen_fit_cdf()
    if "ps1" in which:
        gen_ps1()
    if "dust" in which:
        gen_dust()
    if "psd" in which:
        gen_psd()
    if "cluster" in which:
        gen_cluster()
    if "orion" in which:
        gen_orion()
    if "philox" in which:
        gen_fit_philox()
    if "galprior" in which:
        gen_galprior_pieces()
    if "helpers" in which:
        gen_helpers()
    if "setup" in which:
        gen_setup()
    if "loglike" in which:
        gen_loglike()
    if "fit" in which:
        gen_fit()

