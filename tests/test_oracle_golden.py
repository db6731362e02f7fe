"""CPU: the oracle restatement against the reference-generated golden vectors
(tests/golden/, made by tools/gen_golden.py from the upstream code)."""
import os

import numpy as np
import pytest

from helpers import (GOLDEN, galprior, load_loglike_case, loglike_golden_files,
                     relerr)
from oracle import brutus_oracle as O
from brutus_amd import synth

TOL = 1e-10  # observed ~1e-14 (libm vs numpy SIMD `pow` last-ulp differences)


@pytest.mark.parametrize("path", loglike_golden_files(),
                         ids=lambda p: os.path.basename(p)[8:-4])
def test_loglike_matches_reference(path):
    z, kw, par, perr = load_loglike_case(path)
    tr = {}
    out = O.loglike(z["flux"], z["err"], z["mask"], z["models"], parallax=par,
                    parallax_err=perr, return_vals=True, trace=tr, **kw)
    lnl, Ndim, chi2, scale, av, rv, icov = out
    assert Ndim == int(z["Ndim"])
    assert tr["K2"] == int(z["K2"])
    assert len(tr["init_sel"]) == int(z["nsel"])
    for name, got in (("lnl", lnl), ("chi2", chi2), ("scale", scale),
                      ("av", av), ("rv", rv), ("icov", icov)):
        assert relerr(z[name], got) < TOL, name


def test_fit_star_matches_reference():
    z = np.load(os.path.join(GOLDEN, "fit_synth.npz"))
    models, labels, lmask = synth.make_grid(int(z["grid_nmodel"]),
                                            int(z["grid_nfilt"]),
                                            seed=int(z["grid_seed"]))
    lnprior = O.static_lnprior(labels, lmask)
    assert np.array_equal(lnprior, z["lnprior"])
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(z["flux"])):
        rs = np.random.RandomState(int(z["seed0"]) + i)
        out = O.fit_star(z["flux"][i], z["err"][i], z["mask"][i], models,
                         lnprior, labels, z["coords"][i], z["parallax"][i],
                         z["parallax_err"][i], rs, galprior, Nmc_prior=50,
                         Ndraws=250)
        assert np.array_equal(out[0], z["sidxs"][i]), "star %d indices" % i
        for n, got in zip(names[1:], out[1:]):
            assert relerr(z[n][i], got) < 1e-9, (i, n)


def test_helpers_match_reference():
    z = np.load(os.path.join(GOLDEN, "helpers.npz"))
    assert relerr(z["inv3_out"], O.inverse3(z["inv3_in"])) < 1e-13
    assert relerr(z["chi2_df5"], O.chisquare_logpdf(z["chi2_x"], 5)) < 1e-14
    assert relerr(z["chi2_df9"], O.chisquare_logpdf(z["chi2_x"], 9)) < 1e-14
    mvn = O.sample_multivariate_normal(z["mvn_mean"], z["mvn_cov"], 11,
                                       np.random.RandomState(9))
    assert relerr(z["mvn_out"], mvn) < 1e-13
    assert relerr(z["imf_out"], O.imf_lnprior(z["imf_m"])) < 1e-14
    s, e = z["sp_scales"], z["sp_serrs"]
    assert relerr(z["sp_hi"], O.scale_parallax_lnprior(s, e, 1.0, 0.1)) < 1e-14
    assert relerr(z["sp_lo"], O.scale_parallax_lnprior(s, e, 1.0, 0.3)) < 1e-14
    assert relerr(z["sp_nan"], O.scale_parallax_lnprior(s, e, np.nan, 0.3)) == 0
    assert relerr(z["pl_out"], O.parallax_lnprior(np.sqrt(s), 1.1, 0.2)) < 1e-14
    assert relerr(z["pl_nan"], O.parallax_lnprior(np.sqrt(s), np.nan, 0.2)) == 0
    assert relerr(z["p2s_hi"], np.array(O.parallax_to_scale(1.0, 0.1))) < 1e-15
    assert relerr(z["p2s_lo"], np.array(O.parallax_to_scale(1.0, 0.3))) < 1e-15
    mag, magerr = O.magnitude(z["mag_flux"], z["mag_ferr"])
    assert relerr(z["mag_out"], mag) < 1e-15
    assert relerr(z["magerr_out"], magerr) < 1e-15


@pytest.mark.parametrize("path", loglike_golden_files(),
                         ids=lambda p: os.path.basename(p)[8:-4])
def test_c_oracle_matches_reference(path):
    """oracle/loglike_ref.c (the CPU-baseline port) against the same vectors."""
    import __graft_entry__
    from oracle import c_oracle
    __graft_entry__.build_oracle()
    z, kw, par, perr = load_loglike_case(path)
    tr = {}
    out = c_oracle.loglike(z["flux"], z["err"], z["mask"], z["models"],
                           parallax=par, parallax_err=perr, trace=tr, **kw)
    assert out[1] == int(z["Ndim"])
    assert tr["K2"] == int(z["K2"]) and tr["nsel"] == int(z["nsel"])
    for name, got in (("lnl", out[0]), ("chi2", out[2]), ("scale", out[3]),
                      ("av", out[4]), ("rv", out[5]), ("icov", out[6])):
        assert relerr(z[name], got) < 1e-11, name


def test_fit_orion_catalogue_matches_reference():
    """20 objects of the reference's real-data demo catalogue (missing bands,
    NaN parallaxes, poor fits) through the oracle's `_fit` restatement."""
    z = np.load(os.path.join(GOLDEN, "fit_orion20.npz"))
    models, labels, lmask = synth.make_mist_like_grid(int(z["grid_nmodel"]),
                                                      int(z["grid_nfilt"]),
                                                      seed=int(z["grid_seed"]))
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(z["flux"])):
        out = O.fit_star(z["flux"][i], z["err"][i], z["mask"][i], models,
                         z["lnprior"], labels, z["coords"][i], z["parallax"][i],
                         z["parallax_err"][i],
                         np.random.RandomState(int(z["seed0"]) + i), galprior,
                         Nmc_prior=30, Ndraws=100)
        assert np.array_equal(out[0], z["sidxs"][i]), i
        for n, got in zip(names[1:], out[1:]):
            assert relerr(z[n][i], got) < 1e-8, (i, n)


def test_fit_with_philox_rstate_matches_reference():
    """The counter-based rstate (brutus_amd.rng) through the oracle against the
    upstream run with the same object (tests/golden/fit_philox.npz)."""
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    z = np.load(os.path.join(GOLDEN, "fit_philox.npz"))
    models, labels, lmask = synth.make_mist_like_grid(int(z["grid_nmodel"]),
                                                      int(z["grid_nfilt"]),
                                                      seed=int(z["grid_seed"]))
    rs = PhiloxRandomState(int(z["seed"]))
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(z["flux"])):
        out = O.fit_star(z["flux"][i], z["err"][i], z["mask"][i], models,
                         z["lnprior"], labels, z["coords"][i], z["parallax"][i],
                         z["parallax_err"][i], rs, gal_lnprior, Nmc_prior=25,
                         Ndraws=80)
        assert np.array_equal(out[0], z["sidxs"][i]), i
        for n, got in zip(names[1:], out[1:]):
            assert relerr(z[n][i], got) < 1e-8, (i, n)
    assert (rs.n_normal, rs.n_uniform) == (int(z["n_normal"]), int(z["n_uniform"]))


def test_loglike_with_init_arrays_matches_reference():
    """Per-model `av_init` / `rv_init` (reference fitting.py:697-707)."""
    z = np.load(os.path.join(GOLDEN, "init_loglike.npz"))
    for tag, kw in (("both", dict(av_init=z["av_init"], rv_init=z["rv_init"])),
                    ("av", dict(av_init=z["av_init"]))):
        out = O.loglike(z["flux"], z["err"], z["mask"], z["models"], parallax=float(z["parallax"]),
                        parallax_err=float(z["parallax_err"]), return_vals=True, **kw)
        assert out[1] == int(z[tag + "_Ndim"])
        for name, got in zip("lnl Ndim chi2 scale av rv icov".split(), out):
            if name != "Ndim":
                assert relerr(z["%s_%s" % (tag, name)], got) < TOL, (tag, name)
    # ... and the starting point matters: the default start gives other numbers
    dflt = O.loglike(z["flux"], z["err"], z["mask"], z["models"], parallax=float(z["parallax"]),
                     parallax_err=float(z["parallax_err"]), return_vals=True)
    assert relerr(z["both_av"], dflt[4]) > 1e-6


def test_fit_star_cdf_thresholding_matches_reference():
    """`wt_thresh=None`: CDF thresholding as the reference does it (ascending sort: the
    most probable models are dropped, the rest handed on in sort order; fitting.py:992-998,
    1017-1022), with and without the `Nsel_max` clip."""
    z = np.load(os.path.join(GOLDEN, "fit_cdf.npz"))
    models, labels, lmask = synth.make_grid(int(z["grid_nmodel"]), int(z["grid_nfilt"]),
                                            seed=int(z["grid_seed"]))
    names = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds "
             "dreds logwts").split()
    for i in range(len(z["flux"])):
        out = O.fit_star(z["flux"][i], z["err"][i], z["mask"][i], models, z["lnprior"], labels,
                         z["coords"][i], z["parallax"][i], z["parallax_err"][i],
                         np.random.RandomState(int(z["seed0"]) + i), galprior, Nmc_prior=12,
                         Ndraws=40, wt_thresh=None, cdf_thresh=2e-3, mem_lim=float(z["mem_lim"][i]))
        assert np.array_equal(out[0], z["sidxs"][i]), i
        for n, got in zip(names[1:], out[1:]):
            assert relerr(z[n][i], got) < 1e-9, (i, n)


def test_utils_small_helpers_vs_reference_golden():
    """The rest of `brutus.utils.__all__` (`_adjoint3`, `_inverse_transpose3`, `_dot3`, `_isPSD`,
    `_truncnorm_*`, `_get_seds`, `quantile`, `luptitude`, `inv_luptitude`, `add_mag`) against
    vectors generated by the upstream code (tools/gen_golden.py utilsmisc); every name the
    reference exports exists here, the downloaders say why they do not download."""
    import ast
    import re
    from brutus_amd import utils as U
    z = np.load(os.path.join(GOLDEN, "utils_misc.npz"))
    tol = dict(rtol=1e-13, atol=0)
    assert np.allclose(U._adjoint3(z["A"]), z["adj"], **tol)
    assert np.allclose(U._inverse_transpose3(z["A"]), z["invT"], **tol)
    assert np.allclose(U._dot3(z["A"], z["B"]), z["dot"], **tol)
    assert np.allclose(U._inverse3(z["A"]), np.swapaxes(z["invT"], -1, -2), rtol=1e-12, atol=0)
    got = [U._isPSD(m) for m in z["spd"]] + [U._isPSD(m) for m in z["notpd"]]
    assert got == [bool(v) for v in z["psd"]]
    assert np.allclose(U._truncnorm_pdf(z["x"].copy(), -1.5, 2.5, loc=1., scale=1.7), z["tn_pdf"], **tol)
    lp = U._truncnorm_logpdf(z["x"].copy(), -1.5, 2.5, loc=1., scale=1.7)
    assert np.array_equal(np.isfinite(lp), np.isfinite(z["tn_logpdf"]))
    fin = np.isfinite(lp)
    assert np.allclose(lp[fin], z["tn_logpdf"][fin], rtol=1e-13, atol=1e-15)
    sc = [U._truncnorm_pdf(0.3, -1.5, 2.5, 1., 1.7), U._truncnorm_pdf(9., -1.5, 2.5, 1., 1.7),
          U._truncnorm_logpdf(0.3, -1.5, 2.5, 1., 1.7), U._truncnorm_logpdf(9., -1.5, 2.5, 1., 1.7)]
    assert np.allclose(sc[:3], z["tn_scalar"][:3], **tol) and sc[3] == -np.inf == z["tn_scalar"][3]
    for k, rf in (("mag", False), ("flux", True)):
        sd, rvc, drv = U._get_seds(z["coeffs"], z["av"], z["rv"], return_flux=rf)
        assert np.allclose(sd, z["seds_" + k], **tol) and np.allclose(rvc, z["rvecs_" + k], **tol)
        assert np.allclose(drv, z["drvecs_" + k], **tol)
    assert np.allclose(U.quantile(z["samp"], z["q"]), z["quant"], **tol)
    assert np.allclose(U.quantile(z["samp"], z["q"], weights=z["wts"]), z["quant_w"], **tol)
    lm, le = U.luptitude(z["phot"], z["err"], skynoise=2e-9, zeropoints=3.)
    assert np.allclose(lm, z["lup"], **tol) and np.allclose(le, z["lup_err"], **tol)
    ip, ie = U.inv_luptitude(lm, le, skynoise=2e-9, zeropoints=3.)
    assert np.allclose(ip, z["ilup"], rtol=1e-12, atol=0) and np.allclose(ie, z["ilup_err"], rtol=1e-12, atol=0)
    assert np.allclose(U.add_mag(np.linspace(10, 20, 7), np.linspace(21, 9, 7), f1=0.7, f2=1.3), z["add"], **tol)
    with pytest.raises(ValueError):
        U.quantile(z["samp"], [1.5])
    with pytest.raises(NotImplementedError, match="fetch_grids"):
        U.fetch_grids()
    w = U._function_wrapper(lambda x, a, b=0: x * a + b, (3,), dict(b=1))
    assert w(2) == 7
    ref_all = ['_function_wrapper', '_adjoint3', '_inverse_transpose3', '_inverse3', '_dot3', '_isPSD',
               '_chisquare_logpdf', '_truncnorm_pdf', '_truncnorm_logpdf', '_get_seds', 'fetch_isos',
               'fetch_tracks', 'fetch_dustmaps', 'fetch_grids', 'fetch_offsets', 'fetch_nns',
               'load_models', 'load_offsets', 'quantile', 'draw_sar', 'sample_multivariate_normal',
               'magnitude', 'inv_magnitude', 'luptitude', 'inv_luptitude', 'add_mag', 'get_seds',
               'phot_loglike', 'photometric_offsets']
    assert [n for n in ref_all if n not in U.__all__ or not hasattr(U, n)] == []
