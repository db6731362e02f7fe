"""CPU: host-side pieces of the product that need no GPU -- `BruteForce._setup`
against the reference golden (tests/golden/setup.npz), host helpers against
tests/golden/helpers.npz, `lnpost`'s host stage against the oracle."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, galprior, relerr
from brutus_amd import fitting, pdf, synth, utils


def test_setup_matches_reference():
    z = np.load(os.path.join(GOLDEN, "setup.npz"))
    models, labels, lmask = synth.make_grid(int(z["grid_nmodel"]),
                                            int(z["grid_nfilt"]),
                                            seed=int(z["grid_seed"]))
    st = synth.make_stars(models, 8, seed=11)
    BF = fitting.BruteForce(models, labels, lmask)
    out = BF._setup(z["flux"], z["err"], z["mask"], None,
                    phot_offsets=z["offsets"], data_coords=st["coords"],
                    lngalprior=galprior, parallax=st["parallax"],
                    parallax_err=st["parallax_err"])
    assert relerr(z["out_flux"], out[0]) < 1e-15
    assert relerr(z["out_err"], out[1]) < 1e-15
    assert np.array_equal(z["out_mask"], out[2])
    assert relerr(z["lnprior"], out[5]) < 1e-13
    assert tuple(z["av_gauss"]) == tuple(map(float, out[8]))
    assert float(z["wt_thresh"]) == out[9]
    bad = z["mask"].copy()
    bad[5, :3] = False
    with pytest.raises(ValueError, match="fewer than 4 bands"):
        BF._setup(z["flux"], z["err"], bad, None, data_coords=st["coords"],
                  lngalprior=galprior)
    with pytest.raises(ValueError, match="initial threshold"):
        BF._setup(z["flux"], z["err"], z["mask"], None, data_coords=st["coords"],
                  lngalprior=galprior, logl_initthresh=0.1, ltol_subthresh=1e-2)
    with pytest.raises(ValueError, match="data_coords"):
        BF._setup(z["flux"], z["err"], z["mask"], None)


def test_host_helpers_match_reference():
    z = np.load(os.path.join(GOLDEN, "helpers.npz"))
    assert relerr(z["inv3_out"], utils._inverse3(z["inv3_in"])) < 1e-12
    assert relerr(z["chi2_df5"], utils._chisquare_logpdf(z["chi2_x"], 5)) < 1e-13
    assert relerr(z["chi2_df9"], utils._chisquare_logpdf(z["chi2_x"], 9)) < 1e-13
    mvn = utils.sample_multivariate_normal(z["mvn_mean"], z["mvn_cov"], size=11,
                                           rstate=np.random.RandomState(9))
    assert relerr(z["mvn_out"], mvn) < 1e-13
    assert relerr(z["imf_out"], pdf.imf_lnprior(z["imf_m"])) < 1e-13
    s, e = z["sp_scales"], z["sp_serrs"]
    assert relerr(z["sp_hi"], pdf.scale_parallax_lnprior(s, e, 1.0, 0.1)) < 1e-13
    assert relerr(z["sp_lo"], pdf.scale_parallax_lnprior(s, e, 1.0, 0.3)) < 1e-13
    assert relerr(z["sp_nan"], pdf.scale_parallax_lnprior(s, e, np.nan, 0.3)) == 0
    assert relerr(z["pl_out"], pdf.parallax_lnprior(np.sqrt(s), 1.1, 0.2)) < 1e-13
    assert relerr(z["p2s_hi"], np.array(pdf.parallax_to_scale(1.0, 0.1))) < 1e-15
    assert relerr(z["p2s_lo"], np.array(pdf.parallax_to_scale(1.0, 0.3))) < 1e-15
    mag, magerr = utils.magnitude(z["mag_flux"], z["mag_ferr"])
    assert relerr(z["mag_out"], mag) < 1e-15 and relerr(z["magerr_out"], magerr) < 1e-15


def test_lnpost_host_stage_matches_oracle():
    """Feed the product's `lnpost` the oracle's full-grid loglike results: the
    host stage (cuts, PSD repair, MC integral, RNG order) must agree."""
    from oracle import brutus_oracle as O
    models, labels, lmask = synth.make_grid(1500, 8, seed=9)
    st = synth.make_stars(models, 4, seed=10)
    lnprior = O.static_lnprior(labels, lmask)
    for i in range(4):
        par, pe = st["parallax"][i], st["parallax_err"][i]
        res = O.loglike(st["flux"][i], st["err"][i], st["mask"][i], models,
                        parallax=par, parallax_err=pe, return_vals=True)
        a = O.lnpost(tuple(np.copy(x) if isinstance(x, np.ndarray) else x for x in res),
                     parallax=par, parallax_err=pe, coord=st["coords"][i],
                     Nmc_prior=20, lnprior=lnprior, lngalprior=galprior,
                     dlabels=labels, rstate=np.random.RandomState(3),
                     apply_av_prior=False)
        b = fitting.lnpost(res, parallax=par, parallax_err=pe,
                           coord=st["coords"][i], Nmc_prior=20, lnprior=lnprior,
                           lngalprior=galprior, dlabels=labels,
                           rstate=np.random.RandomState(3), apply_av_prior=False)
        assert np.array_equal(a[0], b[0])
        for x, y in zip(a[1:], b[1:]):
            assert relerr(x, y) < 1e-9


def test_bands_in_use_and_pdf_names():
    """Host logic of the band compaction (no GPU): which bands a call needs, and the
    reference's `brutus.pdf` names (pdf.py:30-35) importable from `brutus_amd.pdf`."""
    from brutus_amd import fitting
    m = np.zeros((3, 49), bool)
    m[:, [3, 9, 10, 11, 40]] = True
    m[1, 9] = False
    assert np.array_equal(fitting._bands_in_use(m, 49), [3, 9, 10, 11, 40])
    assert fitting._bands_in_use(np.ones((2, 12), bool), 12) is None
    k = np.ones((2, 12), bool)
    k[:, 11] = False                      # 11 of 12: the padded band count stays 12
    assert fitting._bands_in_use(k, 12) is None
    k[:, 6:] = False                      # 6 of 12 -> the 8-band kernels
    assert np.array_equal(fitting._bands_in_use(k, 12), np.arange(6))
    w = np.ones((1, 40), bool)            # 40 used of 40: nothing to drop (DeviceGrid raises)
    assert fitting._bands_in_use(w, 40) is None
    from brutus_amd.pdf import (gal_lnprior, logn_disk, logn_halo, logp_feh,   # noqa: F401
                                logp_age_from_feh, dust_lnprior, imf_lnprior)
    import brutus_amd.pdf as pdf
    ref_all = ["imf_lnprior", "ps1_MrLF_lnprior", "parallax_lnprior",
               "scale_parallax_lnprior", "parallax_to_scale", "logn_disk", "logn_halo",
               "logp_feh", "logp_age_from_feh", "gal_lnprior", "dust_lnprior"]
    assert [n for n in ref_all if n not in pdf.__all__] == []
