"""CPU restatement (numpy, float64) of the brutus per-star grid-likelihood path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker*: it may be imported by
`tests/`, by `__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline` leg,
and by nothing else.  The product package (`brutus_amd/`) never imports it and
fails loudly when its HIP library is missing.

Parity pin: every function below was checked in the build container against
the upstream reference imported under `tools/ref_shim.py`
(`tests/test_oracle_vs_reference.py`), and against the committed golden vectors
under `tests/golden/` that `tools/gen_golden.py` generated from that import.
The reference ships no tests or golden vectors of its own.

All `file:line` citations are relative to the upstream repository root
(`brutus/fitting.py`, `brutus/utils.py`, `brutus/pdf.py`).

The arithmetic follows the reference operation-by-operation (same association
order, per-band sums accumulated sequentially like the numba loops, numpy
pairwise `np.sum(axis=1)` where the reference uses it), vectorised over the
model axis.  Model-grid coefficients are float32 values promoted to float64,
which is what numba does with the `float32` grid `load_models` returns.
"""
from math import gamma as _gamma
from math import log as _log

import numpy as np
from scipy.special import logsumexp

__all__ = ["get_seds", "optimize_fit_mag", "get_sed_mle", "optimize_fit_flux",
           "loglike", "chisquare_logpdf", "inverse3",
           "sample_multivariate_normal", "imf_lnprior", "parallax_lnprior",
           "scale_parallax_lnprior", "parallax_to_scale", "lnpost",
           "static_lnprior", "clean_mask", "fit_star", "magnitude"]

AV_REG = 0.05  # fitting.py:431 (`av_reg`)
RV_REG = 0.1   # fitting.py:431 (`rv_reg`)


# ---------------------------------------------------------------------------
# utils.py
# ---------------------------------------------------------------------------
def get_seds(mag_coeffs, av, rv, return_flux=False):
    """utils.py:286-347 (`_get_seds`).  mag_coeffs (N, Nb, 3) f64."""
    mags = mag_coeffs[:, :, 0]
    r0 = mag_coeffs[:, :, 1]
    dr = mag_coeffs[:, :, 2]
    av = np.asarray(av, dtype=np.float64)
    rv = np.asarray(rv, dtype=np.float64)
    drvecs = np.array(dr, dtype=np.float64)                  # utils.py:337
    rvecs = r0 + rv[:, None] * dr                            # utils.py:338
    seds = mags + av[:, None] * rvecs                        # utils.py:339
    if return_flux:
        fac = -0.4 * _log(10.)                               # utils.py:328
        seds = 10. ** (-0.4 * seds)                          # utils.py:343
        rvecs = rvecs * (fac * seds)                         # utils.py:344
        drvecs = drvecs * (fac * seds)                       # utils.py:345
    return seds, rvecs, drvecs


def chisquare_logpdf(x, df):
    """utils.py:130-176 (`_chisquare_logpdf`, loc=0, scale=1)."""
    y = np.array(x, dtype=np.float64)
    keys = y <= 0
    y[keys] = 0.1
    ans = -_log(2 ** (df / 2.) * _gamma(df / 2.))
    with np.errstate(all="ignore"):
        ans = ans + (df / 2. - 1.) * np.log(y) - y / 2. - _log(1)
    ans[keys] = -np.inf
    return ans


def inverse3(A):
    """utils.py:71-114 (`_adjoint3`, `_dot3`, `_inverse_transpose3`,
    `_inverse3`): adjugate by row cross products, determinant as the mean of
    the three row dots, then transpose."""
    A = np.asarray(A, dtype=np.float64)
    AI = np.empty_like(A)
    for i in range(3):
        AI[..., i, :] = np.cross(A[..., i - 2, :], A[..., i - 1, :])
    det = np.einsum('...i,...i->...', AI, A).mean(axis=-1)
    with np.errstate(all="ignore"):
        return np.swapaxes(AI / det[..., None, None], -1, -2)


def sample_multivariate_normal(mean, cov, size, rstate, eps=1e-30):
    """utils.py:845-905, many-distribution branch.  Returns (dim, size, N)."""
    N, d = np.shape(mean)
    K = cov + eps * np.full((N, d, d), np.identity(d))
    L = np.linalg.cholesky(K)
    z = rstate.normal(loc=0, scale=1, size=d * size * N).reshape(N, d, size)
    ans = np.repeat(mean[:, :, np.newaxis], size, axis=2) + np.matmul(L, z)
    ans = np.swapaxes(ans, 0, 1)
    ans = np.swapaxes(ans, 1, 2)
    return ans


def magnitude(phot, err):
    """utils.py:908-940."""
    with np.errstate(all="ignore"):
        mag = -2.5 * np.log10(phot / 1.)
        mag_err = 2.5 / np.log(10.) * err / phot
    return mag, mag_err


# ---------------------------------------------------------------------------
# pdf.py
# ---------------------------------------------------------------------------
def imf_lnprior(mgrid, alpha_low=1.3, alpha_high=2.3, mass_break=0.5):
    """pdf.py:38-108 (Kroupa IMF, single stars)."""
    mgrid = np.asarray(mgrid, dtype=np.float64)
    lnprior = np.zeros_like(mgrid) - np.inf
    low = (mgrid <= mass_break) & (mgrid > 0.08)
    lnprior[low] = -alpha_low * np.log(mgrid[low])
    high = mgrid > mass_break
    lnprior[high] = (-alpha_high * np.log(mgrid[high])
                     + (alpha_high - alpha_low) * np.log(mass_break))
    norm_low = mass_break ** (1. - alpha_low) / (alpha_high - 1.)
    norm_high = 0.08 ** (1. - alpha_low) / (alpha_low - 1.)
    norm_high -= mass_break ** (1. - alpha_low) / (alpha_low - 1.)
    norm = norm_low + norm_high
    return lnprior - np.log(norm)


def parallax_lnprior(parallaxes, p_meas, p_err):
    """pdf.py:144-175."""
    if np.isfinite(p_meas) and np.isfinite(p_err):
        chi2 = (parallaxes - p_meas) ** 2 / p_err ** 2
        lnorm = np.log(2. * np.pi * p_err ** 2)
        return -0.5 * (chi2 + lnorm)
    return np.zeros_like(parallaxes)


def parallax_to_scale(p_meas, p_err, snr_lim=4.):
    """pdf.py:225-260."""
    if p_meas / p_err > snr_lim:
        pm, pe = max(0., p_meas), p_err
        s_mean = pm ** 2 + pe ** 2
        s_std = np.sqrt(2 * pe ** 4 + 4 * pm ** 2 * pe ** 2)
    else:
        s_mean, s_std = 1e-20, 1e20
    return s_mean, s_std


def scale_parallax_lnprior(scales, scale_errs, p_meas, p_err, snr_lim=4.):
    """pdf.py:178-222."""
    if (np.isfinite(p_meas) and np.isfinite(p_err)
            and p_meas / p_err > snr_lim):
        s_mean, s_std = parallax_to_scale(p_meas, p_err)
        svar_tot = s_std ** 2 + scale_errs ** 2
        chi2 = (scales - s_mean) ** 2 / svar_tot
        lnorm = np.log(2. * np.pi * svar_tot)
        return -0.5 * (chi2 + lnorm)
    return np.zeros_like(scales)


# ---------------------------------------------------------------------------
# fitting.py: the three numba loops, vectorised over models
# ---------------------------------------------------------------------------
def _clamp_step(step, lo, hi, cur):
    """The two sequential `if` clamps (e.g. fitting.py:195-198).  NaN steps
    fail both comparisons and stay NaN, like the scalar code."""
    step = np.where(step < lo - cur, lo - cur, step)
    step = np.where(step > hi - cur, hi - cur, step)
    return step


def optimize_fit_mag(resid, rvecs, drvecs, av, rv, stepsize, mags_var,
                     avlim, av_gauss, rvlim, rv_gauss, tol, init_thresh,
                     max_sweeps=100000):
    """fitting.py:141-264 (`_optimize_fit_mag` main loop, without the final
    `_get_sed_mle` call, which `loglike` below does explicitly).

    resid, rvecs, drvecs: (N, Nb) f64, updated and returned.
    mags_var: (Nb,) -- the reference's (Nmodel, Nb) array has identical rows.
    Returns resid, rvecs, av, rv, nsweeps (K1).
    """
    N, Nb = resid.shape
    avmin, avmax = avlim
    rvmin, rvmax = rvlim
    Av_mean, Av_std = av_gauss
    Rv_mean, Rv_std = rv_gauss
    Av_varinv, Rv_varinv = 1. / Av_std ** 2, 1. / Rv_std ** 2
    log_init_thresh = _log(init_thresh)                      # fitting.py:150
    resid = np.array(resid, dtype=np.float64)
    rvecs = np.array(rvecs, dtype=np.float64)
    av = np.array(av, dtype=np.float64)
    rv = np.array(rv, dtype=np.float64)

    # constants, fitting.py:158-164
    s_den = np.zeros(N)
    rp_den = np.zeros(N)
    srp_mix = np.zeros(N)
    for j in range(Nb):
        s_den += 1. / mags_var[j]
        rp_den += drvecs[:, j] * drvecs[:, j] / mags_var[j]
        srp_mix += drvecs[:, j] / mags_var[j]

    nsweeps = 0
    with np.errstate(all="ignore"):
        while True:
            nsweeps += 1
            # Av solve, fitting.py:176-192
            a_den = np.zeros(N)
            sa_mix = np.zeros(N)
            resid_s = np.zeros(N)
            resid_a = np.zeros(N)
            for j in range(Nb):
                a_den += rvecs[:, j] * rvecs[:, j] / mags_var[j]
                sa_mix += rvecs[:, j] / mags_var[j]
                resid_s += resid[:, j] / mags_var[j]
                resid_a += resid[:, j] * rvecs[:, j] / mags_var[j]
            resid_a += (Av_mean - av) * Av_varinv
            a_den += Av_varinv
            sa_idet = 1. / (s_den * a_den - sa_mix * sa_mix)
            dav = sa_idet * (s_den * resid_a - sa_mix * resid_s)
            dav = dav * stepsize
            dav = _clamp_step(dav, avmin, avmax, av)         # fitting.py:195-198
            av = av + dav                                    # fitting.py:201
            resid = resid - dav[:, None] * rvecs             # fitting.py:203-204

            # Rv solve, fitting.py:207-224
            r_den = rp_den * av * av
            sr_mix = srp_mix * av
            resid_s = np.zeros(N)
            resid_r = np.zeros(N)
            for j in range(Nb):
                resid_s += resid[:, j] / mags_var[j]
                resid_r += resid[:, j] * drvecs[:, j] / mags_var[j]
            resid_r = resid_r * av
            resid_r += (Rv_mean - rv) * Rv_varinv
            r_den += Rv_varinv
            sr_idet = 1. / (s_den * r_den - sr_mix * sr_mix)
            drv = sr_idet * (s_den * resid_r - sr_mix * resid_s)
            drv = drv * stepsize
            drv = _clamp_step(drv, rvmin, rvmax, rv)         # fitting.py:227-230
            rv = rv + drv                                    # fitting.py:233
            resid = resid - (av * drv)[:, None] * drvecs     # fitting.py:236
            rvecs = rvecs + drv[:, None] * drvecs            # fitting.py:237

            # chi2 / logwt, fitting.py:240-243
            chi2 = np.zeros(N)
            for j in range(Nb):
                chi2 += resid[:, j] * resid[:, j] / mags_var[j]
            logwt = -0.5 * chi2

            # global convergence test, fitting.py:246-264
            good = logwt[logwt > -1e300]
            max_logwt = good.max() if good.size else -1e300
            sel = logwt > max_logwt + log_init_thresh
            err = -1e300
            for step in (np.abs(dav[sel]), np.abs(drv[sel])):
                step = step[step > err]                      # drops NaN
                if step.size:
                    err = max(err, step.max())
            if err < tol or nsweeps >= max_sweeps:
                break
    return resid, rvecs, av, rv, nsweeps


def get_sed_mle(data, tot_var, mag_coeffs, av, rv, av_gauss, rv_gauss):
    """fitting.py:502-576 (`_get_sed_mle`).

    data, tot_var: (Nb,).  mag_coeffs (N, Nb, 3) f64.
    Returns models, rvecs, drvecs (scaled), scale, icov_sar (N,3,3), resid.
    """
    Av_mean, Av_std = av_gauss
    Rv_mean, Rv_std = rv_gauss
    with np.errstate(all="ignore"):
        models, rvecs, drvecs = get_seds(mag_coeffs, av, rv, return_flux=True)
        N, Nb = models.shape
        # scale, fitting.py:511-518
        s_num = np.zeros(N)
        s_den = np.zeros(N)
        for j in range(Nb):
            s_num += models[:, j] * data[j] / tot_var[j]
            s_den += models[:, j] * models[:, j] / tot_var[j]
        scale = s_num / s_den
        scale = np.where(scale <= 1e-20, 1e-20, scale)

        sr_mix = np.zeros(N)
        sa_mix = np.zeros(N)
        a_den = np.zeros(N)
        r_den = np.zeros(N)
        ar_mix = np.zeros(N)
        Av_varinv, Rv_varinv = 1. / Av_std ** 2, 1. / Rv_std ** 2
        a_den_reg, r_den_reg = 1. / AV_REG ** 2, 1. / RV_REG ** 2
        resid = np.empty((N, Nb))
        for j in range(Nb):                                  # fitting.py:527-553
            models_int = 10. ** (-0.4 * mag_coeffs[:, j, 0])
            reddening = models[:, j] - models_int
            models[:, j] = models[:, j] * scale
            resid[:, j] = data[j] - models[:, j]
            sr_mix += drvecs[:, j] * ((models[:, j] - resid[:, j]) / tot_var[j])
            sa_mix += rvecs[:, j] * ((models[:, j] - resid[:, j]) / tot_var[j])
            rvecs[:, j] = rvecs[:, j] * scale
            drvecs[:, j] = drvecs[:, j] * scale
            reddening = reddening * scale
            ar_mix += drvecs[:, j] * ((reddening - resid[:, j]) / tot_var[j])
            a_den += rvecs[:, j] * rvecs[:, j] / tot_var[j]
            r_den += drvecs[:, j] * drvecs[:, j] / tot_var[j]
        a_den += Av_varinv                                   # fitting.py:556-561
        r_den += Rv_varinv
        a_den += a_den_reg
        r_den += r_den_reg

    icov = np.zeros((N, 3, 3))                               # fitting.py:564-574
    icov[:, 0, 0] = s_den
    icov[:, 1, 1] = a_den
    icov[:, 2, 2] = r_den
    icov[:, 0, 1] = icov[:, 1, 0] = sa_mix
    icov[:, 0, 2] = icov[:, 2, 0] = sr_mix
    icov[:, 1, 2] = icov[:, 2, 1] = ar_mix
    return models, rvecs, drvecs, scale, icov, resid


def optimize_fit_flux(data, tot_var, rvecs, drvecs, av, rv, mag_coeffs, resid,
                      stepsize, avlim, av_gauss, rvlim, rv_gauss):
    """fitting.py:365-427 (`_optimize_fit_flux`): one damped 1-D step in Av
    and in Rv (both from the OLD residuals), then `_get_sed_mle`."""
    N, Nb = resid.shape
    avmin, avmax = avlim
    rvmin, rvmax = rvlim
    Av_mean, Av_std = av_gauss
    Rv_mean, Rv_std = rv_gauss
    Av_varinv, Rv_varinv = 1. / Av_std ** 2, 1. / Rv_std ** 2
    with np.errstate(all="ignore"):
        a_num = np.zeros(N)
        a_den = np.zeros(N)
        for j in range(Nb):
            a_num += rvecs[:, j] * resid[:, j] / tot_var[j]
            a_den += rvecs[:, j] * rvecs[:, j] / tot_var[j]
        a_num += (Av_mean - av) * Av_varinv
        a_den += Av_varinv
        dav = a_num / a_den
        dav = dav * stepsize
        r_num = np.zeros(N)
        r_den = np.zeros(N)
        for j in range(Nb):
            r_num += drvecs[:, j] * resid[:, j] / tot_var[j]
            r_den += drvecs[:, j] * drvecs[:, j] / tot_var[j]
        r_num += (Rv_mean - rv) * Rv_varinv
        r_den += Rv_varinv
        drv = r_num / r_den
        drv = drv * stepsize
        dav = _clamp_step(dav, avmin, avmax, av)
        av = av + dav
        drv = _clamp_step(drv, rvmin, rvmax, rv)
        rv = rv + drv
    return (av, rv) + get_sed_mle(data, tot_var, mag_coeffs, av, rv,
                                  av_gauss, rv_gauss)


def clean_mask(data, data_err, data_mask):
    """fitting.py:706-710."""
    with np.errstate(all="ignore"):
        clean = np.isfinite(data) & np.isfinite(data_err) & (data_err > 0.)
    return np.asarray(data_mask, dtype=bool) & clean


def loglike(data, data_err, data_mask, mag_coeffs,
            avlim=(0., 20.), av_gauss=(0., 1e6),
            rvlim=(1., 8.), rv_gauss=(3.32, 0.18),
            dim_prior=True, ltol=3e-2, ltol_subthresh=1e-2, init_thresh=5e-3,
            parallax=None, parallax_err=None, return_vals=False, trace=None,
            max_sweeps=100000, av_init=None, rv_init=None):
    """fitting.py:579-820 (`loglike`).  `mag_coeffs` may be float32 or float64;
    float32 values are promoted to float64 before any arithmetic.

    If `trace` is a dict it receives K1, K2 and init_sel."""
    if init_thresh > ltol_subthresh:                         # fitting.py:691-693
        raise ValueError("The initial threshold must be smaller than or equal "
                         "to the final threshold applied to be useful!")
    if av_gauss is None:
        av_gauss = (0., 1e6)
    data = np.asarray(data, dtype=np.float64)
    data_err = np.asarray(data_err, dtype=np.float64)
    Nmodels = mag_coeffs.shape[0]
    if av_init is None:                                      # fitting.py:697-703
        av_init = np.zeros(Nmodels) + av_gauss[0]
    if rv_init is None:
        rv_init = np.zeros(Nmodels) + rv_gauss[0]
    av_init = np.array(av_init, dtype=np.float64)
    rv_init = np.array(rv_init, dtype=np.float64)

    mask = clean_mask(data, data_err, data_mask)
    Ndim = int(np.sum(mask))
    flux, fluxerr = data[mask], data_err[mask]
    mcoeffs = np.asarray(mag_coeffs[:, mask, :], dtype=np.float64)
    tot_var = np.square(fluxerr)                             # (Nb,)

    with np.errstate(all="ignore"):                          # fitting.py:719-725
        mags = -2.5 * np.log10(flux)
        mags_var = np.square(2.5 / np.log(10.)) * tot_var / np.square(flux)
        mclean = np.isfinite(mags)
        mags[~mclean], mags_var[~mclean] = 0., 1e50

    models, rvecs, drvecs = get_seds(mcoeffs, av_init, rv_init)
    mtol = 2.5 * ltol
    resid = mags - models
    stepsize = np.ones(Nmodels)
    resid, rvecs, av, rv, K1 = optimize_fit_mag(
        resid, rvecs, drvecs, av_init, rv_init, stepsize, mags_var,
        avlim, av_gauss, rvlim, rv_gauss, mtol, init_thresh,
        max_sweeps=max_sweeps)
    (models, rvecs, drvecs, scale, icov_sar,
     resid) = get_sed_mle(flux, tot_var, mcoeffs, av, rv, av_gauss, rv_gauss)

    tv2 = np.repeat(tot_var[np.newaxis, :], Nmodels, axis=0)  # fitting.py:716
    with np.errstate(all="ignore"):
        # cull, fitting.py:743-768
        chi2 = np.sum(np.square(resid) / tv2, axis=1)
        lnl = -0.5 * chi2
        lnl_p = lnl
        if parallax is not None and parallax_err is not None:
            if np.isfinite(parallax) and np.isfinite(parallax_err):
                par = np.sqrt(scale)
                chi2_p = (par - parallax) ** 2 / parallax_err ** 2
                lnl_p = lnl - 0.5 * chi2_p
        lnl_sel = lnl_p > np.max(lnl_p) + np.log(init_thresh)
        init_sel = np.where(lnl_sel)[0]
        tv_s = tv2[init_sel]
        rvecs = rvecs[init_sel]
        drvecs = drvecs[init_sel]
        av_new = av[init_sel]
        rv_new = rv[init_sel]
        mc_s = mcoeffs[init_sel]
        resid = resid[init_sel]

        # flux phase, fitting.py:778-803
        lnl_old, lerr = -1e300, 1e300
        stepsize = np.ones(Nmodels)[init_sel]
        rescaling = 1.2
        ln_sub = np.log(ltol_subthresh)
        K2 = 0
        while lerr > ltol:
            K2 += 1
            (av_new, rv_new, models, rvecs, drvecs, scale_new, icov_new,
             resid) = optimize_fit_flux(flux, tot_var, rvecs, drvecs, av_new,
                                        rv_new, mc_s, resid, stepsize, avlim,
                                        av_gauss, rvlim, rv_gauss)
            chi2_new = np.sum(np.square(resid) / tv_s, axis=1)
            lnl_new = -0.5 * chi2_new
            sel = np.where(lnl_new > np.max(lnl_new) + ln_sub)[0]
            lerr = np.max(np.abs(lnl_new - lnl_old)[sel])
            stepsize[lnl_new < lnl_old] /= rescaling
            lnl_old = lnl_new

        # fitting.py:806-810
        lnl_new = lnl_new + -0.5 * (Ndim * np.log(2. * np.pi)
                                    + np.sum(np.log(tv_s), axis=1))
        lnl = np.array(lnl)
        lnl[init_sel], chi2[init_sel] = lnl_new, chi2_new
        scale[init_sel], av[init_sel], rv[init_sel] = scale_new, av_new, rv_new
        icov_sar[init_sel] = icov_new
        if dim_prior:                                        # fitting.py:813-815
            lnl = chisquare_logpdf(chi2, Ndim - 3)
    if trace is not None:
        trace["K1"], trace["K2"], trace["init_sel"] = K1, K2, init_sel
    if return_vals:
        return lnl, Ndim, chi2, scale, av, rv, icov_sar
    return lnl, Ndim, chi2


# ---------------------------------------------------------------------------
# fitting.py: lnpost and the per-star tail of BruteForce._fit
# ---------------------------------------------------------------------------
def lnpost(results, parallax=None, parallax_err=None, coord=None,
           Nmc_prior=100, lnprior=None, wt_thresh=1e-3, cdf_thresh=2e-3,
           lngalprior=None, lndustprior=None, dustfile=None, dlabels=None,
           avlim=(0., 20.), rvlim=(1., 8.), rstate=None,
           apply_av_prior=True, mem_lim=8000.):
    """fitting.py:934-1107 (`lnpost`), both thresholding branches (`wt_thresh=None`:
    CDF thresholding exactly as the reference does it, ascending sort and all).
    `lngalprior` must be supplied (the reference's default hook needs astropy: unpinned)."""
    if wt_thresh is None and cdf_thresh is None:             # fitting.py:935-936
        wt_thresh = -np.inf
    if lngalprior is None:
        raise ValueError("oracle needs an explicit `lngalprior` hook")
    if coord is None:
        coord = np.zeros(2)
    Nsel_max = int(mem_lim / Nmc_prior / 4.0e-4)             # fitting.py:969-970
    lnlike, Ndim, chi2, scales, avs, rvs, icovs_sar = results

    with np.errstate(all="ignore"):
        if parallax is not None and parallax_err is not None:  # :976-982
            ds2 = icovs_sar[:, 0, 0]
            scales_err = 1. / np.sqrt(np.abs(ds2))
            lnprob = lnlike + scale_parallax_lnprior(scales, scales_err,
                                                     parallax, parallax_err)
        else:
            lnprob = np.array(lnlike)
        lnprob = np.array(lnprob)
        lnprob[~np.isfinite(lnprob)] = -1e300                # :983-985
        # the reference aliases lnprob and lnlike when the prior term is
        # the zero array? No: `lnlike + zeros` is a new array; only the
        # `parallax is None` branch aliases (fitting.py:982).
        if parallax is None or parallax_err is None:
            lnlike = lnprob

        if wt_thresh is not None:
            lwt_min = np.log(wt_thresh) + np.max(lnprob)     # :988-991
            sel = np.where(lnprob > lwt_min)[0]
        else:                                                # :992-998
            idx_sort = np.argsort(lnprob)
            prob = np.exp(lnprob - logsumexp(lnprob))
            cdf = np.cumsum(prob[idx_sort])
            sel = idx_sort[cdf <= (1. - cdf_thresh)]

        lnp = lnlike[sel]                                    # :1000-1010
        lnp = lnp + lnprior[sel]
        dist = 1. / np.sqrt(scales[sel])
        lnp = lnp + lngalprior(dist, coord,
                               labels=None if dlabels is None else dlabels[sel])
        if apply_av_prior:
            lnp = lnp + lndustprior(dist, coord, avs[sel], dustfile=dustfile)

        if wt_thresh is not None:
            lwt_min = np.log(wt_thresh) + np.max(lnp)        # :1013-1016
            sel = sel[np.where(lnp > lwt_min)[0]]
        else:                                                # :1017-1022
            idx_sort = np.argsort(lnp)
            prob = np.exp(lnp - logsumexp(lnp))
            cdf = np.cumsum(prob[idx_sort])
            sel = sel[idx_sort[cdf <= (1. - cdf_thresh)]]
        lnp = lnlike[sel] + lnprior[sel]                     # :1023
        scale, av, rv = scales[sel], avs[sel], rvs[sel]
        icov_sar = np.array(icovs_sar[sel])
        Nsel = len(sel)

        if Nsel > Nsel_max:                                  # :1029-1036
            idx_sort = np.argsort(lnp)[::-1][:Nsel_max]
            lnp = lnp[idx_sort]
            scale, av, rv = scale[idx_sort], av[idx_sort], rv[idx_sort]
            icov_sar = icov_sar[idx_sort]
            sel = sel[idx_sort]
            Nsel = len(sel)

        cov_sar = inverse3(icov_sar)                         # :1039
        not_psd = np.where(~np.all(np.linalg.eigvals(cov_sar) > 0,
                                   axis=1))[0]               # :1042
        width = 0.02
        count = 1
        while len(not_psd) > 0:                              # :1045-1065
            sfracs = scale[not_psd] * width
            i1 = cov_sar[not_psd][:, 0, 0] <= 0
            i2 = cov_sar[not_psd][:, 1, 1] <= 0
            i3 = cov_sar[not_psd][:, 2, 2] <= 0
            s1 = i1 | (~i2 & ~i3)
            s2 = i2 | (~i1 & ~i3)
            s3 = i3 | (~i1 & ~i2)
            add = np.zeros((len(not_psd), 3, 3))
            add[:, 0, 0] = count / sfracs ** 2 * s1
            add[:, 1, 1] = count / width ** 2 * s2
            add[:, 2, 2] = count / width ** 2 * s3
            icov_sar[not_psd] += add
            cov_sar[not_psd] = inverse3(icov_sar[not_psd])
            new_idx = np.where(~np.all(
                np.linalg.eigvals(cov_sar[not_psd]) > 0, axis=1))[0]
            not_psd = not_psd[new_idx]
            count *= 2

        # Monte Carlo integral, :1068-1098
        s_mc, a_mc, r_mc = sample_multivariate_normal(
            np.transpose([scale, av, rv]), cov_sar, size=Nmc_prior,
            rstate=rstate)
        if dlabels is not None:
            dlabels_mc = np.tile(dlabels[sel], Nmc_prior).reshape(-1, Nsel)
        else:
            dlabels_mc = None
        par_mc = np.sqrt(s_mc)
        dist_mc = 1. / par_mc
        lnp_mc = lngalprior(dist_mc, coord, labels=dlabels_mc)
        if apply_av_prior:
            lnp_mc = lnp_mc + lndustprior(dist_mc, coord, a_mc,
                                          dustfile=dustfile)
        if parallax is not None and parallax_err is not None:
            lnp_mc = lnp_mc + parallax_lnprior(par_mc, parallax, parallax_err)
        inbounds = ((s_mc >= 1e-20)
                    & (a_mc >= avlim[0]) & (a_mc <= avlim[1])
                    & (r_mc >= rvlim[0]) & (r_mc <= rvlim[1]))
        lnp_mc = np.array(lnp_mc)
        lnp_mc[~inbounds] = -1e300
        Nmc_prior_eff = np.sum(inbounds, axis=0)
        lnp = lnp + (logsumexp(lnp_mc, axis=0) - np.log(Nmc_prior_eff))
        lnp[~np.isfinite(lnp)] = -1e300                      # :1103-1105
    return sel, cov_sar, lnp, dist_mc.T, a_mc.T, r_mc.T, lnp_mc.T


def ps1_MrLF_lnprior(Mr, table=None):
    """pdf.py:111-141: `interp1d(grid_Mr, grid_lnp, fill_value='extrapolate')` of the
    reference's bundled two-column data table (read here as DATA from the copy the
    package ships, brutus_amd/PSMrLF_lnprior.dat)."""
    import os
    from scipy.interpolate import interp1d
    if table is None:
        table = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "brutus_amd", "PSMrLF_lnprior.dat")
    grid_Mr, grid_lnp = np.loadtxt(table).T
    return interp1d(grid_Mr, grid_lnp, fill_value='extrapolate')(Mr)


def static_lnprior(labels, labels_mask, apply_agewt=True, apply_grad=True):
    """fitting.py:1334-1359 (`_setup`): IMF prior on `mini` (PS1 luminosity function on
    `Mr` when the grid has no `mini`, :1338-1346), age weights and grid-spacing terms."""
    if 'mini' in labels.dtype.names:
        lnprior = imf_lnprior(labels['mini'])
    else:
        lnprior = ps1_MrLF_lnprior(labels['Mr'])
    if apply_agewt and 'agewt' in labels.dtype.names:
        with np.errstate(all="ignore"):
            lnprior = lnprior + np.log(np.abs(labels['agewt']))
    if apply_grad:
        for l in labels.dtype.names:
            label = labels[l]
            if labels_mask[l][0]:
                ulabel = np.unique(label)
                if len(ulabel) > 1:
                    lngrad = np.log(np.gradient(ulabel))
                    lnprior = lnprior + np.interp(label, ulabel, lngrad)
    return lnprior


def fit_star(data, data_err, data_mask, models, lnprior, labels, coord,
             parallax, parallax_err, rstate, lngalprior, lndustprior=None,
             Nmc_prior=50, avlim=(0., 20.), av_gauss=None, rvlim=(1., 8.),
             rv_gauss=(3.32, 0.18), wt_thresh=1e-3, Ndraws=250,
             dim_prior=True, ltol=3e-2, ltol_subthresh=1e-2,
             init_thresh=5e-3, mem_lim=8000., return_distreds=True, cdf_thresh=2e-3):
    """One iteration of the star loop of `BruteForce._fit`
    (fitting.py:1980-2065), after `_setup`.  `av_gauss=None` with
    `lndustprior=None` follows fitting.py:1396-1398 (flat prior (0, 1e6))."""
    apply_av_prior = False
    if lndustprior is None and av_gauss is None:
        av_gauss = (0, 1e6)
    elif av_gauss is None:
        apply_av_prior = True
    results = loglike(data, data_err, data_mask, models, avlim=avlim,
                      av_gauss=av_gauss, rvlim=rvlim, rv_gauss=rv_gauss,
                      dim_prior=dim_prior, ltol=ltol,
                      ltol_subthresh=ltol_subthresh, init_thresh=init_thresh,
                      parallax=parallax, parallax_err=parallax_err,
                      return_vals=True)
    return finish_star(results, lnprior, labels, coord, parallax, parallax_err, rstate,
                       lngalprior, lndustprior=lndustprior, Nmc_prior=Nmc_prior,
                       avlim=avlim, rvlim=rvlim, wt_thresh=wt_thresh, Ndraws=Ndraws,
                       mem_lim=mem_lim, return_distreds=return_distreds,
                       apply_av_prior=apply_av_prior, cdf_thresh=cdf_thresh)


def finish_star(results, lnprior, labels, coord, parallax, parallax_err, rstate,
                lngalprior, lndustprior=None, Nmc_prior=50, avlim=(0., 20.),
                rvlim=(1., 8.), wt_thresh=1e-3, Ndraws=250, mem_lim=8000.,
                return_distreds=True, apply_av_prior=False, cdf_thresh=2e-3):
    """`lnpost` + the resampling tail of the star loop (fitting.py:2010-2065) from
    full-grid `loglike` results."""
    lnlike, Ndim, chi2, scales, avs, rvs, icovs = results
    sel, cov_sar, lnprob, dists, reds, dreds, logwts = lnpost(
        results, parallax=parallax, parallax_err=parallax_err, coord=coord,
        Nmc_prior=Nmc_prior, lnprior=lnprior, wt_thresh=wt_thresh, cdf_thresh=cdf_thresh,
        lngalprior=lngalprior, lndustprior=lndustprior, dlabels=labels,
        avlim=avlim, rvlim=rvlim, rstate=rstate,
        apply_av_prior=apply_av_prior, mem_lim=mem_lim)
    Nsel = len(sel)
    with np.errstate(all="ignore"):
        if np.isfinite(parallax) and np.isfinite(parallax_err):  # :2025-2030
            chi2 = chi2 + ((np.sqrt(scales) - parallax) ** 2
                           / parallax_err ** 2)
            Ndim += 1
        levid = logsumexp(lnprob)                            # :2033-2034
        chi2min = np.min(chi2[sel])
        wt = np.exp(lnprob - levid)
        wt /= wt.sum()
        idxs = rstate.choice(Nsel, size=Ndraws, p=wt)        # :2039
        sidxs = sel[idxs]
        scales, avs, rvs = scales[sidxs], avs[sidxs], rvs[sidxs]
        cov_sar = cov_sar[idxs]
        lnprob = lnprob[idxs]
        if not return_distreds:
            return (sidxs, scales, avs, rvs, cov_sar, Ndim, lnprob, levid,
                    chi2min)
        imc = np.zeros(Ndraws, dtype='int')                  # :2049-2053
        for i, idx in enumerate(idxs):
            w = np.exp(logwts[idx] - logsumexp(logwts[idx]))
            w /= w.sum()
            imc[i] = rstate.choice(Nmc_prior, p=w)
        dists = dists[idxs, imc]
        reds = reds[idxs, imc]
        dreds = dreds[idxs, imc]
        logwts = logwts[idxs, imc]
    return (sidxs, scales, avs, rvs, cov_sar, Ndim, lnprob, levid, chi2min,
            dists, reds, dreds, logwts)


# ---------------------------------------------------------------------------
# cluster.py
# ---------------------------------------------------------------------------
def isochrone_loglike(theta, isochrone, phot, err, cluster_params='free',
                      offsets='fixed', corr_params='fixed', mini_bound=0.08,
                      eep_binary_max=480., smf_grid=None, eep_grid=None,
                      parallax=None, parallax_err=None, cluster_prob=0.95,
                      dim_prior=True, return_lnls=False):
    """cluster.py:170-419 (`isochrone_loglike`), numpy restatement that keeps
    the reference's (Ncmd, Nobj, Nb) broadcast and scipy calls."""
    from scipy.stats import chi2 as chisquare
    Nobjs, Nbands = phot.shape
    phot_mask = np.isfinite(phot) & np.isfinite(err)            # :179
    phot_n = np.sum(phot_mask, axis=1)
    if smf_grid is None:                                        # :184-187
        smf_grid = np.array([0., 0.2, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65,
                             0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 1.0])
    grad_smf = np.gradient(smf_grid) if len(smf_grid) > 1 else np.array([1.])
    if eep_grid is None:
        eep_grid = np.linspace(202., 808., 2000)                # :198

    def take(pos, spec, n):
        out = np.zeros(n)
        for i in range(n):
            if (isinstance(spec, str) and spec == 'free') or spec[i] is None:
                out[i] = theta[pos]
                pos += 1
            else:
                out[i] = spec[i]
        return out, pos

    pos = 0
    (feh, loga, av, rv, dist, fout), pos = take(pos, cluster_params, 6)  # :231-248
    fout = max(min(1. - 1e-10, fout), 1e-10)
    if isinstance(offsets, str) and offsets == 'fixed':         # :251-270
        Xb = np.ones(Nbands)
        pos += Nbands
    else:
        Xb, pos = take(pos, offsets, Nbands)
    if isinstance(corr_params, str) and corr_params == 'fixed':  # :273-290
        corr_coef = None
    else:
        corr_coef, pos = take(pos, corr_params, 4)

    chi2_p = np.zeros(Nobjs)                                    # :292-301
    lnorm_p = np.zeros(Nobjs)
    if parallax is not None and parallax_err is not None:
        pm = np.isfinite(parallax) & np.isfinite(parallax_err)
        chi2_p[pm] += (parallax[pm] - 1e3 / dist) ** 2 / parallax_err[pm] ** 2
        lnorm_p[pm] += np.log(2. * np.pi * parallax_err[pm] ** 2)
        phot_n = phot_n + pm
    with np.errstate(all="ignore"):
        if dim_prior:                                           # :304-307
            lnl_outlier = chisquare.logpdf(chisquare.ppf(1. - 1e-5, phot_n), phot_n)
        else:                                                   # :308-321
            omax = np.nanmax(phot + 3. * err, axis=0)
            omin = np.nanmin(phot - 3. * err, axis=0)
            osize = (6. * err) / (omax - omin)
            osize[~phot_mask] = 1.
            ovol = np.prod(osize * phot_mask + 1. * ~phot_mask, axis=1)
            if parallax is not None and parallax_err is not None:
                p_max = np.nanmax((parallax + 3. * parallax_err)[pm])
                p_min = np.nanmin((parallax - 3. * parallax_err)[pm])
                ovol[pm] *= (6. * parallax_err[pm]) / (p_max - p_min)
            lnl_outlier = np.log(1. / ovol)
        ln_fin = np.log(cluster_prob * (1. - fout))             # :324-325
        ln_fout = np.log(1. - cluster_prob * (1. - fout))
        lnls = np.full((len(smf_grid), Nobjs), -np.inf)
        done_first = False
        for i, smf in enumerate(smf_grid):                      # :336-404
            cmd_sed, params1, _ = isochrone.get_seds(
                feh=feh, loga=loga, av=av, rv=rv, eep=eep_grid, smf=smf,
                dist=dist, mini_bound=mini_bound,
                eep_binary_max=eep_binary_max, corr_params=corr_coef)
            cmd_mini = params1['mini']
            grad_mini = np.gradient(cmd_mini)
            ok = np.any(np.isfinite(cmd_sed), axis=1) & (grad_mini > 0.)
            if done_first:
                ok &= eep_grid <= eep_binary_max
            done_first = True
            cmd_mask = np.where(ok)[0]
            if len(cmd_mask) == 0:
                continue
            cmd_sed, grad_mini = cmd_sed[cmd_mask], grad_mini[cmd_mask]
            phot_t, err_t = phot * Xb, err * Xb
            cmd_phot = 10 ** (-0.4 * cmd_sed)
            chi2_cmd = np.nansum((phot_t - cmd_phot[:, None]) ** 2 / err_t ** 2,
                                 axis=-1)
            lnorm_cmd = np.nansum(np.log(2. * np.pi * err_t ** 2), axis=-1)
            chi2 = chi2_cmd + chi2_p
            lnorm = lnorm_cmd + lnorm_p
            if dim_prior:
                lnl_cmd = chisquare.logpdf(chi2, phot_n)
            else:
                lnl_cmd = -0.5 * (chi2 + lnorm)
            lnl_cmd[~np.isfinite(lnl_cmd)] = -np.inf
            lnprior = np.log(grad_mini) + np.log(grad_smf[i])
            lnls[i] = logsumexp(lnl_cmd + lnprior[:, None], axis=0)
        lnl = logsumexp(lnls, axis=0)                           # :407
        lnl_mix = np.logaddexp(lnl + ln_fin, lnl_outlier + ln_fout)  # :410-411
    lnl_tot = np.sum(lnl_mix)
    if return_lnls:
        return lnl_tot, lnl_mix
    return lnl_tot
