"""Co-eval population ("cluster") log-likelihood, MI355X-native.

Host-side mirror of reference `brutus/cluster.py:isochrone_loglike`
(cluster.py:23-419).  The population model (`isochrone.get_seds`, the MIST/NN
isochrone generator of reference `seds.py`) stays a host-side, duck-typed
plug-in exactly as in the reference (cluster.py:339-344); the hot block -- the
chi2 of every object against every isochrone point of every
secondary-mass-fraction slice, the chi-square/normal log-pdf and the
marginalisation over mass and mass fraction (cluster.py:379-407) -- runs in the
HIP kernel `k_cluster` through `brutus_cluster_lnl`.
"""
import warnings

import numpy as np

from . import _lib

__all__ = ["isochrone_loglike"]

_DEFAULT_SMF = (0., 0.2, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8,
                0.85, 0.9, 0.95, 1.0)


def _take(theta, pos, spec, n):
    """Read `n` values: from `theta` where `spec[i] is None`, else the fixed
    value (the reference's 'free' / per-entry constraint convention)."""
    out = np.zeros(n)
    for i in range(n):
        if spec == 'free' or spec[i] is None:
            out[i] = theta[pos]
            pos += 1
        else:
            out[i] = spec[i]
    return out, pos


def isochrone_loglike(theta, isochrone, phot, err, cluster_params='free',
                      offsets='fixed', corr_params='fixed', mini_bound=0.08,
                      eep_binary_max=480., smf_grid=None, eep_grid=None,
                      parallax=None, parallax_err=None, cluster_prob=0.95,
                      dim_prior=True, return_lnls=False, device=None):
    """ln-likelihood of a co-eval stellar population.  Arguments, defaults and
    return value follow reference cluster.py:23-168: `theta` packs
    `(feh, loga, av, rv, dist[pc], fout)`, then per-band multiplicative offsets
    and four empirical-correction coefficients, each group only where it is
    declared free.  `isochrone` must provide
    `get_seds(feh=, loga=, av=, rv=, eep=, smf=, dist=, mini_bound=,
    eep_binary_max=, corr_params=) -> (seds (Neep, Nbands) mags, params, params2)`
    with `params['mini']` the initial-mass grid."""
    from .fitting import _torch, _stream_ptr
    from scipy.stats import chi2 as chisquare
    if phot is None:
        raise ValueError("The photometry must be provided to compute the "
                         "log-likelihood!")
    if err is None:
        raise ValueError("The errors on the photometry must be provided to "
                         "compute the log-likelihood!")
    phot = np.asarray(phot, dtype=np.float64)
    err = np.asarray(err, dtype=np.float64)
    Nobjs, Nbands = phot.shape
    phot_mask = np.isfinite(phot) & np.isfinite(err)
    phot_n = np.sum(phot_mask, axis=1)
    if np.any(phot_n == 0):
        raise ValueError("At least one object has no valid data entries!")
    smf_grid = np.asarray(_DEFAULT_SMF if smf_grid is None else smf_grid, float)
    grad_smf = np.gradient(smf_grid) if len(smf_grid) > 1 else np.array([1.])
    if eep_grid is None:
        eep_grid = np.linspace(202., 808., 2000)
    eep_grid = np.asarray(eep_grid, dtype=np.float64)
    free_str = lambda v: isinstance(v, str) and v == 'free'
    if parallax is None and parallax_err is None:          # cluster.py:200-208
        if free_str(offsets) and (free_str(cluster_params)
                                  or cluster_params[4] is None):
            raise ValueError("Without any measured parallaxes, there is a "
                             "degeneracy between the photometry offsets "
                             "and the distance. Please provide either a "
                             "distance value in `cluster_params` or at "
                             "least one offset in `offsets`.")
    if not (isinstance(corr_params, str) and corr_params == 'fixed'):
        if ((corr_params[0] is None or corr_params[1] is None)
                and corr_params[3] is None):               # cluster.py:212-217
            raise ValueError("If `feh_scale` is not provided, then `dtdm` and "
                             "`drdm` must be fixed since the parameters are "
                             "perfectly degenerate.")
    if parallax is None and parallax_err is not None:
        raise ValueError("You forgot to provide the parallaxes to go along "
                         "with the errors!")
    if parallax is not None and parallax_err is None:
        raise ValueError("You forgot to provide the parallax errors to go "
                         "along with the parallaxes!")

    # ---- unpack theta (cluster.py:227-290) ------------------------------------
    pos = 0
    (feh, loga, av, rv, dist, fout), pos = _take(theta, pos, cluster_params, 6)
    fout = max(min(1. - 1e-10, fout), 1e-10)
    if isinstance(offsets, str) and offsets == 'fixed':
        Xb = np.ones(Nbands)
        pos += Nbands
    else:
        Xb, pos = _take(theta, pos, offsets, Nbands)
    if isinstance(corr_params, str) and corr_params == 'fixed':
        corr_coef = None
        pos += 4
    else:
        corr_coef, pos = _take(theta, pos, corr_params, 4)

    # ---- per-object terms (cluster.py:292-325) ----------------------------------
    chi2_p = np.zeros(Nobjs)
    lnorm_p = np.zeros(Nobjs)
    pmask = np.zeros(Nobjs, dtype=bool)
    if parallax is not None and parallax_err is not None:
        parallax = np.asarray(parallax, dtype=np.float64)
        parallax_err = np.asarray(parallax_err, dtype=np.float64)
        pmask = np.isfinite(parallax) & np.isfinite(parallax_err)
        chi2_p[pmask] = (parallax[pmask] - 1e3 / dist) ** 2 / parallax_err[pmask] ** 2
        lnorm_p[pmask] = np.log(2. * np.pi * parallax_err[pmask] ** 2)
        phot_n = phot_n + pmask
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        if dim_prior:
            lnl_outlier = chisquare.logpdf(chisquare.ppf(1. - 1e-5, phot_n), phot_n)
        else:
            side = np.nanmax(phot + 3. * err, axis=0) - np.nanmin(phot - 3. * err, axis=0)
            frac = np.where(phot_mask, 6. * err / side, 1.)
            vol = np.prod(frac, axis=1)
            if parallax is not None and parallax_err is not None:
                span = (np.nanmax((parallax + 3. * parallax_err)[pmask])
                        - np.nanmin((parallax - 3. * parallax_err)[pmask]))
                vol[pmask] *= 6. * parallax_err[pmask] / span
            lnl_outlier = np.log(1. / vol)
    ln_fin = np.log(cluster_prob * (1. - fout))
    ln_fout = np.log(1. - cluster_prob * (1. - fout))

    # ---- isochrone points of every SMF slice (cluster.py:336-366) -----------------
    flux_parts, lnw_parts = [], []
    first = True
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for i, smf in enumerate(smf_grid):
            seds, params, _ = isochrone.get_seds(
                feh=feh, loga=loga, av=av, rv=rv, eep=eep_grid, smf=smf,
                dist=dist, mini_bound=mini_bound, eep_binary_max=eep_binary_max,
                corr_params=corr_coef)
            mini = np.asarray(params['mini'], dtype=np.float64)
            gmini = np.gradient(mini)
            keep = np.any(np.isfinite(seds), axis=1) & (gmini > 0.)
            if not first:   # evolved-star models do not depend on the SMF
                keep &= eep_grid <= eep_binary_max
            first = False
            if np.any(keep):
                flux_parts.append(10. ** (-0.4 * np.asarray(seds, float)[keep]))
                lnw_parts.append(np.log(gmini[keep]) + np.log(grad_smf[i]))
    if not flux_parts:
        lnl = np.full(Nobjs, -np.inf)
    else:
        pts_flux = np.ascontiguousarray(np.concatenate(flux_parts))
        pts_lnw = np.ascontiguousarray(np.concatenate(lnw_parts))
        phot_t, err_t = phot * Xb, err * Xb
        with np.errstate(all="ignore"):
            ivar = np.where(phot_mask, 1. / err_t ** 2, 0.)
            lnorm = np.nansum(np.log(2. * np.pi * err_t ** 2), axis=1) + lnorm_p
        d = np.where(phot_mask, phot_t, 0.)
        torch = _torch()
        L = _lib.lib()
        dev = torch.device(device if device is not None
                           else "cuda:%d" % torch.cuda.current_device())
        with torch.cuda.device(dev):
            up = lambda a, dt=np.float64: torch.from_numpy(
                np.ascontiguousarray(a, dtype=dt)).to(dev)
            t_flux, t_lnw, t_d, t_iv = up(pts_flux), up(pts_lnw), up(d), up(ivar)
            t_cp, t_ln, t_n = up(chi2_p), up(lnorm), up(phot_n, np.int32)
            ws = torch.empty(L.brutus_cluster_workspace_bytes(Nobjs),
                             dtype=torch.uint8, device=dev)
            out = torch.empty(Nobjs, dtype=torch.float64, device=dev)
            _lib.check(L.brutus_cluster_lnl(
                Nobjs, Nbands, pts_flux.shape[0], t_flux.data_ptr(),
                t_lnw.data_ptr(), t_d.data_ptr(), t_iv.data_ptr(),
                t_cp.data_ptr(), t_ln.data_ptr(), t_n.data_ptr(),
                1 if dim_prior else 0, ws.data_ptr(), ws.numel(), out.data_ptr(),
                _stream_ptr(torch)))
            lnl = out.cpu().numpy()
    # ---- outlier mixture (cluster.py:410-414) ---------------------------------------
    with np.errstate(all="ignore"):
        lnl_mix = np.logaddexp(lnl + ln_fin, lnl_outlier + ln_fout)
    lnl_tot = np.sum(lnl_mix)
    if return_lnls:
        return lnl_tot, lnl_mix
    return lnl_tot
