"""Host-side helpers on the fit() path (numpy, float64).

Counterparts of the small utilities the reference keeps in `brutus/utils.py`:
`_inverse3` (utils.py:71-114), `_chisquare_logpdf` (utils.py:130-176),
`sample_multivariate_normal` (utils.py:845-905), `magnitude` (utils.py:908-940).
They act on the few hundred-to-thousand models that survive the device-side
cuts, never on the full grid.
"""
from math import lgamma, log

import numpy as np

__all__ = ["_function_wrapper", "_adjoint3", "_inverse_transpose3", "_inverse3", "_dot3",
           "_isPSD", "_chisquare_logpdf", "_truncnorm_pdf", "_truncnorm_logpdf", "_get_seds",
           "fetch_isos", "fetch_tracks", "fetch_dustmaps", "fetch_grids", "fetch_offsets",
           "fetch_nns", "load_models", "load_offsets", "quantile", "draw_sar",
           "sample_multivariate_normal", "magnitude", "inv_magnitude", "luptitude",
           "inv_luptitude", "add_mag", "get_seds", "phot_loglike", "photometric_offsets"]


def _inverse3(A):
    """Inverse of a stack of 3x3 matrices (..., 3, 3) by the adjugate.

    Rows of the adjugate-transpose are cross products of the other two rows;
    the determinant is taken as the mean of the three row.cofactor-row dots
    (same estimator as reference utils.py:96-105, so results agree to rounding).
    """
    A = np.asarray(A, dtype=np.float64)
    r0, r1, r2 = A[..., 0, :], A[..., 1, :], A[..., 2, :]
    c0 = np.cross(r1, r2)
    c1 = np.cross(r2, r0)
    c2 = np.cross(r0, r1)
    det = ((c0 * r0).sum(-1) + (c1 * r1).sum(-1) + (c2 * r2).sum(-1)) / 3.
    with np.errstate(all="ignore"):
        # inverse = cofactor^T / det: cofactor rows become columns
        return np.stack([c0, c1, c2], axis=-1) / det[..., None, None]


def _chisquare_logpdf(x, df, loc=0, scale=1):
    """ln pdf of a chi-square variate with `df` degrees of freedom; -inf for
    arguments <= 0 (reference utils.py:161-176)."""
    y = (np.asarray(x, dtype=np.float64) - loc) / scale
    half = df / 2.
    norm = half * log(2.) + lgamma(half)
    with np.errstate(all="ignore"):
        out = (half - 1.) * np.log(np.where(y > 0, y, 1.)) - y / 2. - norm - log(scale)
    out = np.where(y > 0, out, -np.inf)
    return float(out) if out.ndim == 0 else out


def sample_multivariate_normal(mean, cov, size=1, eps=1e-30, rstate=None):
    """Draw `size` samples from each of N trivariate normals.

    Returns (dim, size, N) like reference utils.py:845-905.  The standard
    normals are consumed from `rstate` in the reference's order
    (`normal(size=dim*size*N).reshape(N, dim, size)`), which is what makes
    seeded runs reproduce the reference draw for draw.
    """
    if rstate is None:
        rstate = np.random
    mean = np.asarray(mean, dtype=np.float64)
    if mean.ndim == 1:
        return rstate.multivariate_normal(mean, cov, size=size)
    N, d = mean.shape
    chol = np.linalg.cholesky(cov + eps * np.eye(d)[None, :, :])
    z = rstate.normal(loc=0, scale=1, size=d * size * N).reshape(N, d, size)
    draws = mean[:, :, None] + np.matmul(chol, z)        # (N, d, size)
    return np.transpose(draws, (1, 2, 0))


def magnitude(phot, err, zeropoints=1.):
    """Flux densities -> AB magnitudes and errors (reference utils.py:908-940)."""
    phot = np.asarray(phot, dtype=np.float64)
    with np.errstate(all="ignore"):
        mag = -2.5 * np.log10(phot / zeropoints)
        mag_err = 2.5 / np.log(10.) * np.asarray(err) / phot
    return mag, mag_err


def inv_magnitude(mag, err, zeropoints=1.):
    """AB magnitudes -> flux densities (reference utils.py:943-975)."""
    phot = 10. ** (-0.4 * np.asarray(mag, dtype=np.float64)) * zeropoints
    phot_err = np.asarray(err) * 0.4 * np.log(10.) * phot
    return phot, phot_err


# ---------------------------------------------------------------------------
# grid / offsets loaders (reference utils.py:520-715)
# ---------------------------------------------------------------------------
def load_models(filepath, filters=None, labels=None, include_ms=True,
                include_postms=True, include_binaries=False, verbose=True):
    """Read a model-grid HDF5 file into `(models, labels, label_mask)`.

    Same contract as reference `utils.load_models` (utils.py:520-662):
    `models` is `(Nmodel, Nfilt, 3)` float32 magnitude coefficients for the
    requested `filters` (all-zero / absent filters dropped), `labels` a
    structured float array of the requested label fields that exist in the
    file (`labels` dataset = grid inputs, `parameters` dataset = predictions),
    `label_mask` a structured `(1,)` bool array flagging the grid inputs.
    Main-sequence / post-main-sequence (EEP 454 split) and binary (`smf`)
    cuts as in the reference.  Reads through libhdf5 (`brutus_amd.h5io`).
    """
    import sys
    from . import h5io
    from .filters import FILTERS
    if filters is None:
        filters = FILTERS
    if labels is None:
        labels = ['mini', 'feh', 'eep', 'smf', 'loga', 'logl', 'logt', 'logg',
                  'Mr', 'agewt']
    if not include_ms and not include_postms:
        raise ValueError("If you don't include the Main Sequence and "
                         "Post-Main Sequence models you have nothing left!")
    coeffs = h5io.read_dataset(filepath, "mag_coeffs")
    have = coeffs.dtype.names or ()
    ncoef = coeffs.dtype[have[0]].shape[0] if have else 3
    models = np.zeros((len(coeffs), len(filters), ncoef), dtype='float32')
    for j, filt in enumerate(filters):
        if filt in have:
            models[:, j] = coeffs[filt]
            if verbose:
                sys.stderr.write('\rReading filter {}           '.format(filt))
                sys.stderr.flush()
    if verbose:
        sys.stderr.write('\n')
    models = models[:, ~np.all(models == 0., axis=(0, 2)), :]

    present = set(h5io.list_datasets(filepath))
    combined = np.full(len(models), np.nan,
                       dtype=np.dtype([(n, np.float64) for n in labels]))
    label_mask = np.zeros(1, dtype=np.dtype([(n, np.bool_) for n in labels]))
    if "labels" in present:
        tab = h5io.read_dataset(filepath, "labels")
        for n in tab.dtype.names:
            if n in labels:
                combined[n] = tab[n]
                label_mask[n] = True
    if "parameters" in present:
        tab = h5io.read_dataset(filepath, "parameters")
        for n in tab.dtype.names:
            if n in labels:
                combined[n] = tab[n]
    kept = [n for n in labels if not np.isnan(combined[n][0])]

    sel = np.ones(len(combined), dtype=bool)
    if 'eep' in kept and not (include_ms and include_postms):
        sel = combined['eep'] > 454. if include_postms else combined['eep'] <= 454.
    if not include_binaries and 'smf' in kept:
        sel = sel & (combined['smf'] == 0.)
        kept = [n for n in kept if n != 'smf']
    return models[sel], combined[kept][sel], label_mask[kept]


def load_offsets(filepath, filters=None, verbose=True):
    """Multiplicative photometric offsets per filter from a two-column text
    file `(filter name, value)`; filters without an entry get 1
    (reference utils.py:665-715)."""
    import sys
    from .filters import FILTERS
    if filters is None:
        filters = FILTERS
    names, vals = np.loadtxt(filepath, dtype='str').T
    vals = vals.astype(float)
    offsets = np.ones(len(filters))
    for j, filt in enumerate(filters):
        hit = np.where(names == filt)[0]
        if len(hit) > 1:
            raise ValueError("Something went wrong when extracting "
                             "offsets for filter {}.".format(filt))
        if len(hit) == 1:
            offsets[j] = vals[hit[0]]
    if verbose:
        for filt, zp in zip(filters, offsets):
            sys.stderr.write('{0} ({1:3.2}%)\n'.format(filt, 100 * (zp - 1.)))
    return offsets


# ---------------------------------------------------------------------------
# consumers of the fit output: host-side, they act on the
# Ndraws resampled models of one object, not on the grid
# ---------------------------------------------------------------------------
def get_seds(mag_coeffs, av=None, rv=None, return_flux=False,
             return_rvec=False, return_drvec=False):
    """Reddened SEDs `mag + Av (R0 + Rv dR/dRv)` (optionally as flux densities)
    from `(Nmodel, Nband, 3)` magnitude coefficients; same call and returns
    as reference `utils.get_seds` (utils.py:1089-1159, kernel utils.py:286-347).
    Float32 coefficients are promoted to float64 first, like numba does."""
    c = np.asarray(mag_coeffs, dtype=np.float64)
    n = c.shape[0]
    av = np.zeros(n) if av is None else np.broadcast_to(np.asarray(av, float), (n,))
    rv = np.full(n, 3.3) if rv is None else np.broadcast_to(np.asarray(rv, float), (n,))
    drvecs = c[:, :, 2].copy()
    rvecs = c[:, :, 1] + rv[:, None] * drvecs
    seds = c[:, :, 0] + av[:, None] * rvecs
    if return_flux:
        seds = 10. ** (-0.4 * seds)
        scale = (-0.4 * log(10.)) * seds
        rvecs = rvecs * scale
        drvecs = drvecs * scale
    out = (seds,)
    if return_rvec:
        out += (rvecs,)
    if return_drvec:
        out += (drvecs,)
    return out if len(out) > 1 else seds


def draw_sar(scales, avs, rvs, covs_sar, ndraws=500, avlim=(0., 6.),
             rvlim=(1., 8.), rstate=None):
    """`ndraws` in-bounds draws of (scale, Av, Rv) around each resampled model,
    consuming `rstate.multivariate_normal` like reference utils.py:765-842
    (redraw batches of `ndraws` until enough fall inside the bounds)."""
    if rstate is None:
        rstate = getattr(np, "random_intel", np.random)
    n = len(scales)
    out = np.zeros((3, n, ndraws))
    for i in range(n):
        kept = [np.empty((3, 0))]
        have = 0
        while have < ndraws:
            d = rstate.multivariate_normal([scales[i], avs[i], rvs[i]],
                                           covs_sar[i], size=ndraws).T
            ok = ((d[0] >= 0.) & (d[1] >= avlim[0]) & (d[1] <= avlim[1])
                  & (d[2] >= rvlim[0]) & (d[2] <= rvlim[1]))
            kept.append(d[:, ok])
            have += int(ok.sum())
        out[:, i, :] = np.concatenate(kept, axis=1)[:, :ndraws]
    return out[0], out[1], out[2]


def phot_loglike(data, data_err, data_mask, models, dim_prior=True):
    """ln-likelihood of noisy fluxes `data` against noiseless model fluxes
    `models (Nmodel, Nfilt)` (reference utils.py:1162-1215): Gaussian, or the
    chi-square-distribution form with `Ndim - 3` degrees of freedom."""
    from scipy.special import gammaln, xlogy
    mask = np.asarray(data_mask, dtype=bool)
    flux, var = np.asarray(data)[mask], np.square(np.asarray(data_err)[mask])
    ndim = int(mask.sum())
    resid = flux - np.asarray(models)[:, mask]
    chi2 = np.sum(np.square(resid) / var, axis=1)
    if dim_prior:
        a = 0.5 * (ndim - 3)
        return xlogy(a - 1., chi2) - chi2 / 2. - gammaln(a) - np.log(2.) * a
    return -0.5 * chi2 - 0.5 * (ndim * np.log(2. * np.pi) + np.sum(np.log(var)))


def _phot_loglike_many(data, data_err, band_mask, models, dim_prior):
    """`phot_loglike` for many objects sharing one band mask: `data`, `data_err` (G, Nfilt),
    `models` (G, Nmodel, Nfilt) -> (G, Nmodel); row g equals
    `phot_loglike(data[g], data_err[g], band_mask, models[g])` bit for bit."""
    from scipy.special import gammaln, xlogy
    m = np.asarray(band_mask, dtype=bool)
    flux, var = data[:, m], np.square(data_err[:, m])
    ndim = int(m.sum())
    resid = flux[:, None, :] - models[:, :, m]
    chi2 = np.sum(np.square(resid) / var[:, None, :], axis=2)
    if dim_prior:
        a = 0.5 * (ndim - 3)
        return xlogy(a - 1., chi2) - chi2 / 2. - gammaln(a) - np.log(2.) * a
    return -0.5 * chi2 - 0.5 * (ndim * np.log(2. * np.pi) + np.sum(np.log(var), axis=1)[:, None])


def photometric_offsets(phot, err, mask, models, idxs, reds, dreds, dists,
                        sel=None, weights=None, mask_fit=None, Nmc=150,
                        old_offsets=None, dim_prior=True, prior_mean=None,
                        prior_std=None, verbose=True, rstate=None, device=None):
    """Multiplicative photometric offsets (model / data) per band from the
    resampled fits of many objects, with bootstrap errors; same arguments,
    RNG call order and returns `(ratios, ratios_err, nratio)` as reference
    `utils.photometric_offsets` (utils.py:1218-1400).

    For a band that took part in the fit the model draws of every object are
    re-weighted by the likelihood of the *other* bands (leave-one-band-out),
    so that the band does not calibrate itself.

    `device` (not in the reference): a torch device (`"cuda"`, `"cuda:0"`) runs
    the SEDs, the leave-one-out weights and the bootstrap rounds in the HIP
    library (`brutus_offsets_weights`, `brutus_offsets_bootstrap`); `rstate` is
    consumed exactly as on the host (same uniforms, same `searchsorted` rule).  The
    cumulative weights a uniform is looked up in are built differently, though: a
    block-parallel scan of `exp(lnl - max)` on the device, a sequential `cumsum` of
    `exp(lnl - logsumexp)` on the host, equal to a few ulp -- so a draw can differ where a
    uniform falls within rounding of a cdf step (statistically equivalent; measured
    agreement of the resulting offsets: 1e-11)."""
    import sys
    if device is not None:
        return _photometric_offsets_device(
            phot, err, mask, models, idxs, reds, dreds, dists, sel, weights, mask_fit, Nmc,
            old_offsets, dim_prior, prior_mean, prior_std, verbose, rstate, device)
    from scipy.special import logsumexp
    phot, err = np.asarray(phot, float), np.asarray(err, float)
    mask = np.asarray(mask, dtype=bool)
    Nobj, Nfilt = phot.shape
    Nsamps = idxs.shape[1]
    sel = np.ones(Nobj, dtype=bool) if sel is None else np.asarray(sel, bool)
    weights = np.ones((Nobj, Nsamps)) if weights is None else np.asarray(weights, float)
    mask_fit = np.ones(Nfilt, dtype=bool) if mask_fit is None else np.asarray(mask_fit, bool)
    old_offsets = np.ones(Nfilt) if old_offsets is None else np.asarray(old_offsets, float)
    if rstate is None:
        rstate = getattr(np, "random_intel", np.random)

    seds = get_seds(models[idxs.ravel()], av=reds.ravel(), rv=dreds.ravel(),
                    return_flux=True)
    seds = (seds / dists.ravel()[:, None] ** 2).reshape(Nobj, Nsamps, Nfilt)

    ratios, ratios_err = np.ones(Nfilt), np.zeros(Nfilt)
    nratio = np.zeros(Nfilt, dtype=int)
    nbands = mask.sum(axis=1)
    usable = sel & (weights.sum(axis=1) > 0)
    for b in range(Nfilt):
        need = 3 + (1 if mask_fit[b] else 0)      # bands besides this one
        s = np.where(mask[:, b] & usable & (nbands > need))[0]
        n = nratio[b] = len(s)
        if n == 0:
            continue
        ratio = seds[s, :, b] / phot[s, None, b]
        if mask_fit[b]:
            others = mask[s].copy()
            others[:, b] = False
            # the reference evaluates phot_loglike object by object; objects with the same
            # band pattern are evaluated together here (same sums in the same order)
            lnl = np.empty((n, Nsamps))
            pat, inv = np.unique(others, axis=0, return_inverse=True)
            inv = np.asarray(inv).reshape(-1)
            for k, m in enumerate(pat):
                g = np.where(inv == k)[0]
                lnl[g] = _phot_loglike_many(phot[s[g]] * old_offsets, err[s[g]] * old_offsets,
                                            m, seds[s[g]], dim_prior)
            wt = np.exp(lnl - logsumexp(lnl, axis=1)[:, None])
        else:
            wt = np.ones((n, Nsamps))
        wt = wt * weights[s]
        wt /= wt.sum(axis=1)[:, None]
        wt_obj = np.array(weights[s].sum(axis=1) > 0, dtype=float)
        wt_obj /= wt_obj.sum()
        # The reference draws `ridx = choice(n, size=n, p=wt_obj)` and then one
        # `choice(Nsamps, p=wt[i])` per drawn object: 2 n `random_sample` values in that
        # order, each turned into an index by `searchsorted(cumsum(p) / cumsum(p)[-1], u,
        # 'right')` (numpy's legacy `choice`).  Same stream, same indices, without the n
        # Python-level calls per bootstrap round.
        cdf_obj = wt_obj.cumsum()
        cdf_obj /= cdf_obj[-1]
        cdf = wt.cumsum(axis=1)
        cdf /= cdf[:, -1:]
        sample = getattr(rstate, "random_sample", None) or rstate.random
        meds = np.empty(Nmc)
        for j in range(Nmc):
            if verbose:
                sys.stderr.write('\rBand {0} ({1}/{2})     '.format(b + 1, j + 1, Nmc))
                sys.stderr.flush()
            ridx = cdf_obj.searchsorted(sample(n), side='right')
            u = sample(n)
            midx = np.empty(n, dtype=np.intp)
            for lo in range(0, n, 4096):              # bounded temporaries
                hi = min(n, lo + 4096)
                midx[lo:hi] = np.sum(cdf[ridx[lo:hi]] <= u[lo:hi, None], axis=1)
            meds[j] = np.median(ratio[ridx, midx])
        ratios[b], ratios_err[b] = np.median(meds), np.std(meds)
    if verbose:
        sys.stderr.write('\n')
    if prior_mean is not None and prior_std is not None:
        var = ratios_err ** 2 + prior_std ** 2
        ratios = (ratios * prior_std ** 2 + prior_mean * ratios_err ** 2) / var
        ratios_err = ratios_err * prior_std / np.sqrt(var)
    return ratios, ratios_err, nratio


def _offsets_subsets(mask, sel, weights, mask_fit):
    """Objects that enter each band (reference utils.py:1337-1350)."""
    nbands = mask.sum(axis=1)
    usable = sel & (weights.sum(axis=1) > 0)
    return [np.where(mask[:, b] & usable & (nbands > 3 + (1 if mask_fit[b] else 0)))[0]
            for b in range(mask.shape[1])]


def _photometric_offsets_device(phot, err, mask, models, idxs, reds, dreds, dists, sel, weights,
                                mask_fit, Nmc, old_offsets, dim_prior, prior_mean, prior_std,
                                verbose, rstate, device):
    """`photometric_offsets` with the arithmetic on the GPU (csrc/offsets_kernels.hpp).
    The host draws the uniforms (2 n per bootstrap round, the order of the reference's
    `choice` calls, utils.py:1381-1385) and takes median / std of the Nmc medians."""
    import sys
    from . import _lib
    from .fitting import _torch, _stream_ptr
    phot, err = np.asarray(phot, float), np.asarray(err, float)
    mask = np.asarray(mask, dtype=bool)
    Nobj, Nfilt = phot.shape
    idxs = np.asarray(idxs)
    Nsamps = idxs.shape[1]
    sel = np.ones(Nobj, dtype=bool) if sel is None else np.asarray(sel, bool)
    weights = np.ones((Nobj, Nsamps)) if weights is None else np.asarray(weights, float)
    mask_fit = np.ones(Nfilt, dtype=bool) if mask_fit is None else np.asarray(mask_fit, bool)
    old_offsets = np.ones(Nfilt) if old_offsets is None else np.asarray(old_offsets, float)
    if rstate is None:
        rstate = getattr(np, "random_intel", np.random)
    models = np.asarray(models)
    if models.ndim != 3 or models.shape[1] != Nfilt or models.shape[2] != 3:
        raise ValueError("models must have shape (Nmodel, Nfilt, 3)")
    idxs = np.asarray(idxs)
    if idxs.size and (idxs.min() < -models.shape[0] or idxs.max() >= models.shape[0]):
        # (numpy's fancy indexing raises on the host path; the kernel would clamp)
        raise IndexError("model index out of range for a grid of %d models" % models.shape[0])
    if models.dtype != np.float32:
        # the kernels read the float32 coefficients of `load_models` (utils.py:588-591)
        m32 = models.astype(np.float32)
        if not np.array_equal(m32.astype(models.dtype), models):
            raise ValueError("the device path needs float32 magnitude coefficients "
                             "(as `load_models` returns them)")
        models = m32
    torch = _torch()
    L = _lib.lib()
    dev = torch.device("cuda:%d" % torch.cuda.current_device() if device in (True, "cuda")
                       else device)

    def up(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)

    subsets = _offsets_subsets(mask, sel, weights, mask_fit)
    use = np.zeros((Nfilt, Nobj), dtype=np.uint8)
    for b, s in enumerate(subsets):
        use[b, s] = 1
    with torch.cuda.device(dev):
        sp = _stream_ptr(torch)
        t_models = up(models, np.float32)
        t_phot, t_err = up(phot, np.float64), up(err, np.float64)
        t_w = up(weights, np.float64)
        t_flux = torch.empty((Nfilt, Nobj, Nsamps), dtype=torch.float64, device=dev)
        t_cdf = torch.empty((Nfilt, Nobj, Nsamps), dtype=torch.float64, device=dev)
        keep = [up(idxs, np.int64), up(reds, np.float64), up(dreds, np.float64),
                up(dists, np.float64), up(mask, np.uint8), up(old_offsets, np.float64),
                up(use, np.uint8), up(mask_fit, np.uint8)]
        _lib.check(L.brutus_offsets_weights(
            Nobj, Nsamps, Nfilt, models.shape[0], t_models.data_ptr(), keep[0].data_ptr(),
            keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), t_phot.data_ptr(),
            t_err.data_ptr(), keep[4].data_ptr(), t_w.data_ptr(), keep[5].data_ptr(),
            keep[6].data_ptr(), keep[7].data_ptr(), int(bool(dim_prior)), t_flux.data_ptr(),
            t_cdf.data_ptr(), sp))
        ratios, ratios_err = np.ones(Nfilt), np.zeros(Nfilt)
        nratio = np.zeros(Nfilt, dtype=int)
        sample = getattr(rstate, "random_sample", None) or rstate.random
        nmax = max(len(s) for s in subsets)
        t_ws = None
        if nmax:
            nbytes = int(L.brutus_offsets_workspace_bytes(nmax, Nmc))
            if nbytes == 0:
                raise ValueError("Nmc x objects per band is too large for one bootstrap call")
            t_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            t_meds = torch.empty(Nmc, dtype=torch.float64, device=dev)
        for b, s in enumerate(subsets):
            n = nratio[b] = len(s)
            if n == 0:
                continue
            if verbose:
                sys.stderr.write('\rBand {0} ({1}/{1})     '.format(b + 1, Nmc))
                sys.stderr.flush()
            wt_obj = np.array(weights[s].sum(axis=1) > 0, dtype=float)
            wt_obj /= wt_obj.sum()
            cdf_obj = wt_obj.cumsum()
            cdf_obj /= cdf_obj[-1]
            u = sample(2 * n * Nmc)         # round j: n object draws, then n model draws
            t_u, t_s, t_co = up(u, np.float64), up(s, np.int32), up(cdf_obj, np.float64)
            _lib.check(L.brutus_offsets_bootstrap(
                b, Nobj, Nsamps, Nfilt, n, Nmc, t_s.data_ptr(), t_co.data_ptr(), t_u.data_ptr(),
                t_flux.data_ptr(), t_cdf.data_ptr(), t_phot.data_ptr(), t_ws.data_ptr(),
                t_ws.numel(), t_meds.data_ptr(), sp))
            meds = t_meds.cpu().numpy()
            ratios[b], ratios_err[b] = np.median(meds), np.std(meds)
    if verbose:
        sys.stderr.write('\n')
    if prior_mean is not None and prior_std is not None:
        var = ratios_err ** 2 + prior_std ** 2
        ratios = (ratios * prior_std ** 2 + prior_mean * ratios_err ** 2) / var
        ratios_err = ratios_err * prior_std / np.sqrt(var)
    return ratios, ratios_err, nratio


# ---------------------------------------------------------------------------
# the rest of `brutus.utils.__all__` (utils.py:28-38): small host helpers of the reference's
# scripts and notebooks, restated so that `from brutus_amd.utils import ...` finds every
# name the reference exports.  None of them is on the grid-likelihood path.
# ---------------------------------------------------------------------------
class _function_wrapper(object):
    """Picklable closure `x -> func(x, *args, **kwargs)` (the role of reference utils.py:43-68):
    a failure inside `func` is logged to stderr together with the call's arguments before it
    propagates, so that a worker pool's traceback says which evaluation died."""

    def __init__(self, func, args, kwargs, name='input'):
        self.func = func
        self.args = tuple(args)
        self.kwargs = dict(kwargs)
        self.name = name

    def __call__(self, x):
        try:
            return self.func(x, *self.args, **self.kwargs)
        except Exception as exc:
            import sys
            sys.stderr.write("brutus_amd: the %s function raised %s: %s\n  at x = %r\n  with args = %r, "
                             "kwargs = %r\n" % (self.name, type(exc).__name__, exc, x, self.args,
                                                self.kwargs))
            raise


def _dot3(A, B):
    """Row-wise dot products over the last axis (utils.py:86-93)."""
    return np.einsum('...i,...i->...', A, B)


def _adjoint3(A):
    """Adjugate-transpose of a stack of 3x3 matrices: row i = cross product of the two rows
    after it, cyclically (utils.py:71-83)."""
    A = np.asarray(A)
    out = np.empty_like(A)
    out[..., 0, :] = np.cross(A[..., 1, :], A[..., 2, :])
    out[..., 1, :] = np.cross(A[..., 2, :], A[..., 0, :])
    out[..., 2, :] = np.cross(A[..., 0, :], A[..., 1, :])
    return out


def _inverse_transpose3(A):
    """Inverse-transpose of a stack of 3x3 matrices (utils.py:96-105): adjugate-transpose over
    the mean of its three row dots with `A`."""
    adj = _adjoint3(A)
    return adj / _dot3(adj, A).mean(axis=-1)[..., None, None]


def _isPSD(A):
    """True if `A` has a Cholesky factor (utils.py:117-127)."""
    try:
        np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return False
    return True


def _truncnorm_terms(x, a, b, loc, scale):
    from math import erf, sqrt
    lo, hi = scale * a + loc, scale * b + loc          # bounds in the units of x
    z = (x - loc) / scale
    mass = erf(b / sqrt(2.)) - erf(a / sqrt(2.))         # 2 (Phi(b) - Phi(a))
    outside = np.logical_or(x < lo, x > hi)
    return z, mass, outside


def _truncnorm_pdf(x, a, b, loc=0.0, scale=1.0):
    """pdf of a normal (`loc`, `scale`) truncated to `[a, b]` standard deviations, like
    `scipy.stats.truncnorm.pdf` (utils.py:179-229)."""
    z, mass, outside = _truncnorm_terms(x, a, b, loc, scale)
    ans = np.exp(-0.5 * z ** 2) / np.sqrt(2. * np.pi) / (scale * 0.5 * mass)
    if isinstance(x, (float, int)):
        return 0 if outside else ans
    ans[outside] = 0
    return ans


def _truncnorm_logpdf(x, a, b, loc=0.0, scale=1.0):
    """ln of `_truncnorm_pdf` (utils.py:232-283)."""
    z, mass, outside = _truncnorm_terms(x, a, b, loc, scale)
    ans = np.subtract(-log(np.sqrt(2. * np.pi)) - 0.5 * np.square(z),
                      log(scale / 2.) + log(mass))
    if isinstance(x, (float, int)):
        return -np.inf if outside else ans
    ans[outside] = -np.inf
    return ans


def _get_seds(mag_coeffs, av, rv, return_flux=False):
    """The reference's numba kernel (utils.py:286-347) as a host function:
    `(seds, rvecs, drvecs)` of `(Nmodel, Nband, 3)` coefficients at per-model `av`, `rv`."""
    return get_seds(mag_coeffs, av=av, rv=rv, return_flux=return_flux, return_rvec=True,
                    return_drvec=True)


def quantile(x, q, weights=None):
    """Sample quantiles `q` (in [0, 1]) of `x`, optionally weighted (the contract of reference
    utils.py:718-762).  Unweighted: linear interpolation between order statistics
    (`numpy.quantile`).  Weighted: the sorted samples are placed at the cumulative weight that
    PRECEDES them, normalised by the total without the last sample's weight, and `q` is
    interpolated linearly on that curve."""
    samples = np.atleast_1d(np.asarray(x))
    qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if qs.size and (qs.min() < 0. or qs.max() > 1.):
        raise ValueError("Quantiles must be between 0. and 1.")
    if weights is None:
        return np.quantile(samples, qs)
    w = np.atleast_1d(np.asarray(weights, dtype=np.float64))
    if w.shape[0] != samples.shape[0]:
        raise ValueError("Dimension mismatch: len(weights) != len(x).")
    rank = np.argsort(samples, kind="stable")
    before = np.concatenate([[0.], np.cumsum(w[rank])[:-1]])     # weight in front of each sample
    return np.interp(qs, before / before[-1], samples[rank]).tolist()


def luptitude(phot, err, skynoise=1., zeropoints=1.):
    """asinh magnitudes and their errors (utils.py:978-1017)."""
    mag = -2.5 / np.log(10.) * (np.arcsinh(phot / (2. * skynoise)) + np.log(skynoise / zeropoints))
    mag_err = np.sqrt(np.square(2.5 * np.log10(np.e) * err)
                      / (np.square(2. * skynoise) + np.square(phot)))
    return mag, mag_err


def inv_luptitude(mag, err, skynoise=1., zeropoints=1.):
    """Inverse of `luptitude` (utils.py:1020-1059)."""
    phot = (2. * skynoise) * np.sinh(np.log(10.) / -2.5 * mag - np.log(skynoise / zeropoints))
    phot_err = np.sqrt((np.square(2. * skynoise) + np.square(phot)) * np.square(err)) \
        / (2.5 * np.log10(np.e))
    return phot, phot_err


def add_mag(mag1, mag2, f1=1., f2=1.):
    """Magnitude of `f1 flux(mag1) + f2 flux(mag2)` (utils.py:1062-1086)."""
    return -2.5 * np.log10(f1 * 10 ** (-0.4 * mag1) + f2 * 10 ** (-0.4 * mag2))


def _no_download(what):
    def fetch(target_dir=".", *args, **kwargs):
        raise NotImplementedError(
            "brutus_amd.utils.%s: downloading %s from the Harvard Dataverse (reference "
            "utils.py:363-517, via pooch) is outside this package's scope -- fetch the file with "
            "the reference package or by hand and pass its path to load_models / load_offsets"
            % (fetch.__name__, what))
    return fetch


fetch_isos = _no_download("the MIST isochrone file")
fetch_isos.__name__ = "fetch_isos"
fetch_tracks = _no_download("the MIST evolutionary tracks")
fetch_tracks.__name__ = "fetch_tracks"
fetch_dustmaps = _no_download("the Bayestar dust map")
fetch_dustmaps.__name__ = "fetch_dustmaps"
fetch_grids = _no_download("a pre-computed model grid")
fetch_grids.__name__ = "fetch_grids"
fetch_offsets = _no_download("the photometric offsets file")
fetch_offsets.__name__ = "fetch_offsets"
fetch_nns = _no_download("the bolometric-correction network")
fetch_nns.__name__ = "fetch_nns"
