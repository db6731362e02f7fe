"""Host-side priors on the fit() path (numpy, float64).

Counterparts of `brutus/pdf.py`: `imf_lnprior` (pdf.py:38-108),
`ps1_MrLF_lnprior` (pdf.py:111-141), `parallax_lnprior` (pdf.py:144-175),
`scale_parallax_lnprior` (pdf.py:178-222), `parallax_to_scale`
(pdf.py:225-260).  The full-grid application of the scale-parallax term is
done on the device (k_finalize); these host versions serve the public API and
the Monte Carlo stage of `lnpost`, which acts on selected models only.
"""
import os
import warnings

import numpy as np

from .galprior import (gal_lnprior, logn_disk, logn_halo, logp_age_from_feh,   # noqa: F401
                       logp_feh)

# the reference's `brutus.pdf.__all__` (pdf.py:30-35), plus the table type of the Bayestar-free
# dust interface
__all__ = ["imf_lnprior", "ps1_MrLF_lnprior", "parallax_lnprior",
           "scale_parallax_lnprior", "parallax_to_scale",
           "logn_disk", "logn_halo",
           "logp_feh", "logp_age_from_feh",
           "gal_lnprior", "dust_lnprior",
           "bin_pdfs_distred",
           "LOSTable"]


def _kroupa_segment(m, alpha_low, alpha_high, mass_break):
    out = np.full(m.shape, -np.inf)
    lo = (m > 0.08) & (m <= mass_break)
    hi = m > mass_break
    with np.errstate(all="ignore"):
        out[lo] = -alpha_low * np.log(m[lo])
        out[hi] = (-alpha_high * np.log(m[hi])
                   + (alpha_high - alpha_low) * np.log(mass_break))
    return out


def imf_lnprior(mgrid, alpha_low=1.3, alpha_high=2.3, mass_break=0.5,
                mgrid2=None):
    """Kroupa broken-power-law ln prior over initial mass; -inf at or below
    the hydrogen-burning limit 0.08 Msun (reference pdf.py:72-108)."""
    m = np.asarray(mgrid, dtype=np.float64)
    lnp = _kroupa_segment(m, alpha_low, alpha_high, mass_break)
    n_low = mass_break ** (1. - alpha_low) / (alpha_high - 1.)
    n_high = (0.08 ** (1. - alpha_low) - mass_break ** (1. - alpha_low)) \
        / (alpha_low - 1.)
    norm = n_low + n_high
    if mgrid2 is not None:
        lnp = lnp + _kroupa_segment(np.asarray(mgrid2, dtype=np.float64),
                                    alpha_low, alpha_high, mass_break)
        norm = n_low ** 2 + n_high ** 2 + 2 * n_low * n_high
    return lnp - np.log(norm)


_PS_TABLE = None


def ps1_MrLF_lnprior(Mr):
    """PS1 r-band luminosity-function prior: linear interpolation (with linear
    extrapolation) of a two-column (M_r, ln LF) table (reference pdf.py:111-141).

    The table ships with the package (`PSMrLF_lnprior.dat`, the reference's data
    file brutus/PSMrLF_lnprior.dat, unchanged); $BRUTUS_AMD_PSLF overrides it.
    """
    global _PS_TABLE
    if _PS_TABLE is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.environ.get("BRUTUS_AMD_PSLF",
                              os.path.join(here, "PSMrLF_lnprior.dat"))
        if not os.path.exists(path):
            raise IOError("PS1 luminosity-function table not found at %s; copy "
                          "brutus/PSMrLF_lnprior.dat there or set "
                          "BRUTUS_AMD_PSLF" % path)
        _PS_TABLE = np.loadtxt(path).T
    gx, gy = _PS_TABLE
    Mr = np.asarray(Mr, dtype=np.float64)
    out = np.interp(Mr, gx, gy)
    lo, hi = Mr < gx[0], Mr > gx[-1]
    out = np.where(lo, gy[0] + (Mr - gx[0]) * (gy[1] - gy[0]) / (gx[1] - gx[0]), out)
    out = np.where(hi, gy[-1] + (Mr - gx[-1]) * (gy[-1] - gy[-2]) / (gx[-1] - gx[-2]), out)
    return out


def parallax_lnprior(parallaxes, p_meas, p_err):
    """Gaussian ln prior in parallax; flat when no measurement (pdf.py:166-173)."""
    parallaxes = np.asarray(parallaxes, dtype=np.float64)
    if not (np.isfinite(p_meas) and np.isfinite(p_err)):
        return np.zeros_like(parallaxes)
    with np.errstate(all="ignore"):
        return -0.5 * ((parallaxes - p_meas) ** 2 / p_err ** 2
                       + np.log(2. * np.pi * p_err ** 2))


def parallax_to_scale(p_meas, p_err, snr_lim=4.):
    """Moments of s = p^2 for a Normal parallax (pdf.py:249-258)."""
    if p_meas / p_err > snr_lim:
        pm = max(0., p_meas)
        return pm ** 2 + p_err ** 2, np.sqrt(2 * p_err ** 4 + 4 * pm ** 2 * p_err ** 2)
    return 1e-20, 1e20


def scale_parallax_lnprior(scales, scale_errs, p_meas, p_err, snr_lim=4.):
    """Gaussian ln prior in scale s ~ p^2, applied only for S/N > snr_lim
    (pdf.py:209-220)."""
    scales = np.asarray(scales, dtype=np.float64)
    if not (np.isfinite(p_meas) and np.isfinite(p_err)
            and p_meas / p_err > snr_lim):
        return np.zeros_like(scales)
    s_mean, s_std = parallax_to_scale(p_meas, p_err, snr_lim=snr_lim)
    with np.errstate(all="ignore"):
        var = s_std ** 2 + np.asarray(scale_errs, dtype=np.float64) ** 2
        return -0.5 * ((scales - s_mean) ** 2 / var + np.log(2. * np.pi * var))


# ---------------------------------------------------------------------------
# 3-D dust prior without Bayestar / healpy
# ---------------------------------------------------------------------------
class LOSTable(object):
    """Line-of-sight reddening profiles supplied by the caller: what the reference
    obtains from `dust.Bayestar.query(coord)` (dust.py:184-299), i.e. for a
    sightline the arrays `(av_dist [kpc], av_mean, av_err)`, without the Bayestar
    HDF5 map or healpy.

    `LOSTable(l, b, dist, av_mean, av_err)`: `l, b` (Nlos,) Galactic degrees of the
    tabulated sightlines, `dist` (Ndist,) kpc, `av_mean`, `av_err` (Nlos, Ndist).
    `query(coord)` returns the profile of the nearest tabulated sightline (great-
    circle distance); profiles with NaNs mean "no coverage", like the reference.
    `LOSTable.load(path)` reads the same five arrays from an `.npz` file.
    """

    def __init__(self, l, b, dist, av_mean, av_err):
        self.l = np.atleast_1d(np.asarray(l, dtype=np.float64))
        self.b = np.atleast_1d(np.asarray(b, dtype=np.float64))
        self.dist = np.asarray(dist, dtype=np.float64)
        self.av_mean = np.atleast_2d(np.asarray(av_mean, dtype=np.float64))
        self.av_err = np.atleast_2d(np.asarray(av_err, dtype=np.float64))
        if self.av_mean.shape != (self.l.size, self.dist.size) or \
                self.av_err.shape != self.av_mean.shape or self.b.size != self.l.size:
            raise ValueError("LOSTable: av_mean / av_err must be (Nlos, Ndist)")

    @classmethod
    def load(cls, path):
        z = np.load(path)
        return cls(z["l"], z["b"], z["dist"], z["av_mean"], z["av_err"])

    def query(self, coord):
        l0, b0 = np.deg2rad(coord[0]), np.deg2rad(coord[1])
        l, b = np.deg2rad(self.l), np.deg2rad(self.b)
        cosd = np.sin(b0) * np.sin(b) + np.cos(b0) * np.cos(b) * np.cos(l - l0)
        k = int(np.argmax(cosd))
        return self.dist, self.av_mean[k], self.av_err[k]


def _los_provider(dustfile):
    if dustfile is None:
        raise ValueError("dust_lnprior needs a line-of-sight table: pass "
                         "`dustfile=` a LOSTable, an object with `.query(coord)`, a "
                         "callable `coord -> (dist, av_mean, av_err)` or the path of "
                         "an .npz file with arrays l, b, dist, av_mean, av_err")
    if hasattr(dustfile, "query"):
        return dustfile.query
    if callable(dustfile):
        return dustfile
    if isinstance(dustfile, str) and dustfile.endswith(".npz"):
        return LOSTable.load(dustfile).query
    raise NotImplementedError(
        "dustfile=%r: reading the Bayestar HDF5 map needs healpy, which is outside "
        "this package's scope; convert the sightlines you need to a LOSTable "
        "(see brutus_amd.pdf.LOSTable)" % (dustfile,))


def los_tables(dustfile, coords):
    """Line-of-sight profiles of a batch of objects as one array for the device stage:
    `(los (N, 3, nd) = dist, Av_mean, Av_err; ok (N,) int32)`.  Profiles shorter than the
    longest of the batch are padded by repeating their last node, which leaves
    `numpy.interp` (end value beyond the table) and the device's interpolation unchanged.
    Called per batch (`coords[a:b]`), so neither the Python loop over sightlines nor the
    table ever spans the whole catalogue."""
    q = _los_provider(dustfile)
    rows, ok = [], []
    for c in np.asarray(coords, dtype=np.float64):
        d, m, e = (np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in q(c))
        if not (d.shape == m.shape == e.shape) or d.ndim != 1 or d.size < 1:
            raise ValueError("a line-of-sight profile is three equally long 1-d arrays "
                             "(dist, mean, err)")
        if d.size == 1:
            # `numpy.interp` (the host form) takes a one-node profile as a constant; a second
            # node with the same values further out says the same to the device stage
            d = np.array([d[0], d[0] + 1.])
            m, e = np.repeat(m, 2), np.repeat(e, 2)
        good = bool(np.all(np.isfinite(m) & np.isfinite(e)))
        ok.append(1 if good else 0)
        rows.append(np.stack([d, np.where(np.isfinite(m), m, 0.), np.where(np.isfinite(e), e, 0.)]))
    nd = max(r.shape[1] for r in rows)
    rows = [r if r.shape[1] == nd else np.concatenate(
        [r, np.repeat(r[:, -1:], nd - r.shape[1], axis=1)], axis=1) for r in rows]
    return np.stack(rows), np.asarray(ok, dtype=np.int32)


def dust_lnprior(dists, coord, avs, dustfile=None, offset=0., scale=1., smooth=1.,
                 scatter=0.2, return_components=False):
    """ln prior of a 3-D dust model: Gaussian in Av around the line-of-sight
    profile interpolated at `dists` (reference pdf.py:752-840, same arithmetic);
    flat if the sightline has no coverage.  The profile comes from `dustfile`,
    here a caller-supplied table instead of the Bayestar map (see `LOSTable`).
    Same hook signature as the reference: `lndustprior(dists, coord, avs, dustfile=)`.
    """
    av_dist, av_mean, av_err = _los_provider(dustfile)(coord)
    dists = np.asarray(dists, dtype=np.float64)
    avs = np.asarray(avs, dtype=np.float64)
    if np.all(np.isfinite(av_mean) & np.isfinite(av_err)):
        av_mean = scale * np.interp(dists, av_dist, av_mean) + offset
        av_err = smooth * scale * np.interp(dists, av_dist, av_err)
        av_err = np.sqrt(av_err ** 2 + scatter ** 2)
        with np.errstate(all="ignore"):
            chi2 = (avs - av_mean) ** 2 / av_err ** 2
            lnorm = np.log(2. * np.pi * av_err ** 2)
        lnprior = -0.5 * (chi2 + lnorm)
    else:
        lnprior = np.zeros_like(avs)
    if not return_components:
        return lnprior
    return lnprior, (av_mean, av_err)


def bin_pdfs_distred(data, cdf=False, ebv=False, dist_type='distance_modulus',
                     lndistprior=None, coord=None, avlim=(0., 6.), rvlim=(1., 8.),
                     parallaxes=None, parallax_errors=None, Nr=100,
                     bins=(750, 300), span=None, smooth=0.01, rstate=None,
                     verbose=False):
    """Binned 2-D (distance, reddening) posteriors of a set of fitted objects, the input of
    the line-of-sight fits and of `plotting.dist_vs_red`; same arguments and return values as
    reference `pdf.bin_pdfs_distred` (pdf.py:843-1113).  Host numpy: it acts on the few
    hundred draws per object that `fit()` wrote.

    `data` is `(dists, reds, dreds)` as saved by `fit(save_dar_draws=True)`, each
    `(Nobj, Nsamps)`, or `(scales, avs, rvs, covs_sar)`, from which `Nr` realisations per draw
    are regenerated with `utils.draw_sar` and re-weighted by the distance prior
    `lndistprior(dists, coord)` (default: the Galactic prior) and the parallax likelihood.
    Returns `(binned_vals (Nobj, Nxbin, Nybin) float32, xedges, yedges)`; every object's
    histogram is divided by `Nsamps` and smoothed with a Gaussian whose width along the
    distance axis is capped by the object's parallax error.
    """
    import sys
    from scipy.ndimage import gaussian_filter
    from scipy.special import logsumexp
    from .utils import draw_sar
    nobjs, nsamps = np.shape(data[0])[:2]
    if rstate is None:
        rstate = getattr(np, "random_intel", np.random)
    if dist_type not in ('parallax', 'scale', 'distance', 'distance_modulus'):
        raise ValueError("The provided `dist_type` is not valid.")
    regenerate = len(data) != 3
    if regenerate and lndistprior is None and coord is None:
        raise ValueError("`coord` must be passed if the default distance "
                         "prior was used.")
    if lndistprior is None:
        lndistprior = gal_lnprior
    parallaxes = (np.full(nobjs, np.nan) if parallaxes is None
                  else np.asarray(parallaxes, dtype=np.float64))
    parallax_errors = (np.full(nobjs, np.nan) if parallax_errors is None
                       else np.asarray(parallax_errors, dtype=np.float64))

    # bin edges: reddening along y, the chosen distance measure along x
    if span is None:
        avlims, dlims = avlim, 10. ** (np.array([4., 19.]) / 5. - 2.)
    else:
        avlims, dlims = span
    dlims = np.asarray(dlims, dtype=np.float64)
    try:
        xbin, ybin = bins
    except TypeError:
        xbin = ybin = bins
    to_x = {'scale': lambda d: 1. / d ** 2, 'parallax': lambda d: 1. / d,
            'distance': lambda d: d, 'distance_modulus': lambda d: 5. * np.log10(d) + 10.}[dist_type]
    xlims = to_x(dlims[::-1]) if dist_type in ('scale', 'parallax') else to_x(dlims)
    ylims = avlims
    xbins = np.linspace(xlims[0], xlims[1], xbin + 1)
    ybins = np.linspace(ylims[0], ylims[1], ybin + 1)
    dx, dy = xbins[1] - xbins[0], ybins[1] - ybins[0]
    xspan, yspan = xlims[1] - xlims[0], ylims[1] - ylims[0]
    # smoothing widths: a fraction of the span below 1, a number of bins from 1 on
    try:
        sx, sy = smooth[0], smooth[1]
    except (TypeError, IndexError):
        sx = sy = smooth
    xsmooth = sx * xspan if sx < 1 else sx * dx
    ysmooth = sy * yspan if sy < 1 else sy * dy

    binned = np.zeros((nobjs, xbin, ybin), dtype='float32')
    xedges, yedges = xbins, ybins
    for i in range(nobjs):
        if verbose:
            sys.stderr.write('\rBinning object {0}/{1}'.format(i + 1, nobjs))
        if not regenerate:
            d = np.array(data[0][i], dtype=np.float64)
            y = np.array(data[1][i], dtype=np.float64)
            if ebv:
                y = y / np.asarray(data[2][i], dtype=np.float64)
            weights = None
        else:
            sd, ad, rd = draw_sar(data[0][i], data[1][i], data[2][i], data[3][i], ndraws=Nr,
                                  avlim=avlim, rvlim=rvlim, rstate=rstate)
            with np.errstate(all="ignore"):
                pd = np.sqrt(sd)
                d = 1. / pd
                lnp = np.array(lndistprior(d, coord[i]), dtype=np.float64)
                lnp = lnp + parallax_lnprior(pd, parallaxes[i], parallax_errors[i])
                w = np.exp(lnp - logsumexp(lnp, axis=1)[:, None])
                w /= w.sum(axis=1)[:, None]
            weights = w.reshape(-1)
            y = ad.reshape(-1)
            if ebv:
                y = y / rd.reshape(-1)
            d = d.reshape(-1)
        with np.errstate(all="ignore"):
            H, xedges, yedges = np.histogram2d(to_x(d), y, bins=(xbins, ybins), weights=weights)
        # the parallax caps the smoothing along the distance axis
        p1 = np.array([parallaxes[i] + parallax_errors[i],
                       max(parallaxes[i] - parallax_errors[i], 1e-10)])
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            cap = abs(np.diff({'scale': p1 ** 2, 'parallax': p1, 'distance': 1. / p1,
                               'distance_modulus': 5. * np.log10(1. / p1)}[dist_type])[0]) / 2.
        xs = min(cap, xsmooth) if np.isfinite(cap) else xsmooth
        binned[i] = gaussian_filter((H / nsamps).astype('float32'), (xs / dx, ysmooth / dy))
    if cdf:
        for i in range(nobjs):
            binned[i] = binned[i].cumsum(axis=0)
    return binned, xedges, yedges
