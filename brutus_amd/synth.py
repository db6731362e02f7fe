"""Synthetic model grids and star catalogues.

The real MIST v9 grid (`grid_mist_v9.h5`) cannot be downloaded here, so the
benchmark and the parity tests use a seed-fixed synthetic grid with the same
shape, dtype and value ranges as `utils.load_models` returns
(reference `brutus/utils.py:588-591`: `(Nmodel, Nfilt, 3)` float32 holding
`(mag, R, dR/dRv)` per band at 1 kpc).
"""
import numpy as np

__all__ = ["make_grid", "make_stars", "make_sharp_grid", "GRID_SEED"]

GRID_SEED = 20250523


def make_grid(nmodel=750000, nfilt=12, seed=GRID_SEED):
    """Return (models f32 (nmodel, nfilt, 3), labels structured, labels_mask).

    Absolute-magnitude sequence M ~ U(-2, 12) with a smooth colour term per
    band; reddening vector falling from 1.25 (bluest band) to 0.12 (reddest);
    small dR/dRv.  Labels carry the fields the static prior uses
    (`mini, eep, feh, loga, agewt`), laid out like the reference's grid files.
    """
    rng = np.random.RandomState(seed)
    M = rng.uniform(-2., 12., size=nmodel)
    kb = np.linspace(0.8, -0.8, nfilt)
    colour = kb[None, :] * (0.3 + 0.12 * M[:, None])
    colour += rng.normal(0., 0.03, size=(nmodel, nfilt))
    mag = M[:, None] + colour
    r0 = np.linspace(1.25, 0.12, nfilt)[None, :] * (
        1. + rng.normal(0., 0.01, size=(nmodel, nfilt)))
    dr = np.linspace(0.06, -0.01, nfilt)[None, :] * (
        1. + rng.normal(0., 0.05, size=(nmodel, nfilt)))
    models = np.stack([mag, r0, dr], axis=-1).astype(np.float32)

    ltype = np.dtype([('mini', 'f8'), ('eep', 'f8'), ('feh', 'f8'),
                      ('loga', 'f8'), ('agewt', 'f8')])
    labels = np.zeros(nmodel, dtype=ltype)
    # gridded labels take values on a lattice so that the spacing prior
    # (reference fitting.py:1351-1359) has something to work with.
    labels['mini'] = np.round(rng.uniform(0.5, 2.0, nmodel) / 0.025) * 0.025
    labels['eep'] = np.round(rng.uniform(202, 808, nmodel) / 2.) * 2.
    labels['feh'] = np.round(rng.uniform(-3., 0.5, nmodel) / 0.05) * 0.05
    labels['loga'] = rng.uniform(8., 10.14, nmodel)
    labels['agewt'] = rng.uniform(0.1, 2., nmodel)
    mtype = np.dtype([(n, '?') for n in ltype.names])
    labels_mask = np.zeros(1, dtype=mtype)
    labels_mask['mini'] = True
    labels_mask['eep'] = True
    labels_mask['feh'] = True
    return models, labels, labels_mask


def make_stars(models, nstar, seed=1, with_parallax=True, frac_no_parallax=0.25,
               min_frac_err=0.02, frac_err=None, parallax_snr=None, av_range=(0., 2.5)):
    """Draw `nstar` synthetic stars from the grid.

    Returns dict with flux, err (nstar, nfilt) f64 in maggies, mask (bool, all
    True), parallax / parallax_err (mas; NaN where absent), coords (l, b) deg
    and the truth columns.  Fractional flux errors follow the spread seen in
    the reference's Orion demo fixture (median 0.025-0.06 mag, tail to 0.14).
    `frac_err`: one fractional error for every band instead (0.02 = S/N 50);
    `parallax_snr`: parallax errors as that fraction of the true parallax instead of the
    log-uniform 0.05-1.5 mas.  `av_range`: the true extinctions are uniform in it.
    """
    rng = np.random.RandomState(seed)
    nmodel, nfilt, _ = models.shape
    idx = rng.randint(0, nmodel, size=nstar)
    av = rng.uniform(av_range[0], av_range[1], size=nstar)
    rv = np.clip(rng.normal(3.32, 0.18, size=nstar), 1., 8.)
    dist = 10. ** rng.uniform(np.log10(0.1), np.log10(5.), size=nstar)  # kpc
    c = models[idx].astype(np.float64)
    sed = c[:, :, 0] + av[:, None] * (c[:, :, 1] + rv[:, None] * c[:, :, 2])
    flux_true = 10. ** (-0.4 * sed) / dist[:, None] ** 2
    frac = np.maximum(min_frac_err,
                      10. ** rng.normal(np.log10(0.04), 0.25,
                                        size=(nstar, nfilt)))
    if frac_err is not None:
        frac = np.full((nstar, nfilt), float(frac_err))
    err = frac * flux_true
    flux = flux_true + rng.normal(size=(nstar, nfilt)) * err
    mask = np.ones((nstar, nfilt), dtype=bool)
    if with_parallax:
        perr = 10. ** rng.uniform(np.log10(0.05), np.log10(1.5), size=nstar)
        if parallax_snr is not None:
            perr = (1. / dist) / float(parallax_snr)
        par = 1. / dist + rng.normal(size=nstar) * perr
        drop = rng.uniform(size=nstar) < frac_no_parallax
        par[drop] = np.nan
        perr[drop] = np.nan
    else:
        par = np.full(nstar, np.nan)
        perr = np.full(nstar, np.nan)
    coords = np.stack([rng.uniform(0., 360., nstar),
                       rng.uniform(-90., 90., nstar)], axis=1)
    return dict(flux=flux, err=err, mask=mask, parallax=par, parallax_err=perr,
                coords=coords, true_idx=idx, true_av=av, true_rv=rv,
                true_dist=dist)


def make_sharp_grid(nmodel=750000, nfilt=12, seed=GRID_SEED):
    """`make_mist_like_grid` with colours that reddening cannot imitate: the temperature term of
    the SED is QUADRATIC in the band index (peaked mid-spectrum) where the reddening vector
    falls monotonically, so a (temperature, Av) trade-off that keeps all bands within their
    errors does not exist.  With S/N 50 photometry and a parallax at S/N 10
    (`make_stars(..., frac_err=0.02, parallax_snr=10., frac_no_parallax=0.)`) a few per cent
    of the grid pass the first cut per object -- the regime of BASELINE.md section 2 and of
    the demo notebooks, where the float32 proof pass and the bookkeeping are the call."""
    models, labels, labels_mask = make_mist_like_grid(nmodel, nfilt, seed)
    rng = np.random.RandomState(seed + 1)
    lam = np.linspace(0., 1., nfilt)
    x = (labels['eep'] - 202.) / 606.
    t = 0.25 + 0.45 * (labels['mini'] - 0.5) / 1.5 - 0.3 * x - 0.04 * (labels['feh'] + 1.)
    mag0 = models[:, :, 0].astype(np.float64)
    M = mag0.mean(axis=1)
    bump = 1. - (2. * lam - 1.) ** 2                     # 0 at both ends, 1 in the middle
    colour = (1. - t)[:, None] * (2.4 * bump - 1.6)[None, :]
    colour += 0.35 * (labels['feh'] + 1.)[:, None] * np.cos(3. * np.pi * lam)[None, :]
    mag = M[:, None] + colour + rng.normal(0., 0.004, size=(nmodel, nfilt))
    models = models.copy()
    models[:, :, 0] = mag.astype(np.float32)
    return models, labels, labels_mask


def make_mist_like_grid(nmodel=750000, nfilt=12, seed=GRID_SEED):
    """A lattice-ordered synthetic grid that mimics the structure of the real
    MIST grid files (reference seds.py:754-765: `mini` x `eep` x `feh`
    lattice, 61 x 220 x 61 = 818 620 points before invalid models are
    dropped, ~750k kept): models are ORDERED along the lattice, luminosity and
    effective temperature vary smoothly with (mini, eep), metallicity nudges
    the colours, and the SED shape responds to temperature differently from
    how it responds to extinction -- so, as with real grids, a star's
    posterior occupies compact runs of the index space instead of the whole
    grid.  Shapes, dtype and value ranges match `utils.load_models`
    (reference utils.py:588-591).
    """
    rng = np.random.RandomState(seed)
    n_mini, n_feh = 61, 61
    n_eep = int(np.ceil(nmodel / float(n_mini * n_feh)))
    mini = np.linspace(0.5, 2.0, n_mini)
    eep = np.linspace(202., 808., n_eep)
    feh = np.linspace(-3., 0.5, n_feh)
    mm, ee, ff = np.meshgrid(mini, eep, feh, indexing="ij")
    mm, ee, ff = mm.ravel()[:nmodel], ee.ravel()[:nmodel], ff.ravel()[:nmodel]
    x = (ee - 202.) / 606.                       # 0 = ZAMS ... 1 = tip of the RGB
    ms = x < 0.42                                # main sequence up to EEP ~454
    # log-temperature proxy t in ~[0, 1] (0 = cool, 1 = hot) and absolute mag M
    t_ms = 0.25 + 0.45 * (mm - 0.5) / 1.5 - 0.08 * (x / 0.42) ** 2
    xp = np.clip((x - 0.42) / 0.58, 0., None)
    t_pm = t_ms - 0.55 * xp ** 0.7
    t = np.where(ms, t_ms, t_pm) - 0.04 * (ff + 1.)
    M_ms = 7.5 - 4.8 * (mm - 0.5) / 1.5 - 0.9 * (x / 0.42)
    M_pm = M_ms - 5.5 * xp ** 1.3
    M = np.where(ms, M_ms, M_pm) + 0.25 * (ff + 1.)
    lam = np.linspace(0., 1., nfilt)             # 0 = bluest band, 1 = reddest
    # colour vs temperature: blackbody-like (steep in the blue, flat in the red)
    colour = (1. - t)[:, None] * (3.2 * (1. - lam) ** 1.6 - 0.9)[None, :]
    colour += 0.15 * (ff + 1.)[:, None] * (np.exp(-6. * lam))[None, :]   # line blanketing
    mag = M[:, None] + colour + rng.normal(0., 0.004, size=(nmodel, nfilt))
    r0 = (1.25 - 1.13 * lam ** 0.8)[None, :] * (1. + 0.03 * (t - 0.4))[:, None]
    dr = (0.06 - 0.07 * lam)[None, :] * (1. + 0.05 * (t - 0.4))[:, None]
    models = np.stack([mag, r0, dr], axis=-1).astype(np.float32)

    ltype = np.dtype([('mini', 'f8'), ('eep', 'f8'), ('feh', 'f8'),
                      ('loga', 'f8'), ('agewt', 'f8')])
    labels = np.zeros(nmodel, dtype=ltype)
    labels['mini'], labels['eep'], labels['feh'] = mm, ee, ff
    # ages stay below 13.5 Gyr so that age priors never exclude the whole grid
    labels['loga'] = np.clip(9.9 - 2.2 * np.log10(mm) + 0.2 * x, 8.0, 10.13)
    labels['agewt'] = 0.05 + np.abs(np.gradient(labels['loga']))
    mtype = np.dtype([(n, '?') for n in ltype.names])
    labels_mask = np.zeros(1, dtype=mtype)
    for n in ('mini', 'eep', 'feh'):
        labels_mask[n] = True
    return models, labels, labels_mask


class TableIsochrone(object):
    """A co-eval population as a precomputed SED table:
    `mags[smf index]` is `(Neep, Nbands)` absolute magnitudes on `eep_grid`,
    `mini` the initial-mass grid.  Provides the one method
    `cluster.isochrone_loglike` needs from reference `seds.Isochrone`
    (`get_seds`, cluster.py:339-344); reddening and distance are applied here
    on the host, the (object x point) block runs on the device."""

    def __init__(self, nbands=12, neep=2000, smf_grid=None, seed=7):
        rng = np.random.RandomState(seed)
        self.smf_grid = np.asarray(
            (0., 0.2, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9,
             0.95, 1.0) if smf_grid is None else smf_grid, dtype=float)
        self.eep_grid = np.linspace(202., 808., neep)
        x = (self.eep_grid - 202.) / 606.
        self.mini = 0.25 + 1.5 * x + 0.02 * x * x
        lam = np.linspace(0., 1., nbands)
        M = 9.5 - 8.5 * x
        col = (0.9 - 0.7 * x)[:, None] * (2.5 * (1. - lam) ** 1.5 - 0.7)[None, :]
        base = M[:, None] + col + 0.01 * rng.normal(size=(neep, nbands))
        self.rvec = 1.25 - 1.1 * lam
        self.mags = []
        for smf in self.smf_grid:           # unresolved equal-age secondary
            self.mags.append(base - 2.5 * np.log10(1. + smf ** 3.5))

    def get_seds(self, feh=0., loga=9., av=0., rv=3.3, eep=None, smf=0.,
                 dist=1000., mini_bound=0.08, eep_binary_max=480.,
                 corr_params=None):
        k = int(np.argmin(np.abs(self.smf_grid - smf)))
        eep = self.eep_grid if eep is None else np.asarray(eep, dtype=float)
        mag = np.empty((eep.size, self.rvec.size))
        for b in range(self.rvec.size):
            mag[:, b] = np.interp(eep, self.eep_grid, self.mags[k][:, b])
        mini = np.interp(eep, self.eep_grid, self.mini)
        mag = mag + 0.25 * feh - 0.15 * (loga - 9.)
        mag = mag + av * (self.rvec + 0.02 * (rv - 3.3))[None, :]
        mag = mag + 5. * np.log10(dist / 10.)
        mag[mini < mini_bound] = np.nan
        return mag, {"mini": mini}, {"mini": mini * smf}


    def get_seds_grid(self, smf_grid=None, feh=0., loga=9., av=0., rv=3.3, eep=None,
                      dist=1000., mini_bound=0.08, eep_binary_max=480., corr_params=None,
                      out=None):
        """All mass-fraction slices in one call (the batched hook
        `cluster.isochrone_loglike` looks for): `(mags (Nsmf, Neep, Nbands), mini (Neep,))`,
        slice by slice what `get_seds` returns."""
        smf_grid = self.smf_grid if smf_grid is None else np.asarray(smf_grid, dtype=float)
        ks = [int(np.argmin(np.abs(self.smf_grid - smf))) for smf in smf_grid]
        if not hasattr(self, "_stack"):
            self._stack = np.stack(self.mags)
        eep = self.eep_grid if eep is None else np.asarray(eep, dtype=float)
        if eep is self.eep_grid or (eep.shape == self.eep_grid.shape
                                    and np.array_equal(eep, self.eep_grid)):
            if ks == list(range(ks[0], ks[0] + len(ks))):       # consecutive slices: a view
                base = self._stack[ks[0]:ks[0] + len(ks)]
            else:
                base = self._stack[ks]
            mini = self.mini
        else:
            base = np.empty((len(ks), eep.size, self.rvec.size))
            for n, k in enumerate(ks):
                for b in range(self.rvec.size):
                    base[n, :, b] = np.interp(eep, self.eep_grid, self.mags[k][:, b])
            mini = np.interp(eep, self.eep_grid, self.mini)
        # same order of additions as get_seds, in place in `out` or in a buffer this object
        # keeps (the caller uploads it before asking again)
        if out is not None:
            mag = out
        else:
            if getattr(self, "_buf", None) is None or self._buf.shape != base.shape:
                self._buf = np.empty(base.shape)
            mag = self._buf
        np.add(base, 0.25 * feh, out=mag)
        mag -= 0.15 * (loga - 9.)
        mag += (av * (self.rvec + 0.02 * (rv - 3.3)))[None, None, :]
        mag += 5. * np.log10(dist / 10.)
        low = mini < mini_bound
        if low.any():
            mag[:, low] = np.nan
        return mag, mini


def make_cluster(iso, nobj, seed=11, frac_no_parallax=0.3, frac_nan_band=0.05):
    """`nobj` members of the population `iso` at 850 pc with 3 % photometry,
    a few NaN bands and NaN parallaxes (what reference cluster.py expects:
    fluxes in maggies, parallaxes in mas)."""
    rng = np.random.RandomState(seed)
    eep = rng.uniform(230., 760., nobj)
    mag, _, _ = iso.get_seds(feh=-0.1, loga=9.6, av=0.2, rv=3.3, eep=eep, smf=0.,
                             dist=850.)
    flux = 10. ** (-0.4 * mag)
    err = 0.03 * flux
    phot = flux + rng.normal(size=flux.shape) * err
    hole = rng.uniform(size=phot.shape) < frac_nan_band
    hole[:, 0] = False
    phot[hole] = np.nan
    par = 1e3 / 850. + rng.normal(size=nobj) * 0.05
    perr = np.full(nobj, 0.05)
    par[rng.uniform(size=nobj) < frac_no_parallax] = np.nan
    return phot, err, par, perr
