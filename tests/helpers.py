"""Shared helpers for the test-suite (test infrastructure, not product)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def galprior(dists, coord, labels=None):
    """Analytic stand-in for the Galactic prior hook; identical to the one in
    tools/gen_golden.py that produced tests/golden/fit_synth.npz."""
    with np.errstate(all="ignore"):
        lp = 2. * np.log(dists) - dists / 2. + 0.01 * np.cos(np.deg2rad(coord[1]))
    if labels is not None:
        lp = lp + 0.1 * labels['feh']
    return lp


def loglike_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "loglike_*.npz")))


def load_loglike_case(path):
    z = np.load(path)
    kw = {}
    for k, v in zip(z["kw_keys"], z["kw_vals"]):
        v = v[np.isfinite(v)]
        if k == "dim_prior":
            kw[str(k)] = bool(v[0])
        elif v.size == 1:
            kw[str(k)] = float(v[0])
        else:
            kw[str(k)] = tuple(float(x) for x in v)
    if bool(z["parallax_is_none"]):
        par, perr = None, None
    else:
        par, perr = float(z["parallax"]), float(z["parallax_err"])
    return z, kw, par, perr


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(all="ignore"):
        r = np.abs(a - b) / np.maximum(np.abs(a), 1e-300)
    r = np.where(both_inf | both_nan, 0., r)
    r = np.where(np.isnan(r), np.inf, r)
    return float(np.max(r)) if r.size else 0.


class FakeIsochrone(object):
    """Duck-typed stand-in for `seds.Isochrone` (only `get_seds` is used by
    `isochrone_loglike`, reference cluster.py:339-344): a smooth synthetic
    isochrone whose SEDs depend on every argument, with non-existent points
    (NaN rows), a NaN band, and a non-monotonic-mass stretch."""

    def __init__(self, nbands=6):
        self.nbands = nbands

    def get_seds(self, feh=0., loga=9., av=0., rv=3.3, eep=None, smf=0.,
                 dist=1000., mini_bound=0.08, eep_binary_max=480.,
                 corr_params=None):
        x = (np.asarray(eep, float) - 202.) / 606.
        mini = 0.3 + 1.4 * x + 0.05 * np.sin(9. * x) * (x > 0.8)
        lam = np.linspace(0., 1., self.nbands)
        M = 9. - 8. * x + 0.3 * feh - 0.2 * (loga - 9.)
        col = (1. - (0.2 + 0.6 * x))[:, None] * (3.0 * (1. - lam) ** 1.5 - 0.8)[None, :]
        mag = M[:, None] + col + av * (1.2 - 1.0 * lam + 0.02 * (rv - 3.3))[None, :]
        mag = mag + 5. * np.log10(dist / 10.) - 10. + 10.   # distance modulus vs 1 kpc
        # unresolved binary: add a secondary of mass smf * mini (only on the MS)
        sec = -2.5 * np.log10(1. + np.where(np.asarray(eep) <= eep_binary_max,
                                            smf ** 3.5, 0.))
        mag = mag + sec[:, None]
        if corr_params is not None:
            mag = mag + 0.01 * corr_params[0] * (mini < 0.7)[:, None]
        mag[mini < mini_bound] = np.nan
        mag[(x > 0.93)] = np.nan            # non-existent evolved models
        mag[(x > 0.5) & (x < 0.52), 1] = np.nan   # one band missing for some points
        return mag, {"mini": mini}, {"mini": mini * smf}


def make_cluster_data(nobj, nbands, seed):
    rng = np.random.RandomState(seed)
    iso = FakeIsochrone(nbands)
    eep = rng.uniform(210., 740., nobj)
    mag, _, _ = iso.get_seds(feh=-0.1, loga=9.6, av=0.2, rv=3.3, eep=eep, smf=0.,
                             dist=850.)
    flux = 10. ** (-0.4 * mag)
    err = 0.03 * flux
    phot = flux + rng.normal(size=flux.shape) * err
    phot[rng.uniform(size=phot.shape) < 0.08] = np.nan
    bad = np.all(~np.isfinite(phot), axis=1)
    phot[bad, 0] = flux[bad, 0]
    par = 1e3 / 850. + rng.normal(size=nobj) * 0.05
    perr = np.full(nobj, 0.05)
    par[rng.uniform(size=nobj) < 0.3] = np.nan
    out = rng.uniform(size=nobj) < 0.05          # a few non-members
    phot[out] *= rng.uniform(0.3, 3., size=(out.sum(), 1))
    return iso, phot, err, par, perr
