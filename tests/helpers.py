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
