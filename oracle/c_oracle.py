"""ctypes wrapper of oracle/loglike_ref.c (libbrutus_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's `cpu_baseline`
leg.  Never imported by the product package.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libbrutus_ref.so")
_lib = None


class RefParams(C.Structure):
    _fields_ = [("avlim", C.c_double * 2), ("av_gauss", C.c_double * 2),
                ("rvlim", C.c_double * 2), ("rv_gauss", C.c_double * 2),
                ("ltol", C.c_double), ("ltol_subthresh", C.c_double),
                ("init_thresh", C.c_double), ("dim_prior", C.c_int32),
                ("max_iter", C.c_int32)]


def available():
    return os.path.exists(LIB)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.brutus_ref_num_threads.restype = C.c_int
        _lib.brutus_ref_loglike.restype = C.c_int
    return _lib


def num_threads():
    return int(_load().brutus_ref_num_threads())


def loglike(data, data_err, data_mask, mag_coeffs, avlim=(0., 20.),
            av_gauss=(0., 1e6), rvlim=(1., 8.), rv_gauss=(3.32, 0.18),
            dim_prior=True, ltol=3e-2, ltol_subthresh=1e-2, init_thresh=5e-3,
            parallax=None, parallax_err=None, trace=None):
    """Same call/return convention as `brutus_oracle.loglike(return_vals=True)`."""
    L = _load()
    models = np.ascontiguousarray(mag_coeffs, dtype=np.float32)
    nmodel, nfilt, _ = models.shape
    flux = np.ascontiguousarray(data, dtype=np.float64)
    err = np.ascontiguousarray(data_err, dtype=np.float64)
    mask = np.ascontiguousarray(np.asarray(data_mask).astype(np.uint8))
    if av_gauss is None:
        av_gauss = (0., 1e6)
    p = RefParams()
    p.avlim[:] = list(map(float, avlim))
    p.av_gauss[:] = list(map(float, av_gauss))
    p.rvlim[:] = list(map(float, rvlim))
    p.rv_gauss[:] = list(map(float, rv_gauss))
    p.ltol, p.ltol_subthresh, p.init_thresh = ltol, ltol_subthresh, init_thresh
    p.dim_prior = 1 if dim_prior else 0
    p.max_iter = 0
    has_par = parallax is not None and parallax_err is not None
    out = {k: np.empty(nmodel) for k in ("lnl", "chi2", "scale", "av", "rv")}
    icov = np.empty((nmodel, 3, 3))
    ndim, k1, k2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    nsel = C.c_int64(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.brutus_ref_loglike(
        vp(models), C.c_int64(nmodel), C.c_int(nfilt), vp(flux), vp(err), vp(mask),
        C.c_double(parallax if has_par else np.nan),
        C.c_double(parallax_err if has_par else np.nan), C.c_int(1 if has_par else 0),
        C.byref(p), vp(out["lnl"]), vp(out["chi2"]), vp(out["scale"]), vp(out["av"]),
        vp(out["rv"]), vp(icov), C.byref(ndim), C.byref(k1), C.byref(k2),
        C.byref(nsel))
    if rc != 0:
        raise RuntimeError("brutus_ref_loglike failed: %d" % rc)
    if trace is not None:
        trace["K1"], trace["K2"], trace["nsel"] = k1.value, k2.value, nsel.value
    return (out["lnl"], int(ndim.value), out["chi2"], out["scale"], out["av"],
            out["rv"], icov)


def _one(args):
    flux, err, mask, models, par, pe, kw, serial = args
    if serial:
        _load().brutus_ref_set_threads(1)
    if par is not None and not np.isfinite(par):
        par, pe = None, None
    return loglike(flux, err, mask, models, parallax=par, parallax_err=pe, **kw)


def loglike_many(flux, err, mask, models, parallax, parallax_err, threads=1, **kw):
    """Many stars.  `threads == 1`: one star after the other, OpenMP over models
    inside each; `threads > 1`: a pool of host threads, one serial star each
    (ctypes releases the GIL during the C call)."""
    models = np.ascontiguousarray(models, dtype=np.float32)
    jobs = [(flux[i], err[i], mask[i], models,
             None if parallax is None else parallax[i],
             None if parallax_err is None else parallax_err[i], kw, threads > 1)
            for i in range(len(flux))]
    if threads <= 1:
        return [_one(j) for j in jobs]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as pool:
        return list(pool.map(_one, jobs))
