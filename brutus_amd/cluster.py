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
import collections
import threading
import warnings

import numpy as np

from . import _lib

__all__ = ["isochrone_loglike"]

_DEFAULT_SMF = (0., 0.2, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8,
                0.85, 0.9, 0.95, 1.0)
_DEFAULT_SMF_ARR = np.asarray(_DEFAULT_SMF, float)
_DEFAULT_SMF_GRAD = np.gradient(_DEFAULT_SMF_ARR)


# Per-dataset terms (masks, chi-square outlier level, device copies of the photometry)
# and point tables are kept between calls: a sampler evaluates the same catalogue many
# thousand times, and revisits a point table whenever only dist / fout / offsets move.
_DATA_CACHE = collections.OrderedDict()
_TABLE_CACHE = collections.OrderedDict()
_DATA_CACHE_MAX, _TABLE_CACHE_MAX = 4, 16
# The cached entries own device / page-locked buffers that a call fills and reads: one
# evaluation at a time (a sampler's chains on one GPU take turns anyway).
_LOCK = threading.RLock()
# development aid: a list here receives (label, perf_counter) marks of a call (tools/dev/cluster_marks.py)
_TRACE = None


def _mark(label):
    if _TRACE is not None:
        import time
        _TRACE.append((label, time.perf_counter()))


def clear_caches():
    """Drop the cached per-dataset terms and isochrone point tables."""
    _DATA_CACHE.clear()
    _TABLE_CACHE.clear()


try:
    from xxhash import xxh3_64_intdigest as _digest
except ImportError:                                   # pragma: no cover
    from zlib import adler32 as _digest


def _fingerprint(a):
    """Address, layout and a digest of the CONTENT of an input array (an array edited in
    place is a different catalogue)."""
    if a is None:
        return None
    a = np.asarray(a)
    return (a.ctypes.data, a.shape, a.strides, _digest(np.ascontiguousarray(a).data))


def _lru_get(cache, key):
    val = cache.get(key)
    if val is not None:
        cache.move_to_end(key)
    return val


def _lru_put(cache, key, val, cap):
    cache[key] = val
    while len(cache) > cap:
        cache.popitem(last=False)


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
                      dim_prior=True, return_lnls=False, device=None, cache=True):
    """See `_isochrone_loglike`; calls are serialised (the caches own the buffers a call
    works in)."""
    with _LOCK:
        return _isochrone_loglike(theta, isochrone, phot, err, cluster_params, offsets,
                                  corr_params, mini_bound, eep_binary_max, smf_grid, eep_grid,
                                  parallax, parallax_err, cluster_prob, dim_prior, return_lnls,
                                  device, cache)


def _isochrone_loglike(theta, isochrone, phot, err, cluster_params, offsets, corr_params,
                       mini_bound, eep_binary_max, smf_grid, eep_grid, parallax, parallax_err,
                       cluster_prob, dim_prior, return_lnls, device, cache):
    """ln-likelihood of a co-eval stellar population.  Arguments, defaults and
    return value follow reference cluster.py:23-168: `theta` packs
    `(feh, loga, av, rv, dist[pc], fout)`, then per-band multiplicative offsets
    and four empirical-correction coefficients, each group only where it is
    declared free.  `isochrone` must provide
    `get_seds(feh=, loga=, av=, rv=, eep=, smf=, dist=, mini_bound=,
    eep_binary_max=, corr_params=) -> (seds (Neep, Nbands) mags, params, params2)`
    with `params['mini']` the initial-mass grid.

    Extensions: `device`; `cache` (default True) keeps the per-dataset terms and the
    isochrone point tables of recent calls (`clear_caches()` drops them).  A cached point
    table is keyed by the plug-in object, its `cache_token` attribute (if any) and every
    argument it was asked with: the plug-in is assumed deterministic, and one that is
    MODIFIED IN PLACE must change its `cache_token` (or the caller clears the caches / passes
    `cache=False`).  A plug-in that also offers
    `get_seds_grid(smf_grid=, ...same keywords...[, out=]) -> (seds (Nsmf, Neep, Nbands), mini)`
    is asked for a GROUP of consecutive mass fractions at a time (`smf_grid` = that group, 3
    growing groups per call by default, BRUTUS_CLUSTER_PIPELINE) instead of once per mass fraction; with
    `out=` it fills the page-locked buffer the device copy starts from.  Either way the device
    turns a group into fluxes and sums it while the plug-in works on the next one."""
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
    if smf_grid is None:            # (copies: the plug-in is handed slices of it)
        smf_grid, grad_smf = _DEFAULT_SMF_ARR.copy(), _DEFAULT_SMF_GRAD.copy()
    else:
        smf_grid = np.asarray(smf_grid, float)
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

    _mark("checked")
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

    _mark("theta")
    ds = _dataset(phot, err, parallax, parallax_err, dim_prior, device, cache)
    _mark("dataset")
    torch, dev = _torch(), ds.dev
    L = _lib.lib()

    # ---- per-object terms that move with theta (cluster.py:292-325) ---------------
    if ds.pmask is not None:
        chi2_p = (ds.par0 - 1e3 / dist) ** 2 * ds.ivar0
    else:
        chi2_p = np.zeros(Nobjs)
    ln_fin = np.log(cluster_prob * (1. - fout))
    ln_fout = np.log(1. - cluster_prob * (1. - fout))

    with torch.cuda.device(dev):
        up = lambda a, dt=np.float64: torch.from_numpy(
            np.ascontiguousarray(a, dtype=dt)).to(dev)
        stream = _stream_ptr(torch)
        ws, ws_n = ds.ws.data_ptr(), ds.ws.numel()
        objs = []

        def obj_args():
            """The objects' side of the kernel (cluster.py:292-333).  Prepared when the first
            group of isochrone points is already on its way to the device: the copy and the
            flux kernel of that group run under this."""
            if objs:
                return objs[0]
            t_ln = None
            if np.all(Xb == 1.):
                t_d, t_iv, t_ln = ds.t_d, ds.t_iv, ds.t_ln0
            else:                                  # multiplicative offsets (cluster.py:327-333)
                phot_t, err_t = phot * Xb, err * Xb
                with np.errstate(all="ignore"):
                    ivar = np.where(ds.phot_mask, 1. / err_t ** 2, 0.)
                    lnorm = np.nansum(np.log(2. * np.pi * err_t ** 2), axis=1) + ds.lnorm_p
                t_d, t_iv = up(np.where(ds.phot_mask, phot_t, 0.)), up(ivar)
                t_ln = up(lnorm)
            ds.h_cp.numpy()[...] = chi2_p
            t_cp = ds.t_cp.copy_(ds.h_cp, non_blocking=True)
            # (the tensors ride along: they must outlive the launches that read them)
            objs.append(((t_d.data_ptr(), t_iv.data_ptr(), t_cp.data_ptr(), t_ln.data_ptr(),
                          ds.t_n.data_ptr(), 1 if dim_prior else 0, ws, ws_n),
                         (t_d, t_iv, t_cp, t_ln)))
            _mark("objects up")
            return objs[0]

        def part(t_flux, t_lnw, npts, chunk_lo, chunk_n):
            """The sum over `npts` points as partials in chunks [chunk_lo, chunk_lo + chunk_n)."""
            _lib.check(L.brutus_cluster_lnl_part(
                Nobjs, Nbands, npts, t_flux.data_ptr() if npts else None,
                t_lnw.data_ptr() if npts else None, *obj_args()[0], chunk_lo, chunk_n, stream))

        def part_mags(t_src, t_mags, t_lng, t_lnsmf, neep, npts, chunk_lo, chunk_n):
            """The same straight from the plug-in's magnitudes (kept rows `t_src` of `t_mags`)."""
            _lib.check(L.brutus_cluster_lnl_part_mags(
                Nobjs, Nbands, npts, neep, t_src.data_ptr() if npts else None, t_mags.data_ptr(),
                t_lng.data_ptr(), t_lnsmf.data_ptr(), *obj_args()[0], chunk_lo, chunk_n, stream))

        # ---- isochrone points of every SMF slice (cluster.py:336-366) -------------
        tab, nchunk = _point_table(isochrone, feh, loga, av, rv, dist, corr_coef, smf_grid,
                                   grad_smf, eep_grid, mini_bound, eep_binary_max, Nbands, dev,
                                   torch, L, up, cache, part, part_mags)
        _mark("table")
        if tab is None:
            lnl = np.full(Nobjs, -np.inf)
            with np.errstate(all="ignore"):     # outlier mixture (cluster.py:410-414)
                lnl_mix = np.logaddexp(lnl + ln_fin, ds.lnl_outlier + ln_fout)
            lnl_tot = np.sum(lnl_mix)
        else:
            if nchunk:      # the pieces were summed while the plug-in worked on the next one
                _lib.check(L.brutus_cluster_lnl_merge(Nobjs, nchunk, ws, ws_n,
                                                      ds.out.data_ptr(), stream))
            else:           # a table met before
                t_flux, t_lnw = tab[1:]
                _lib.check(L.brutus_cluster_lnl(
                    Nobjs, Nbands, t_lnw.numel(), t_flux.data_ptr(), t_lnw.data_ptr(),
                    *obj_args()[0], ds.out.data_ptr(), stream))
            # ---- outlier mixture and total (cluster.py:410-414), on the device -------------
            _lib.check(L.brutus_cluster_mix(Nobjs, ds.out.data_ptr(), ds.t_lo.data_ptr(),
                                            float(ln_fin), float(ln_fout), ds.t_mix.data_ptr(),
                                            ds.t_mix[Nobjs:].data_ptr(), stream))
            if return_lnls:
                ds.h_mix.copy_(ds.t_mix, non_blocking=True)
            else:
                ds.h_mix[Nobjs:].copy_(ds.t_mix[Nobjs:], non_blocking=True)
            _mark("enqueued")
            torch.cuda.current_stream().synchronize()
            _mark("synced")
            h = ds.h_mix.numpy()
            lnl_tot = h[Nobjs]            # (a numpy scalar: a copy)
            lnl_mix = h[:Nobjs].copy() if return_lnls else None
    _mark("mixed")
    if return_lnls:
        return lnl_tot, lnl_mix
    return lnl_tot


class _Dataset(object):
    """What `isochrone_loglike` derives from the catalogue alone (cluster.py:292-325),
    with the device copies the kernel reads."""
    pass


def _dataset(phot, err, parallax, parallax_err, dim_prior, device, cache):
    from scipy.stats import chi2 as chisquare
    from .fitting import _torch
    def check():    # the reference's data check comes before anything is computed (cluster.py:296)
        if np.any(np.sum(np.isfinite(phot) & np.isfinite(err), axis=1) == 0):
            raise ValueError("At least one object has no valid data entries!")
    try:
        torch_ = _torch()
    except Exception:
        check()     # (no GPU: the argument error still comes first)
        raise
    # (the RESOLVED device: `device=None` follows the current device of each call)
    dev_key = str(torch_.device(device if device is not None
                                else "cuda:%d" % torch_.cuda.current_device()))
    key = (_fingerprint(phot), _fingerprint(err), _fingerprint(parallax),
           _fingerprint(parallax_err), bool(dim_prior), dev_key)
    if cache:
        ds = _lru_get(_DATA_CACHE, key)
        if ds is not None:          # (a cached catalogue passed the check below when it came in)
            return ds
    check()
    Nobjs = phot.shape[0]
    ds = _Dataset()
    ds.phot_mask = np.isfinite(phot) & np.isfinite(err)
    phot_n = np.sum(ds.phot_mask, axis=1)
    torch = torch_
    L = _lib.lib()
    dev = ds.dev = torch.device(device if device is not None
                                else "cuda:%d" % torch.cuda.current_device())
    ds.lnorm_p = np.zeros(Nobjs)
    ds.pmask = ds.par = ds.par_ivar = None
    if parallax is not None and parallax_err is not None:
        ds.par = np.asarray(parallax, dtype=np.float64)
        perr = np.asarray(parallax_err, dtype=np.float64)
        ds.pmask = np.isfinite(ds.par) & np.isfinite(perr)
        with np.errstate(all="ignore"):
            ds.par_ivar = 1. / perr ** 2
            # (zeros where there is no parallax: the per-call term needs no masked indexing)
            ds.par0 = np.where(ds.pmask, ds.par, 0.)
            ds.ivar0 = np.where(ds.pmask, ds.par_ivar, 0.)
            ds.lnorm_p[ds.pmask] = np.log(2. * np.pi * perr[ds.pmask] ** 2)
        phot_n = phot_n + ds.pmask
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        if dim_prior:
            ds.lnl_outlier = chisquare.logpdf(chisquare.ppf(1. - 1e-5, phot_n), phot_n)
        else:
            side = np.nanmax(phot + 3. * err, axis=0) - np.nanmin(phot - 3. * err, axis=0)
            frac = np.where(ds.phot_mask, 6. * err / side, 1.)
            vol = np.prod(frac, axis=1)
            if ds.pmask is not None:
                perr = np.asarray(parallax_err, dtype=np.float64)
                span = (np.nanmax((ds.par + 3. * perr)[ds.pmask])
                        - np.nanmin((ds.par - 3. * perr)[ds.pmask]))
                vol[ds.pmask] *= 6. * perr[ds.pmask] / span
            ds.lnl_outlier = np.log(1. / vol)
        ivar = np.where(ds.phot_mask, 1. / err ** 2, 0.)
        ds.lnorm0 = np.nansum(np.log(2. * np.pi * err ** 2), axis=1) + ds.lnorm_p
    with torch.cuda.device(dev):
        up = lambda a, dt=np.float64: torch.from_numpy(
            np.ascontiguousarray(a, dtype=dt)).to(dev)
        ds.t_d, ds.t_iv = up(np.where(ds.phot_mask, phot, 0.)), up(ivar)
        ds.t_n = up(phot_n, np.int32)
        ds.t_ln0 = up(ds.lnorm0)
        ds.h_cp = torch.empty(Nobjs, dtype=torch.float64).pin_memory()
        ds.t_cp = torch.empty(Nobjs, dtype=torch.float64, device=dev)
        ds.ws = torch.empty(L.brutus_cluster_workspace_bytes(Nobjs), dtype=torch.uint8, device=dev)
        ds.out = torch.empty(Nobjs, dtype=torch.float64, device=dev)
        ds.h_out = torch.empty(Nobjs, dtype=torch.float64).pin_memory()
        # outlier mixture and total on the device (cluster.py:410-414): what comes back is one
        # float64, and the per-object values only when they are asked for
        ds.t_lo = up(ds.lnl_outlier)
        ds.t_mix = torch.empty(Nobjs + 1, dtype=torch.float64, device=dev)      # [mix | total]
        ds.h_mix = torch.empty(Nobjs + 1, dtype=torch.float64).pin_memory()
    if cache:
        _lru_put(_DATA_CACHE, key, ds, _DATA_CACHE_MAX)
    return ds


_STAGE = {}


def _staging(nrow, nb, dev, torch):
    key = (nrow, nb, str(dev))
    st = _STAGE.get(key)
    if st is None:
        if len(_STAGE) > 4:
            _STAGE.clear()
        st = _Dataset()
        st.h_mags = torch.empty((nrow, nb), dtype=torch.float64).pin_memory()
        st.h_lnw = torch.empty(nrow, dtype=torch.float64).pin_memory()
        st.d_mags = torch.empty((nrow, nb), dtype=torch.float64, device=dev)
        st.d_lnw = torch.empty(nrow, dtype=torch.float64, device=dev)
        st.src = st.t_src = None
        st.hook_out = {}
        _STAGE[key] = st
    return st


# The plug-in is asked for a few secondary-mass-fraction slices at a time; while it works on
# the next group the device turns the previous one into fluxes and sums it (a group = one
# host -> device copy from page-locked memory and two launches, ~0.02 ms of host time, plus
# the plug-in's own per-call overhead, ~0.02 ms for the benchmark's table plug-in).  With that
# plug-in (0.27 ms for all 15 slices) the device chain (0.45 ms per call) is the longer one:
# what counts is that it starts early and never waits, so the groups GROW -- 3, 5, 7 slices of
# 15 -- and the objects' side of the kernel is prepared while the first group's copy runs.
# Whole calls per second, same box: 1 group 1 235, 2 groups 1 380, 3 groups 1 400-1 420,
# 4 groups 1 350 (equal halves before the growth: 1 300).  Marks of a call
# (tools/dev/cluster_marks.py, 3 groups): plug-in 0.10 + 0.10 + 0.13 ms, host waiting for the
# device at the end 0.19 ms.  A plug-in that takes milliseconds per slice (the MIST /
# neural-net isochrones) hides the device entirely with any split; BRUTUS_CLUSTER_PIPELINE
# sets the number of groups (1: one call to the plug-in and one sum, as before round 4).
# (Tried: the group's copy on a stream of its own, beside the previous group's sum.  Through
# torch the stream context and two events cost the host more than the 0.05 ms they free
# -- 1 370 -> 1 220; through a library call (hipMemcpyAsync on a side stream + event) no
# difference either way, 1 384-1 402 against 1 399-1 442: the kernels are the chain.)
_PIPELINE_GROUPS = 3
_PIPELINE_GROWTH = 1.5


def _group_bounds(nsmf, ngroup, growth=_PIPELINE_GROWTH):
    """Slice boundaries of `ngroup` consecutive groups whose sizes grow by `growth`."""
    ngroup = max(1, min(nsmf, ngroup, 64))
    w = np.cumsum(growth ** np.arange(ngroup))
    b = np.rint(nsmf * w / w[-1]).astype(int)
    b = np.maximum(b, np.arange(1, ngroup + 1))           # at least one slice per group
    b = np.minimum(b, nsmf - (ngroup - 1 - np.arange(ngroup)))
    return [0] + [int(x) for x in b]


def _point_table(isochrone, feh, loga, av, rv, dist, corr_coef, smf_grid, grad_smf, eep_grid,
                 mini_bound, eep_binary_max, Nbands, dev, torch, L, up, cache, part, part_mags):
    """Device-resident isochrone points `(flux (Npts, Nbands), lnw (Npts))` of all
    secondary-mass-fraction slices, or None if no slice has a usable point
    (cluster.py:336-366), and the number of partial-sum chunks filled on the way: a table
    met before comes back from the cache with 0 chunks (the caller sums it in one go), a new
    one is built group by group -- the plug-in's magnitudes go to the device as they are,
    `brutus_cluster_points` turns them into fluxes and drops the all-NaN points, `part`
    (brutus_cluster_lnl_part) sums the group into its share of the chunks."""
    import os
    from .fitting import _stream_ptr
    key = (id(isochrone), getattr(isochrone, "cache_token", None), feh, loga, av, rv, dist,
           None if corr_coef is None else tuple(corr_coef), smf_grid.tobytes(),
           eep_grid.tobytes(), mini_bound, eep_binary_max, str(dev))
    if cache:
        tab = _lru_get(_TABLE_CACHE, key)
        if tab is not None:
            tab = tab[0]
            if tab is not None and tab[0] == "mags":
                # first revisit of a table kept as magnitudes: make the flux table once (the sum
                # over fluxes is 3 % faster than the one that forms them on the way)
                _, t_src, t_mags, t_lng, t_lnsmf, neep_, npts = tab
                t_flux = torch.empty((npts, Nbands), dtype=torch.float64, device=dev)
                t_lnw = torch.empty(npts, dtype=torch.float64, device=dev)
                _lib.check(L.brutus_cluster_points_grid(
                    npts, Nbands, neep_, t_src.data_ptr(), t_mags.data_ptr(), t_lng.data_ptr(),
                    t_lnsmf.data_ptr(), t_flux.data_ptr(), t_lnw.data_ptr(), _stream_ptr(torch)))
                tab = ("flux", t_flux, t_lnw)
                _lru_put(_TABLE_CACHE, key, (tab, isochrone), _TABLE_CACHE_MAX)
            return tab, 0
    _mark("table key")
    kw = dict(feh=feh, loga=loga, av=av, rv=rv, eep=eep_grid, dist=dist,
              mini_bound=mini_bound, eep_binary_max=eep_binary_max, corr_params=corr_coef)
    nsmf, neep = len(smf_grid), len(eep_grid)
    nrow = nsmf * neep
    bounds = _group_bounds(nsmf, int(os.environ.get("BRUTUS_CLUSTER_PIPELINE", _PIPELINE_GROUPS)))
    ngroup = len(bounds) - 1
    nchunk = L.brutus_cluster_chunks()
    cbounds = [0]                                   # partial-sum chunks in proportion, >= 1 each
    for g in range(ngroup):
        cbounds.append(min(max(nchunk * bounds[g + 1] // nsmf, cbounds[-1] + 1),
                           nchunk - (ngroup - 1 - g)))
    stage = _staging(nrow, Nbands, dev, torch)
    h_mags = stage.h_mags.numpy().reshape(nsmf, neep, Nbands)
    h_lnw = stage.h_lnw.numpy().reshape(nsmf, neep)
    grid_hook = hasattr(isochrone, "get_seds_grid")
    if grid_hook and stage.hook_out.get(type(isochrone)) is None:   # does the hook take `out=`?
        import inspect
        stage.hook_out[type(isochrone)] = \
            "out" in inspect.signature(isochrone.get_seds_grid).parameters
    t_flux = t_lnw = None          # (flux table: only for per-slice mass grids, made on demand)

    def flux_table():
        if cache:      # a cached table keeps its tensors; without the cache one pair is reused
            return (torch.empty((nrow, Nbands), dtype=torch.float64, device=dev),
                    torch.empty(nrow, dtype=torch.float64, device=dev))
        if getattr(stage, "t_flux", None) is None:
            stage.t_flux = torch.empty((nrow, Nbands), dtype=torch.float64, device=dev)
            stage.t_lnw = torch.empty(nrow, dtype=torch.float64, device=dev)
        return stage.t_flux, stage.t_lnw
    if stage.src is None or len(stage.src) != ngroup or tuple(stage.h_lng.shape) != (ngroup, neep):
        stage.src, stage.t_src, stage.srckey = [None] * ngroup, [None] * ngroup, [None] * ngroup
        stage.h_lng = torch.empty((ngroup, neep), dtype=torch.float64).pin_memory()
        stage.d_lng = torch.empty((ngroup, neep), dtype=torch.float64, device=dev)
    smfkey = smf_grid.tobytes()
    if getattr(stage, "smfkey", None) != smfkey:
        with np.errstate(all="ignore"):
            stage.smfkey, stage.d_lnsmf = smfkey, up(np.log(grad_smf))
    d_lnsmf, mini0, d_lng, posb, pos = stage.d_lnsmf, None, None, None, None
    late = eep_grid > eep_binary_max                    # evolved stars: first slice only
    lateb = late.tobytes()
    ln_gsmf = np.log(grad_smf)
    stream = _stream_ptr(torch)
    off = shared = 0
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for g in range(ngroup):
            a, b = bounds[g], bounds[g + 1]
            c0, c1 = cbounds[g], cbounds[g + 1]
            if grid_hook:
                if stage.hook_out[type(isochrone)]:             # straight into pinned memory
                    mags, mini = isochrone.get_seds_grid(smf_grid=smf_grid[a:b], out=h_mags[a:b],
                                                         **kw)
                else:
                    mags, mini = isochrone.get_seds_grid(smf_grid=smf_grid[a:b], **kw)
                mags = np.asarray(mags, dtype=np.float64)
                mini = np.asarray(mini, dtype=np.float64)
                if not np.may_share_memory(mags, h_mags):
                    h_mags[a:b] = mags
            else:
                mini = np.empty((b - a, neep))
                for i in range(a, b):
                    seds, params, _ = isochrone.get_seds(smf=smf_grid[i], **kw)
                    h_mags[i] = seds
                    mini[i - a] = params['mini']
            _mark("plug-in %d" % g)
            # the group's host -> device copy starts now, from page-locked memory, and runs
            # under the host arithmetic below
            stage.d_mags[a * neep:b * neep].copy_(stage.h_mags[a * neep:b * neep], non_blocking=True)
            if mini.ndim == 1:
                # one mass grid for all slices: 2 000 logarithms per call, the (slice, EEP) table
                # of weights is formed on the device and the kept rows are looked up, not rebuilt
                if mini0 is not None and not (mini is mini0 or np.array_equal(mini, mini0)):
                    # (a "shared" grid is one grid: the table kept for a revisit holds ONE vector of
                    # ln(d mini) for all groups, and so does the reference, whose `mini` comes from
                    # the EEP / [Fe/H] / age of the isochrone alone, cluster.py:342-349)
                    raise RuntimeError("isochrone plug-in returned different 1-D mass grids for "
                                       "different groups of mass fractions; return a (slices, EEP) "
                                       "array if the grid depends on the slice")
                if mini0 is None:
                    mini0 = mini
                    gmini = np.gradient(mini)
                    pos = gmini > 0.
                    h_lng = stage.h_lng.numpy()[g]
                    h_lng[...] = -np.inf
                    np.log(gmini, out=h_lng, where=pos)
                    d_lng = stage.d_lng[g]
                    d_lng.copy_(stage.h_lng[g], non_blocking=True)
                    posb = pos.tobytes()
                srckey = (posb, a, b, lateb)
                if stage.srckey[g] != srckey:
                    keep = np.repeat(pos[None, :], b - a, axis=0)
                    keep[(1 if a == 0 else 0):, late] = False        # (slice 0 keeps its evolved stars)
                    src = (np.flatnonzero(keep) + a * neep).astype(np.int32)
                    stage.src[g], stage.t_src[g], stage.srckey[g] = src, up(src, np.int32), srckey
                n = stage.src[g].size
                if os.environ.get("BRUTUS_CLUSTER_MAGS", "1") != "0":
                    # fluxes and weights are formed by the sum itself, from the staged magnitudes
                    part_mags(stage.t_src[g], stage.d_mags, d_lng, d_lnsmf, neep, n, c0, c1 - c0)
                    _mark("group %d out" % g)
                    off += n
                    shared += 1
                    continue
                if t_flux is None:
                    t_flux, t_lnw = flux_table()
                if n:
                    _lib.check(L.brutus_cluster_points_grid(
                        n, Nbands, neep, stage.t_src[g].data_ptr(), stage.d_mags.data_ptr(),
                        d_lng.data_ptr(), d_lnsmf.data_ptr(), t_flux[off:].data_ptr(),
                        t_lnw[off:].data_ptr(), stream))
            else:
                gmini = np.gradient(mini, axis=1)
                keep = gmini > 0.
                lnw = np.where(keep, np.log(gmini) + ln_gsmf[a:b, None], -np.inf)
                first = 1 if a == 0 else 0                  # (slice 0 keeps its evolved stars)
                keep[first:, late] = False
                lnw[first:, late] = -np.inf
                src = (np.flatnonzero(keep) + a * neep).astype(np.int32)
                n = src.size
                if t_flux is None:
                    t_flux, t_lnw = flux_table()
                if n:
                    h_lnw[a:b] = lnw
                    stage.d_lnw[a * neep:b * neep].copy_(stage.h_lnw[a * neep:b * neep],
                                                         non_blocking=True)
                    if stage.src[g] is None or not np.array_equal(stage.src[g], src):
                        stage.src[g], stage.t_src[g] = src, up(src, np.int32)   # (rarely changes)
                    stage.srckey[g] = None
                    _lib.check(L.brutus_cluster_points(
                        n, Nbands, stage.t_src[g].data_ptr(), stage.d_mags.data_ptr(),
                        stage.d_lnw.data_ptr(), t_flux[off:].data_ptr(), t_lnw[off:].data_ptr(),
                        stream))
            part(t_flux[off:off + n], t_lnw[off:off + n], n, c0, c1 - c0)
            _mark("group %d out" % g)
            off += n
    if not off:
        tab = None
    elif shared == ngroup:
        # every group shared the one mass grid: the table IS the magnitudes + the kept rows.
        # Kept for a revisit as copies of the staging buffers (they are reused by the next call).
        tab = None
        if cache:
            srckey = tuple(stage.srckey)
            if getattr(stage, "src_all_key", None) != srckey:
                stage.src_all_key, stage.t_src_all = srckey, torch.cat(stage.t_src)
            # (d_lnsmf and t_src_all are replaced, never rewritten in place, when their keys
            # change -- stage.smfkey / stage.src_all_key above --, so the table may share them)
            tab = ("mags", stage.t_src_all, stage.d_mags.clone(), d_lng.clone(), d_lnsmf, neep, off)
        else:
            tab = ("mags",)                                  # (summed already; not kept)
    elif shared:
        raise RuntimeError("isochrone plug-in returned a shared mass grid for some groups of "
                           "mass fractions and per-slice grids for others")
    else:
        tab = ("flux", t_flux[:off], t_lnw[:off])
    if cache:      # (the plug-in object is kept alive with its tables: `id` stays unique)
        _lru_put(_TABLE_CACHE, key, (tab, isochrone), _TABLE_CACHE_MAX)
    return tab, nchunk
