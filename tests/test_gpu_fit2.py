"""Second-generation hot path of brutus_fit_batch (float32 pre-classification +
exact thresholds + float64 on candidates, csrc/fit2_kernels.hpp) against the
first-generation path (float64 on every model), which the other GPU tests pin to
the oracle and the reference goldens.  The two must emit IDENTICAL record sets;
values may differ in the last bits only where a different but equivalent
exponential routine is used.  Also checks the run-time audit of the float32
error bound `eps` and the K1 logic (float32 decision / exact probe / deep probe).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env(object):
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _both_paths(eng, st, params, with_par=True):
    out = {}
    par = st["parallax"] if with_par else None
    perr = st["parallax_err"] if with_par else None
    for path in (1, 2):
        with _Env(BRUTUS_FIT_PATH=path, BRUTUS_AUDIT=1):
            out[path] = eng.fit_batch(st["flux"], st["err"], st["mask"], par, perr, params)
    return out


def _audit(eng, nmodel, nfilt, S):
    import torch
    from brutus_amd import _lib
    L = _lib.lib()
    ws = eng._workspace(S)
    aud = torch.empty((4, S), dtype=torch.float32, device=ws.device)
    n32 = L.brutus_debug_sizeof_star32() // 4
    s32 = torch.empty((S, n32), dtype=torch.float32, device=ws.device)
    for which, t in ((4, aud), (5, s32)):
        _lib.check(L.brutus_debug_copy(ws.data_ptr(), ws.numel(), nmodel, nfilt, S, which,
                                       t.data_ptr(), t.numel() * 4, None))
    torch.cuda.synchronize()
    return aud.cpu().numpy(), s32[:, 4 * 32 + 9].cpu().numpy()


def _assert_same(r1, r2, tag, exact=True):
    for i, (a, b) in enumerate(zip(r1, r2)):
        assert a["K1"] == b["K1"] and a["K2"] == b["K2"], (tag, i, a["K1"], b["K1"], a["K2"], b["K2"])
        assert np.array_equal(a["sel"], b["sel"]), (tag, i, a["sel"].size, b["sel"].size)
        for k in ("lnlike", "chi2", "scale", "av", "rv", "icov"):
            if exact:
                assert np.array_equal(a[k], b[k]), (tag, i, k)
            else:
                d = np.abs(a[k] - b[k]) / np.maximum(np.abs(a[k]), 1e-12)
                assert d.max() < 1e-9, (tag, i, k, d.max())


@pytest.mark.parametrize("kw", [dict(), dict(rvlim=(3.32, 3.32)), dict(ltol=3e-3),
                                dict(dim_prior=False), dict(avlim=(0., 0.8))],
                         ids=["default", "rv_pinned", "ltol", "no_dim_prior", "avlim"])
def test_paths_agree_small_grid_edge_cases(kw):
    """30k x 8 lattice grid, 24 stars incl. a negative flux (large K2), masked
    bands, NaN parallaxes; K1 = 1, 2 and > 2 all occur."""
    from brutus_amd import fitting, synth
    models, _, _ = synth.make_mist_like_grid(30000, 8, seed=3)
    st = synth.make_stars(models, 24, seed=21)
    st["flux"][3, 2] = -abs(st["flux"][3, 2])
    st["mask"][5, [1, 6]] = False
    grid = fitting.DeviceGrid(models)
    eng = fitting._Engine(grid, max_batch=24)
    params = fitting._make_params(
        kw.get("avlim", (0., 20.)), (0., 1e6), kw.get("rvlim", (1., 8.)), (3.32, 0.18),
        kw.get("ltol", 3e-2), 1e-2, 5e-3, kw.get("dim_prior", True), wt_thresh=1e-3)
    r = _both_paths(eng, st, params)
    # general-mode survivors: path 1 evaluates 10^x with the table-free polynomial in
    # its flux kernel and so does path 2 (same kernel) -> identical bits everywhere
    _assert_same(r[1], r[2], kw)
    aud, eps = _audit(eng, 30000, 8, 24)
    assert np.all(aud.max(axis=0) < 0.25 * eps), (aud.max(axis=0) / eps).max()


@pytest.mark.parametrize("config", [2, 3])
def test_paths_agree_full_size(config):
    """BASELINE configs[1] / configs[2] at 750k x 12, the bench's grid and stars."""
    from brutus_amd import fitting, synth
    models, _, _ = synth.make_mist_like_grid(750000, 12)
    S = 24
    with_par = config == 3
    st = synth.make_stars(models, S, seed=77 + config, with_parallax=with_par)
    grid = fitting.DeviceGrid(models)
    eng = fitting._Engine(grid, max_batch=S, mem_budget=100e9)
    rvlim = (3.32, 3.32) if config == 2 else (1., 8.)
    params = fitting._make_params((0., 20.), (0., 1e6), rvlim, (3.32, 0.18),
                                  3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    r = _both_paths(eng, st, params, with_par)
    _assert_same(r[1], r[2], config)
    aud, eps = _audit(eng, 750000, 12, S)
    assert np.all(aud.max(axis=0) < 0.25 * eps), (aud.max(axis=0) / eps).max()


def test_high_signal_to_noise_and_sharp_posteriors():
    """float32 is weakest at high S/N (the chi2 cancels against sum (S/N)^2): stars
    with 0.2 % photometry and precise parallaxes -> eps grows with them, the record
    sets stay identical and the audit stays inside eps."""
    from brutus_amd import fitting, synth
    models, _, _ = synth.make_mist_like_grid(200000, 12, seed=5)
    S = 16
    st = synth.make_stars(models, S, seed=9, min_frac_err=0.002)
    st["err"] = np.minimum(st["err"], 0.004 * np.abs(st["flux"]))
    st["parallax_err"] = np.where(np.isfinite(st["parallax_err"]),
                                  np.minimum(st["parallax_err"], 0.02), np.nan)
    grid = fitting.DeviceGrid(models)
    eng = fitting._Engine(grid, max_batch=S)
    for rvlim in ((1., 8.), (3.32, 3.32)):
        params = fitting._make_params((0., 20.), (0., 1e6), rvlim, (3.32, 0.18),
                                      3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
        r = _both_paths(eng, st, params)
        _assert_same(r[1], r[2], rvlim)
        aud, eps = _audit(eng, 200000, 12, S)
        assert np.all(aud.max(axis=0) < 0.5 * eps), (aud.max(axis=0) / eps).max()


def test_random_order_grid_and_tiny_shapes():
    """A grid without any index locality, and shapes around the tile / chunk sizes."""
    from brutus_amd import fitting, synth
    for nmodel, nfilt, nstar in ((1, 4, 1), (255, 5, 3), (257, 8, 2), (4099, 12, 5), (70001, 6, 7)):
        models, _, _ = synth.make_grid(nmodel, nfilt, seed=nmodel)
        st = synth.make_stars(models, nstar, seed=nmodel + 1)
        grid = fitting.DeviceGrid(models)
        eng = fitting._Engine(grid, max_batch=nstar)
        for rvlim in ((1., 8.), (3.32, 3.32)):
            params = fitting._make_params((0., 20.), (0., 1e6), rvlim, (3.32, 0.18),
                                          3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
            r = _both_paths(eng, st, params)
            _assert_same(r[1], r[2], (nmodel, nfilt, rvlim))


def test_many_sweeps_star_is_not_an_error():
    """A star whose magnitude phase needs many sweeps (strong Av-Rv degeneracy under a
    wide Rv prior) must neither abort the batch nor change any other star: K1 equals
    the C oracle's, records equal the oracle's (reference fitting.py:173-264 has no
    sweep cap)."""
    from brutus_amd import fitting, synth
    from brutus_amd.pdf import scale_parallax_lnprior
    from oracle import c_oracle
    models, _, _ = synth.make_mist_like_grid(40000, 12, seed=8)
    st = synth.make_stars(models, 6, seed=31)
    kw = dict(rv_gauss=(3.32, 5.), ltol=3e-3)
    grid = fitting.DeviceGrid(models)
    eng = fitting._Engine(grid, max_batch=6)
    params = fitting._make_params((0., 20.), (0., 1e6), (1., 8.), kw["rv_gauss"],
                                  kw["ltol"], 1e-2, 5e-3, True, wt_thresh=1e-3)
    k1s = []
    for path in (1, 2):
        with _Env(BRUTUS_FIT_PATH=path):
            recs = eng.fit_batch(st["flux"], st["err"], st["mask"], st["parallax"],
                                 st["parallax_err"], params)
        for i, rec in enumerate(recs):
            tr = {}
            lnl, nd, chi2, sc, av, rv, icov = c_oracle.loglike(
                st["flux"][i], st["err"][i], st["mask"][i], models, parallax=st["parallax"][i],
                parallax_err=st["parallax_err"][i], trace=tr, **kw)
            with np.errstate(all="ignore"):
                lnprob = lnl + scale_parallax_lnprior(
                    sc, 1. / np.sqrt(np.abs(icov[:, 0, 0])), st["parallax"][i],
                    st["parallax_err"][i])
            lnprob = np.where(np.isfinite(lnprob), lnprob, -1e300)
            sel = np.where(lnprob > np.log(1e-3) + lnprob.max())[0]
            assert rec["K1"] == tr["K1"] and rec["K2"] == tr["K2"], (path, i, rec["K1"], tr["K1"])
            assert np.array_equal(sel, rec["sel"]), (path, i)
            assert np.max(np.abs(lnl[sel] - rec["lnlike"])) < 1e-7
            k1s.append(tr["K1"])
    assert max(k1s) > 8, k1s      # the case the old 8-sweep cap rejected


def test_record_buffer_regrowth_replays_select_and_emit():
    """Record buffers too small for a batch: `records_device` grows them and
    `brutus_fit_gather` replays selection + emit from the workspace (survivor tags in the
    float32 plane, staged flux-phase results, candidate offsets) without redoing the scan;
    the records equal those of a run whose buffers were large enough from the start."""
    import torch
    from brutus_amd import fitting, synth
    models, _, _ = synth.make_mist_like_grid(60000, 12, seed=4)
    st = synth.make_stars(models, 12, seed=5)
    grid = fitting.DeviceGrid(models)
    params = fitting._make_params((0., 20.), (0., 1e6), (1., 8.), (3.32, 0.18),
                                  3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    ref = fitting._Engine(grid, max_batch=12).fit_batch(
        st["flux"], st["err"], st["mask"], st["parallax"], st["parallax_err"], params)
    eng = fitting._Engine(grid, max_batch=12)
    up = eng._upload(st["flux"], st["err"], st["mask"], st["parallax"], st["parallax_err"])
    eng._sel_bufs = (torch.empty(1000, dtype=torch.int32, device=grid.device),
                     torch.empty((11, 1000), dtype=torch.float64, device=grid.device))
    sel_idx, sel_vals, sel_off, off, ndim, k1, k2 = eng.records_device(*up, params)
    assert sel_idx.numel() > 1000 and int(off[-1]) == sum(len(r["sel"]) for r in ref)
    for s, r in enumerate(ref):
        got = eng.record_of(sel_idx, sel_vals, off, s, ndim[s], k1[s], k2[s])
        assert np.array_equal(got["sel"], r["sel"]), s
        for k in ("lnlike", "chi2", "scale", "av", "rv", "icov"):
            assert np.array_equal(got[k], r[k]), (s, k)
