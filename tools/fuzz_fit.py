"""Randomised comparison of the hot path of brutus_fit_batch with the generic full-grid pipeline
(tests/test_gpu_fit2.py::_vs_full_grid: selected sets, K1, K2 identical, values to 1e-9, run-time
audit of the float32 bound) over shapes the fixed tests do not visit -- in particular star lists of
32 and more, which take the star-lane float32 pass (k_pre32s).  GPU box:

    python tools/fuzz_fit.py [cases] [seed]
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

from brutus_amd import fitting, synth
import test_gpu_fit2 as T


def case(rng):
    nb = int(rng.choice([5, 7, 8, 9, 11, 12, 12, 12, 13, 16, 24]))
    nmodel = int(rng.choice([300, 4099, 20000, 30000, 70001, 150000]))
    S = int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 100, 128, 130]))
    gridkind = rng.choice(["mist", "sharp", "random"])
    seed = int(rng.randint(1 << 30))
    if gridkind == "mist":
        models = synth.make_mist_like_grid(nmodel, nb, seed=seed)[0]
    elif gridkind == "sharp":
        models = synth.make_sharp_grid(nmodel, nb, seed=seed)[0]
    else:
        models = synth.make_grid(nmodel, nb, seed=seed)[0]
    skw = dict(seed=seed + 1, frac_no_parallax=float(rng.choice([0., 0.25, 1.])))
    mode = rng.choice(["default", "sn50", "sn200", "sn10"])
    if mode != "default":
        skw["frac_err"] = {"sn50": 0.02, "sn200": 0.005, "sn10": 0.1}[mode]
        skw["min_frac_err"] = min(0.02, skw["frac_err"])
    if rng.rand() < 0.4:
        skw["parallax_snr"] = float(rng.choice([3., 10., 50.]))
    st = synth.make_stars(models, S, **skw)
    # ragged masks, a few negative fluxes
    for i in range(S):
        if rng.rand() < 0.3:
            k = rng.randint(0, max(1, nb - 4))
            st["mask"][i, rng.choice(nb, size=k, replace=False)] = False
        if rng.rand() < 0.05:
            j = rng.randint(nb)
            st["flux"][i, j] = -abs(st["flux"][i, j]) * rng.rand()
    kw = {}
    r = rng.rand()
    if r < 0.35:
        kw["rvlim"] = (3.32, 3.32)
    elif r < 0.45:
        kw["rv_gauss"] = (3.32, float(rng.choice([1e-6, 0.5, 5.])))
    if rng.rand() < 0.2:
        kw["avlim"] = (0., float(rng.choice([0.8, 6., 100.])))
    if rng.rand() < 0.2:
        kw["ltol"] = float(rng.choice([3e-3, 1e-3, 0.3]))
    if rng.rand() < 0.2:
        kw["dim_prior"] = False
    with_par = rng.rand() < 0.7
    desc = dict(nb=nb, nmodel=nmodel, S=S, grid=str(gridkind), stars=skw, kw=kw, with_par=bool(with_par))
    tol = 1e-7 if mode == "sn200" else 1e-9
    return models, st, kw, with_par, tol, desc


def check(models, st, kw, with_par, tol):
    """The comparison of tests/test_gpu_fit2.py::_vs_full_grid with a metric fit for random cases:
    selected sets, K1, K2 and the float32 audit are hard requirements; values are compared as
    |a - b| / max(|a|, 1) (a log-likelihood may pass through zero) against `tol` (1e-8: a flux
    phase of a few hundred damped iterations along an Av-scale degeneracy separates two
    equivalent float64 pipelines by ~1e-9).  Returns (median selected, worst value error)."""
    S = st["flux"].shape[0]
    par = st["parallax"] if with_par else np.full(S, np.nan)
    perr = st["parallax_err"] if with_par else np.full(S, np.nan)
    grid = fitting.DeviceGrid(models)
    eng = fitting._Engine(grid, max_batch=S, mem_budget=200e9)
    with T._Env(BRUTUS_AUDIT=1):
        recs = eng.fit_batch(st["flux"], st["err"], st["mask"], par, perr, T._params(kw))
    aud, eps = T._audit(eng, grid.nmodel, grid.nfilt, S)
    assert np.all(aud.max(axis=0) < eps), ("audit", (aud.max(axis=0) / eps).max())
    full = fitting.loglike_batch(
        st["flux"], st["err"], st["mask"], grid, avlim=kw.get("avlim", (0., 20.)),
        rvlim=kw.get("rvlim", (1., 8.)), rv_gauss=kw.get("rv_gauss", (3.32, 0.18)),
        dim_prior=kw.get("dim_prior", True), ltol=kw.get("ltol", 3e-2), parallax=par,
        parallax_err=perr, max_batch=min(S, 8))
    worst = 0.
    for i, rec in enumerate(recs):
        sel = T._first_cut(full["lnl"][i], full["scale"][i], full["icov6"][0, i], par[i], perr[i])
        assert rec["K1"] == full["k1"][i] and rec["K2"] == full["k2"][i], \
            ("K", i, rec["K1"], full["k1"][i], rec["K2"], full["k2"][i])
        assert np.array_equal(sel, rec["sel"]), ("sel", i, sel.size, rec["sel"].size)
        for k in ("lnl", "chi2", "scale", "rv", "av"):
            a, b = full[k][i][sel], rec["lnlike" if k == "lnl" else k]
            e = np.abs(a - b) / np.maximum(np.abs(a), 1.)
            if e.size:
                worst = max(worst, float(e.max()))
                assert e.max() < tol, (k, i, float(e.max()), int(sel[int(np.argmax(e))]), rec["K2"])
        ic = full["icov6"][:, i, :][:, sel]
        d = np.sqrt(np.abs(ic[[0, 3, 5]]))
        for q, (a, b) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
            if sel.size:
                e = float(np.max(np.abs(rec["icov"][:, a, b] - ic[q]) / (d[a] * d[b])))
                worst = max(worst, e)
                assert e < tol, ("icov", q, i, e)
    return int(np.median([r["sel"].size for r in recs])), worst


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    bad = 0
    t0 = time.time()
    for c in range(n):
        models, st, kw, with_par, tol, desc = case(rng)
        try:
            nsel, worst = check(models, st, kw, with_par, 1e-8 if tol <= 1e-8 else tol)
            if os.environ.get("FUZZ_VERBOSE"):
                print("ok  %3d %s nsel~%d worst %.1e (%.0f s)" % (c, desc, nsel, worst, time.time() - t0), flush=True)
        except Exception:
            bad += 1
            print("BAD %3d %s" % (c, desc), flush=True)
            traceback.print_exc(limit=2)
    print("fuzz: %d cases, %d failures, seed %d" % (n, bad, seed))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
