"""One case of tools/fuzz_fit.py against the C restatement of the reference as the third opinion:
    python tools/dev/fuzz_case.py <seed> <case index> <star> <model>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from brutus_amd import fitting
import fuzz_fit as F, test_gpu_fit2 as T
from oracle import c_oracle
seed, ci, star, model = (int(x) for x in sys.argv[1:5])
rng = np.random.RandomState(seed)
for c in range(ci + 1):
    models, st, kw, with_par, tol, desc = F.case(rng)
print(desc)
if len(sys.argv) > 5 and sys.argv[5] == "solo":
    for k in ("flux", "err", "mask", "parallax", "parallax_err"):
        st[k] = st[k][star:star + 1]
    star = 0
S = st["flux"].shape[0]
par = st["parallax"] if with_par else np.full(S, np.nan)
perr = st["parallax_err"] if with_par else np.full(S, np.nan)
grid = fitting.DeviceGrid(models)
eng = fitting._Engine(grid, max_batch=S, mem_budget=200e9)
recs = eng.fit_batch(st["flux"], st["err"], st["mask"], par, perr, T._params(kw))
full = fitting.loglike_batch(st["flux"], st["err"], st["mask"], grid, avlim=kw.get("avlim", (0., 20.)),
    rvlim=kw.get("rvlim", (1., 8.)), rv_gauss=kw.get("rv_gauss", (3.32, 0.18)), dim_prior=kw.get("dim_prior", True),
    ltol=kw.get("ltol", 3e-2), parallax=par, parallax_err=perr, max_batch=min(S, 8))
tr = {}
okw = {k: kw[k] for k in kw}
lnl, Ndim, chi2, sc, av, rv, icov = c_oracle.loglike(st["flux"][star], st["err"][star], st["mask"][star], models,
                                                     parallax=par[star], parallax_err=perr[star], trace=tr, **okw)
rec = recs[star]
if model < 0:       # the selected model where the two device pipelines differ most in Av
    model = int(rec["sel"][int(np.argmax(np.abs(full["av"][star][rec["sel"]] - rec["av"])))])
    print("model", model)
pos = int(np.where(rec["sel"] == model)[0][0])
print("K1 hot %d full %d C %d   K2 hot %d full %d C %d" % (rec["K1"], full["k1"][star], tr["K1"], rec["K2"], full["k2"][star], tr["K2"]))
for k, cv in (("lnl", lnl), ("chi2", chi2), ("scale", sc), ("av", av), ("rv", rv)):
    h = rec["lnlike" if k == "lnl" else k][pos]; f = full[k][star][model]; cc = cv[model]
    print("%-6s hot %.15g  full %.15g  C %.15g   hot-C %.2e  full-C %.2e" % (k, h, f, cc, abs(h - cc) / max(abs(cc), 1), abs(f - cc) / max(abs(cc), 1)))
print("icov C", icov[model].ravel()[[0, 1, 2, 4, 5, 8]]); print("icov hot", rec["icov"][pos].ravel()[[0, 1, 2, 4, 5, 8]]); print("icov full", full["icov6"][:, star, model])
sel = rec["sel"]
e = np.abs(full["av"][star][sel] - rec["av"])
badpos = np.where(e > 1e-11)[0]
print("star %d: %d selected, %d off in av; positions in the list:" % (star, sel.size, badpos.size), badpos[:40], "models", sel[badpos[:40]])
print("errors", e[badpos[:20]])
# all stars: how many candidates are off
for i, r in enumerate(recs):
    ee = np.abs(full["av"][i][r["sel"]] - r["av"])
    n = int((ee > 1e-11).sum())
    if n:
        print("  star %d K2 %d: %d of %d off, max %.1e, first pos %d last pos %d" % (i, r["K2"], n, ee.size, ee.max(), np.where(ee > 1e-11)[0][0], np.where(ee > 1e-11)[0][-1]))
# Is the decision behind the difference a chaotic one?  The same star with its fluxes changed by
# parts in 1e15: a correct implementation's answers move by ~1e-15 x condition -- unless a
# comparison taken at rounding level (lnl_new < lnl_old -> step /= 1.2, fitting.py flux loop) flips.
f0 = st["flux"][star].copy()
for k, fac in enumerate((1. + 2e-15, 1. - 3e-13, 1. + 7e-12, 1. + 1e-10, 1. - 1e-9)):
    f = f0.copy(); f *= fac; f[k % f.size] *= fac
    o2 = c_oracle.loglike(f, st["err"][star], st["mask"][star], models, parallax=par[star], parallax_err=perr[star], **okw)
    d = np.abs(o2[4] - av)[rec["sel"]]
    print("C oracle, flux[%d] x %.0e: %d of %d selected models move by > 1e-10 in av (max %.1e at model %d); this model: %.1e"
          % (k, fac - 1., int((d > 1e-10).sum()), d.size, d.max(), int(rec["sel"][int(np.argmax(d))]), abs(o2[4][model] - av[model])))
