"""How far is float32 from float64 on the adversarial shapes, and what breaks?  For every entry of
tests/test_gpu_fit2.py's ADVERSARIAL list and each float32 pass (tile kernel, star-lane vector,
star-lane MFMA): the run-time audit max|f32 - f64| / eps per statistic, K1 / K2 mismatches and
selected-set mismatches against the float64 full-grid pipeline.  Reports, never asserts.
    python tools/eps_survey.py [substring of the case names] [eps_scale]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fit2 as T  # noqa: E402

from brutus_amd import fitting  # noqa: E402

want = sys.argv[1] if len(sys.argv) > 1 else ""
eps_scale = sys.argv[2] if len(sys.argv) > 2 else "1.0"
FORMS = (("tile", dict(BRUTUS_PRE32_STAR_LANES_MIN=1000)),
         ("lanes", dict(BRUTUS_PRE32_STAR_LANES_MIN=1, BRUTUS_PRE32_MFMA=0)),
         ("mfma", dict(BRUTUS_PRE32_STAR_LANES_MIN=1, BRUTUS_PRE32_MFMA=1)))
print("%-30s %-6s %10s %10s %10s %10s  K1bad K2bad setbad  ncand/nsel" % ("case", "pass", "eps", "aud0/eps", "aud1/eps", "aud2/eps"))
for case in T.ADVERSARIAL:
    if want not in case[0]:
        continue
    models, st, kw, tol = T._adversarial_inputs(case)
    grid = fitting.DeviceGrid(models)
    S = st["flux"].shape[0]
    par, perr = st["parallax"], st["parallax_err"]
    full = fitting.loglike_batch(
        st["flux"], st["err"], st["mask"], grid, avlim=kw.get("avlim", (0., 20.)),
        rvlim=kw.get("rvlim", (1., 8.)), rv_gauss=kw.get("rv_gauss", (3.32, 0.18)),
        dim_prior=kw.get("dim_prior", True), ltol=kw.get("ltol", 3e-2), parallax=par,
        parallax_err=perr, max_batch=min(S, 8))
    sels = [T._first_cut(full["lnl"][i], full["scale"][i], full["icov6"][0, i], par[i], perr[i]) for i in range(S)]
    for name, env in FORMS:
        if name != "tile" and grid.nfilt > 12:
            continue
        eng = fitting._Engine(grid, max_batch=S, mem_budget=200e9)
        try:
            with T._Env(BRUTUS_AUDIT=1, BRUTUS_EPS_SCALE=eps_scale, **env):
                recs = eng.fit_batch(st["flux"], st["err"], st["mask"], par, perr, T._params(kw))
        except Exception as e:
            print("%-30s %-6s failed: %r" % (case[0], name, e))
            continue
        aud, eps = T._audit(eng, grid.nmodel, grid.nfilt, S)
        k1bad = sum(int(r["K1"] != full["k1"][i]) for i, r in enumerate(recs))
        k2bad = sum(int(r["K2"] != full["k2"][i]) for i, r in enumerate(recs))
        setbad = sum(int(not np.array_equal(sels[i], r["sel"])) for i, r in enumerate(recs))
        r = aud[:3] / eps[None, :]
        print("%-30s %-6s %10.3g %10.3g %10.3g %10.3g  %5d %5d %6d  %d" % (
            case[0], name, eps.max(), r[0].max(), r[1].max(), r[2].max(), k1bad, k2bad, setbad,
            int(np.mean([len(x) for x in sels]))), flush=True)
