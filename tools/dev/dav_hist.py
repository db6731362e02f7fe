"""Development aid (CPU): sizes of the flux-phase steps dAv of the cull's survivors (numpy oracle)."""
import sys, os
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
from brutus_amd import synth
from oracle import brutus_oracle as O
models, _, _ = synth.make_mist_like_grid(60000, 12)
rec = []
orig = O.optimize_fit_flux
def hook(data, tot_var, rvecs, drvecs, av, rv, *a, **k):
    out = orig(data, tot_var, rvecs, drvecs, av, rv, *a, **k)
    rec.append(np.abs(out[0] - av))
    return out
O.optimize_fit_flux = hook
for cfg, kw, wp in ((2, dict(rvlim=(3.32, 3.32)), False), (3, dict(), True)):
    st = synth.make_stars(models, 12, seed=1 if cfg == 2 else 2, with_parallax=wp)
    its = {}
    for i in range(12):
        rec.clear()
        O.loglike(st["flux"][i], st["err"][i], st["mask"][i], models.astype(np.float64),
                  parallax=st["parallax"][i] if wp else None, parallax_err=st["parallax_err"][i] if wp else None, **kw)
        for k, d in enumerate(rec[:2]):
            its.setdefault(k, []).append(d)
    for k, v in its.items():
        d = np.concatenate(v)
        # a wave = 64 consecutive survivors: its largest step decides the path
        w = d[: d.size // 64 * 64].reshape(-1, 64).max(axis=1)
        print("cfg", cfg, "iteration", k + 1, "n", d.size, "quantiles 50/90/99/99.9/max", np.quantile(d, [.5, .9, .99, .999, 1.]).round(4),
              "| waves with max > 0.045: %.3f  > 0.11: %.3f  > 0.2: %.3f" % ((w > 0.045).mean(), (w > 0.11).mean(), (w > 0.2).mean()))
