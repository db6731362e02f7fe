"""Development aid: the end-to-end parity block of bench.py, verbose (which stage differs)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
from oracle import brutus_oracle as O, c_oracle

nmodel = int(sys.argv[1]) if len(sys.argv) > 1 else 750000
rvg = (3.32, float(sys.argv[2])) if len(sys.argv) > 2 else (3.32, 1e-6)
models, labels, lmask = synth.make_mist_like_grid(nmodel, 12)
st = synth.make_stars(models, 4096, seed=4242, with_parallax=False)
n = 8
bf = fitting.BruteForce(models, labels, lmask)
bf.batch_size = 128
lnprior = bf._setup(st["flux"][:n], st["err"][:n], st["mask"][:n], None, data_coords=st["coords"][:n],
                    lngalprior=gal_lnprior)[5]
lnp_o = O.static_lnprior(labels, lmask)
print("lnprior max diff", np.nanmax(np.abs(np.where(np.isfinite(lnprior), lnprior - lnp_o, 0))))
dev = list(bf._fit(st["flux"][:n], st["err"][:n], st["mask"][:n], data_coords=st["coords"][:n],
                   lngalprior=gal_lnprior, rv_gauss=rvg, lnprior=lnprior, Nmc_prior=50, Ndraws=250,
                   rstate=PhiloxRandomState(862)))
bf.device_lnpost = False
host = list(bf._fit(st["flux"][:n], st["err"][:n], st["mask"][:n], data_coords=st["coords"][:n],
                    lngalprior=gal_lnprior, rv_gauss=rvg, lnprior=lnprior, Nmc_prior=50, Ndraws=250,
                    rstate=PhiloxRandomState(862)))
ro = PhiloxRandomState(862)
for i in range(n):
    res = c_oracle.loglike(st["flux"][i], st["err"][i], st["mask"][i], models, parallax=np.nan,
                           parallax_err=np.nan, rv_gauss=rvg)
    ref = O.finish_star(res, lnprior, labels, st["coords"][i], np.nan, np.nan, ro, gal_lnprior,
                        Nmc_prior=50, Ndraws=250)
    print(i, "idx dev==ref", np.array_equal(dev[i][0], ref[0]), "host==ref", np.array_equal(host[i][0], ref[0]),
          "levid dev/host/ref", dev[i][7], host[i][7], ref[7], "chi2min", dev[i][8], host[i][8], ref[8])
