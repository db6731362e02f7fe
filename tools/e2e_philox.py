import os, sys, tempfile, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
n = 2048
st = synth.make_stars(models, n, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128
for rep in range(2):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=PhiloxRandomState(862), verbose=False)
        print("philox: %.0f stars/s" % (n / (time.perf_counter() - t0)))
