import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, 1024, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128; bf.scan_ahead = False
for rep in range(2):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(1024), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=PhiloxRandomState(862), verbose=False)
        print("fit %.1f ms/batch" % (1e3 * (time.perf_counter() - t0) / 8))
