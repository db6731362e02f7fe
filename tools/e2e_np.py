"""End-to-end fit with one sequential numpy RandomState (GPU box): stars/s."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
st = synth.make_stars(models, n, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128
for rep in range(2):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=np.random.RandomState(5), verbose=False)
        dt = time.perf_counter() - t0
    print("numpy stream: %.0f stars/s (%.1f ms per 128)" % (n / dt, 1e3 * dt / (n / 128)))
