"""Development aid: end-to-end fit() rate against the device batch size (GPU box)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
n = 2048
st = synth.make_stars(models, n, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask)
for mode in ("philox", "numpy"):
    for bs in (128, 192, 256):
        bf.batch_size = bs
        best = 1e9
        for rep in range(2):
            with tempfile.TemporaryDirectory() as tmp:
                rs = PhiloxRandomState(862) if mode == "philox" else np.random.RandomState(862)
                t0 = time.perf_counter()
                bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "x"), data_coords=st["coords"],
                       lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=rs, verbose=False)
                best = min(best, time.perf_counter() - t0)
        print(mode, "batch", bs, "%.0f stars/s" % (n / best), flush=True)
