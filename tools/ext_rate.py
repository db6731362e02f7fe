"""fit() with external label constraints (`lnprior_ext`) on the bench's grid: objects/s.
    python tools/ext_rate.py [nstar=256] [device: 1|0]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

from brutus_amd import fitting, synth  # noqa: E402
from brutus_amd.galprior import gal_lnprior  # noqa: E402
from brutus_amd.rng import PhiloxRandomState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
device = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, n, seed=4242)
ext = {"feh": np.stack([labels["feh"][st["true_idx"]] + 0.05, np.full(n, 0.15)], axis=1)}
bf = fitting.BruteForce(models, labels, lmask)
bf.device_lnpost = device
for rep in range(2):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "x"), parallax=st["parallax"],
               parallax_err=st["parallax_err"], data_coords=st["coords"], lngalprior=gal_lnprior,
               lnprior_ext=ext, rstate=PhiloxRandomState(862), verbose=False)
        dt = time.perf_counter() - t0
        print("lnprior_ext, device route %s: %d objects in %.2f s = %.1f objects/s" % (device, n, dt, n / dt), flush=True)
