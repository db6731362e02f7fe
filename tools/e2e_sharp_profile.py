"""cProfile of the end-to-end fit() on the sharp-posterior workload (GPU box): where the time between scan rate
(93 k stars/s) and fit() rate (24 - 26 k) goes.    python tools/e2e_sharp_profile.py [stars=8192]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

import bench
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
models, labels, lmask = synth.make_sharp_grid(750000, 12)
bf = fitting.BruteForce(models, labels, lmask)
bf.batch_size = 128
st = synth.make_stars(models, n, seed=4243, with_parallax=True, **bench.SHARP_STARS)


def run():
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "e2e"),
               parallax=st["parallax"], parallax_err=st["parallax_err"], data_coords=st["coords"],
               lngalprior=gal_lnprior, verbose=False, rstate=PhiloxRandomState(862))
        return time.perf_counter() - t0


run()
dt = run()
print("fit(): %.3f s for %d stars = %.0f stars/s" % (dt, n, n / dt))
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
