"""cProfile of the host stage of `fit()` (a Python `lngalprior` hook: nothing of `lnpost` runs on the device)
on the bench's workload.    python tools/host_stage_profile.py [nstar=6]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

from brutus_amd import fitting, synth  # noqa: E402
from brutus_amd.galprior import gal_lnprior  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, n, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask)
bf.batch_size = n


def hook(dists, coord, labels=None):          # a user's own callable: no `device_params`
    return gal_lnprior(dists, coord, labels=labels)


def run():
    with tempfile.TemporaryDirectory() as tmp:
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=hook, rv_gauss=(3.32, 1e-6), rstate=np.random.RandomState(862), verbose=False)


run()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
print("%d objects in %.2f s = %.2f objects/s" % (n, time.perf_counter() - t0, n / (time.perf_counter() - t0)))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
