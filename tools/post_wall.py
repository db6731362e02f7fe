"""Wall-clock split of the device lnpost call inside fit() (GPU box): the C call itself vs
the host work around it."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brutus_amd import _lib, fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
L = _lib.lib()
acc = {"c": 0., "n": 0, "pb": 0.}
orig_c = L.brutus_post_batch
def timed_c(*a):
    t = time.perf_counter(); r = orig_c(*a); acc["c"] += time.perf_counter() - t; acc["n"] += 1; return r
L.brutus_post_batch = timed_c
orig_pb = fitting._Engine.post_batch_device
def timed_pb(self, *a, **k):
    t = time.perf_counter(); r = orig_pb(self, *a, **k); acc["pb"] += time.perf_counter() - t; return r
fitting._Engine.post_batch_device = timed_pb
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, 2048, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128
for ahead in (False, True):
    bf.scan_ahead = ahead
    for rep in range(2):
        acc.update(c=0., n=0, pb=0.)
        L.brutus_enable_timing(1 if rep else 0)
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            bf.fit(st["flux"], st["err"], st["mask"], np.arange(2048), os.path.join(tmp, "x"), data_coords=st["coords"],
                   lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=PhiloxRandomState(862), verbose=False)
            dt = time.perf_counter() - t0
    print("scan_ahead %s: fit %.1f ms/batch; post_batch_device %.1f; C call %.1f (n=%d)" % (ahead, 1e3 * dt / 16, 1e3 * acc["pb"] / 16, 1e3 * acc["c"] / acc["n"], acc["n"]))
