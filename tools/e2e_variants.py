"""End-to-end fit() rate under a few host-side settings (GPU box):
    python tools/e2e_variants.py [--stars 2048] [--rng philox|numpy]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState

ap = argparse.ArgumentParser()
ap.add_argument("--stars", type=int, default=2048)
ap.add_argument("--rng", default="philox")
a = ap.parse_args()
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, a.stars, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask)


def run(batch, ahead, pipe=True):
    bf.batch_size, bf.scan_ahead, bf.post_pipeline = batch, ahead, pipe
    best = None
    for rep in range(3):
        rs = PhiloxRandomState(862) if a.rng == "philox" else np.random.RandomState(862)
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            bf.fit(st["flux"], st["err"], st["mask"], np.arange(a.stars), os.path.join(tmp, "x"),
                   data_coords=st["coords"], lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6),
                   rstate=rs, verbose=False)
            dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("batch %3d scan_ahead %-5s pipeline %-5s: %6.0f stars/s" % (batch, ahead, pipe, a.stars / best))


for batch in (64, 128, 256):
    for ahead in (True, False):
        run(batch, ahead)
if a.rng == "numpy":
    run(128, True, False)
