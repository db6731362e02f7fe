"""cProfile of one end-to-end `fit()` in the device-lnpost mode (GPU box):
where the host time between the kernels goes.

    python tools/e2e_profile.py [--stars 1024] [--batch 128] [--config 2|3]
"""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stars", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--rng", choices=("philox", "numpy"), default="philox",
                    help="numpy: one sequential numpy RandomState (the reference's semantics)")
    ap.add_argument("--no-cprofile", action="store_true")
    a = ap.parse_args()
    models, labels, lmask = synth.make_mist_like_grid(750000, 12)
    with_par = a.config == 3
    st = synth.make_stars(models, a.stars, seed=4242, with_parallax=with_par)
    bf = fitting.BruteForce(models, labels, lmask)
    bf.batch_size = a.batch

    def run():
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            bf.fit(st["flux"], st["err"], st["mask"], np.arange(a.stars), os.path.join(tmp, "x"),
                   parallax=st["parallax"] if with_par else None,
                   parallax_err=st["parallax_err"] if with_par else None,
                   data_coords=st["coords"], lngalprior=gal_lnprior,
                   rv_gauss=(3.32, 1e-6) if a.config == 2 else (3.32, 0.18),
                   rstate=(PhiloxRandomState(862) if a.rng == "philox"
                           else np.random.RandomState(862)), verbose=False)
            return time.perf_counter() - t0

    run()
    dt = run()
    print("fit(): %.3f s for %d stars = %.0f stars/s" % (dt, a.stars, a.stars / dt))
    if a.no_cprofile:
        return
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(a.top)


if __name__ == "__main__":
    main()
