"""fit() end to end at 128 and 256 objects per device batch, sharp and broad posteriors (GPU box).
Round 6: sharp 40.3 k / 33.8 k, broad 7.2 k / 7.0 k objects/s -- 128 stays the default."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
n = 8192
for kind in ("sharp", "broad"):
    if kind == "sharp":
        models, labels, lmask = synth.make_sharp_grid(750000, 12)
        st = synth.make_stars(models, n, seed=4243, with_parallax=True, **bench.SHARP_STARS)
    else:
        models, labels, lmask = synth.make_mist_like_grid(750000, 12)
        st = synth.make_stars(models, 2048, seed=4242, with_parallax=True)
    bf = fitting.BruteForce(models, labels, lmask)
    for bs in (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,256").split(",")):
        bf.batch_size = bs
        for rep in range(2):
            with tempfile.TemporaryDirectory() as tmp:
                m = st["flux"].shape[0]
                t0 = time.perf_counter()
                bf.fit(st["flux"], st["err"], st["mask"], np.arange(m), os.path.join(tmp, "e2e"),
                       parallax=st["parallax"], parallax_err=st["parallax_err"], data_coords=st["coords"],
                       lngalprior=gal_lnprior, verbose=False, rstate=PhiloxRandomState(862))
                dt = time.perf_counter() - t0
        print("%s batch %d: %.0f stars/s" % (kind, bs, m / dt), flush=True)
