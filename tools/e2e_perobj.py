"""End-to-end `_fit` with RandomState(seed0 + i) per object (GPU box): stars/s."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brutus_amd import fitting, h5io, synth
from brutus_amd.galprior import gal_lnprior
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
st = synth.make_stars(models, n, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128
lnprior = bf._setup(st["flux"], st["err"], st["mask"], None, data_coords=st["coords"], lngalprior=gal_lnprior)[5]
for rep in range(3):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        out = h5io.ResultsFile(os.path.join(tmp, "e2e.h5"), n, 250, np.arange(n), True)
        gen = bf._fit(st["flux"], st["err"], st["mask"], data_coords=st["coords"], lngalprior=gal_lnprior,
                      rv_gauss=(3.32, 1e-6), lnprior=lnprior, Nmc_prior=50, Ndraws=250, seed0=862)
        for i, row in enumerate(gen):
            out.write_row(i, row)
        out.close()
        dt = time.perf_counter() - t0
    print("%s per-object numpy streams: %.0f stars/s (%.1f ms per 128)" % (os.environ.get("BRUTUS_AMD_LIB", "in-tree").split("/")[-1], n / dt, 1e3 * dt / (n / 128)))
