"""Development aid: the sharp-posterior workload object by object (which one breaks what)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState
nmodel = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
models, labels, lmask = synth.make_sharp_grid(nmodel, 12)
bf = fitting.BruteForce(models, labels, lmask)
st = synth.make_stars(models, n, seed=4243, with_parallax=True, frac_err=0.02, parallax_snr=10., frac_no_parallax=0.)
# scan alone: selected models per object
eng = bf._engine(None)
params = fitting._make_params((0., 20.), (0., 1e6), (1., 8.), (3.32, 0.18), 3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
recs = eng.fit_batch(st["flux"], st["err"], st["mask"], st["parallax"], st["parallax_err"], params)
rec = recs[0]
off = rec.off.cpu().numpy()
print("selected per object:", np.diff(off)[:64], flush=True)
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bf.batch_size = bs
for a in range(0, n, bs):
    sl = slice(a, a + bs)
    g = bf._fit(st["flux"][sl], st["err"][sl], st["mask"][sl], parallax=st["parallax"][sl],
                parallax_err=st["parallax_err"][sl], data_coords=st["coords"][sl], lngalprior=gal_lnprior,
                Nmc_prior=50, Ndraws=250, rstate=PhiloxRandomState(862))
    rows = list(g)
    torch.cuda.synchronize()
    print("objects", a, a + bs, "ok", [int(r[5]) for r in rows], flush=True)
