"""Throughput of the 33 - 64 band route (full-grid pipeline + first cut on the device) on the bench's grid size.
    python tools/wide_bands_rate.py [nfilt=49] [nstar=16]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from brutus_amd import fitting, synth  # noqa: E402

nfilt = int(sys.argv[1]) if len(sys.argv) > 1 else 49
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
models, _, _ = synth.make_mist_like_grid(750000, nfilt)
st = synth.make_stars(models, S, seed=3)
grid = fitting.DeviceGrid(models, device="cuda:0")
eng = fitting._Engine(grid, max_batch=S, mem_budget=64e9)
params = fitting._make_params((0., 20.), (0., 1e6), (1., 8.), (3.32, 0.18), 3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
up = eng._upload(st["flux"], st["err"], st["mask"], st["parallax"], st["parallax_err"])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, ndim, k1, k2 = eng.fit_batch_device(*up, params)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d bands x 750k models: %d stars in %.2f s = %.1f stars/s, %d records, wide route %s"
          % (nfilt, S, dt, S / dt, int(rec.counts[0]), eng.wide), flush=True)
