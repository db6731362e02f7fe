"""Development aid: K1 / K2 histogram and kernel times of the sharp-posterior workload."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brutus_amd import _lib, fitting, synth
L = _lib.lib()
models, _, _ = synth.make_sharp_grid(750000, 12)
grid = fitting.DeviceGrid(models)
st = synth.make_stars(models, 128, seed=2, with_parallax=True, frac_err=0.02, parallax_snr=10., frac_no_parallax=0.)
params = fitting._make_params((0., 20.), (0., 1e6), (1., 8.), (3.32, 0.18), 3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
eng = fitting._Engine(grid, max_batch=128, mem_budget=64e9)
up = eng._upload(st["flux"], st["err"], st["mask"], st["parallax"], st["parallax_err"])
for rep in range(2):
    L.brutus_enable_timing(1 if rep else 0)
    rec, ndim, k1, k2 = eng.fit_batch_device(*up, params)
    torch.cuda.synchronize()
print("K1", np.bincount(k1), "K2", np.bincount(k2), "nsel", np.diff(rec.off.cpu().numpy()).mean())
n = C.c_int(0); names = (C.c_char_p * 24)(); ms = (C.c_float * 24)()
L.brutus_last_timing(C.byref(n), names, ms, 24)
print({names[j].decode(): round(float(ms[j]), 3) for j in range(n.value)})
calls, rep = C.c_int64(0), C.c_int64(0)
L.brutus_debug_fit_stats(C.byref(calls), C.byref(rep)); print("calls", calls.value, "repeated host-driven", rep.value)
