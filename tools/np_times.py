import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brutus_amd import _lib, fitting, synth
from brutus_amd.galprior import gal_lnprior
L = _lib.lib()
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, 256, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128; bf.scan_ahead = False
for rep in range(2):
    L.brutus_enable_timing(rep)
    with tempfile.TemporaryDirectory() as tmp:
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(256), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=np.random.RandomState(3), verbose=False)
n = C.c_int(0); names = (C.c_char_p * 32)(); ms = (C.c_float * 32)()
L.brutus_last_timing(C.byref(n), names, ms, 32)
print(os.environ.get("BRUTUS_AMD_LIB", "in-tree").split("/")[-1], {names[i].decode(): round(ms[i], 2) for i in range(n.value)})
