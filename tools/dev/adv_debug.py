"""Development aid: what the C oracle does with the adversarial stars of tests/test_gpu_fit2.py (CPU)."""
import sys, os
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
from brutus_amd import synth
from oracle import c_oracle
models, _, _ = synth.make_mist_like_grid(30000, 12, seed=17)
for frac in (1e-3, 1e-4):
    st = synth.make_stars(models, 8, seed=23, min_frac_err=frac)
    st["err"] = frac * np.abs(st["flux"])
    for i in range(8):
        tr = {}
        try:
            r = c_oracle.loglike(st["flux"][i], st["err"][i], st["mask"][i], models, parallax=st["parallax"][i],
                                 parallax_err=st["parallax_err"][i], trace=tr)
            print(frac, i, tr)
        except Exception as e:
            print(frac, i, "ERR", e)
