"""Per-kernel HIP-event times of one end-to-end `fit()` batch in the device
lnpost mode (brutus_enable_timing / brutus_last_timing).  Runs on the GPU box:

    python tools/post_times.py [--config 2|3] [--stars 64]
"""
import argparse
import ctypes as C
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the HIP library: one HIP runtime per process)

from brutus_amd import _lib, fitting, synth
from brutus_amd.galprior import gal_lnprior
from brutus_amd.rng import PhiloxRandomState


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--stars", type=int, default=64)
    ap.add_argument("--tag", default="")
    ap.add_argument("--no-parallax", action="store_true")
    a = ap.parse_args()
    L = _lib.lib()
    models, labels, lmask = synth.make_mist_like_grid(750000, 12)
    st = synth.make_stars(models, 2 * a.stars, seed=4242)
    bf = fitting.BruteForce(models, labels, lmask)
    bf.batch_size = a.stars
    kw = dict(lngalprior=gal_lnprior, verbose=False)
    if a.config == 2:          # Av-only, as bench.py's end-to-end leg
        kw.update(rvlim=(3.32, 3.32), rv_gauss=(3.32, 1e-6))

    def run(lo, hi):
        with tempfile.TemporaryDirectory() as tmp:
            bf.fit(st["flux"][lo:hi], st["err"][lo:hi], st["mask"][lo:hi], np.arange(hi - lo),
                   os.path.join(tmp, "x"), parallax=None if a.no_parallax else st["parallax"][lo:hi],
                   parallax_err=None if a.no_parallax else st["parallax_err"][lo:hi], data_coords=st["coords"][lo:hi],
                   rstate=PhiloxRandomState(862), **kw)

    # count the records each stage sees: wrap the engine's post call
    counts = {}
    orig = fitting._Engine.post_batch_device

    def counted(self, rec, nstar, statics, coords, parallax, parallax_err, pp, **kw2):
        out = orig(self, rec, nstar, statics, coords, parallax, parallax_err, pp, **kw2)
        counts["first_cut"] = int(rec.off.cpu().numpy()[nstar])
        counts["second_cut"] = int(out[4][nstar] - out[4][0]) // (3 * pp.nmc)
        return out

    fitting._Engine.post_batch_device = counted
    run(0, a.stars)
    L.brutus_enable_timing(1)
    run(a.stars, 2 * a.stars)
    n = C.c_int(0)
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    L.brutus_last_timing(C.byref(n), names, ms, 32)
    print(a.tag, "records/batch", counts, {names[j].decode(): round(ms[j], 2) for j in range(n.value)})


if __name__ == "__main__":
    main()
