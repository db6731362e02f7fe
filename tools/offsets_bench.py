"""`photometric_offsets` at the size SURVEY 8f quotes (23k objects x 250 draws x 12 bands):
device form (full Nmc) next to the host form (a few rounds, scaled).  Runs on the GPU box:

    python tools/offsets_bench.py [--nobj 23000] [--nmc 150] [--host-nmc 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from brutus_amd import synth, utils


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nobj", type=int, default=23000)
    ap.add_argument("--nsamps", type=int, default=250)
    ap.add_argument("--nmc", type=int, default=150)
    ap.add_argument("--host-nmc", type=int, default=3)
    a = ap.parse_args()
    models, _, _ = synth.make_mist_like_grid(750000, 12)
    rng = np.random.RandomState(1)
    No, Ns = a.nobj, a.nsamps
    idxs = rng.randint(0, models.shape[0], (No, Ns))
    reds, dreds = rng.uniform(0, 1, (No, Ns)), rng.uniform(3, 3.6, (No, Ns))
    dists = rng.uniform(0.5, 3, (No, Ns))
    s0 = utils.get_seds(models[idxs[:, 0]], av=reds[:, 0], rv=dreds[:, 0],
                        return_flux=True) / dists[:, 0, None] ** 2
    phot = s0 * (1 + 0.05 * rng.normal(size=s0.shape))
    err = 0.05 * phot
    mask = rng.uniform(size=phot.shape) > 0.1
    kw = dict(verbose=False)
    utils.photometric_offsets(phot[:64], err[:64], mask[:64], models, idxs[:64], reds[:64],
                              dreds[:64], dists[:64], Nmc=2, device="cuda", **kw)     # warm up
    t = time.perf_counter()
    rd = utils.photometric_offsets(phot, err, mask, models, idxs, reds, dreds, dists, Nmc=a.nmc,
                                   rstate=np.random.RandomState(2), device="cuda", **kw)
    td = time.perf_counter() - t
    t = time.perf_counter()
    utils.photometric_offsets(phot, err, mask, models, idxs, reds, dreds, dists, Nmc=0 or 1,
                              rstate=np.random.RandomState(2), **kw)
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    utils.photometric_offsets(phot, err, mask, models, idxs, reds, dreds, dists,
                              Nmc=1 + a.host_nmc, rstate=np.random.RandomState(2), **kw)
    t2 = time.perf_counter() - t
    per_round = (t2 - t1) / a.host_nmc
    th = t1 + per_round * (a.nmc - 1)
    print({"nobj": No, "nsamps": Ns, "nfilt": 12, "nmc": a.nmc, "device_s": round(td, 3),
           "host_s_extrapolated": round(th, 1), "host_setup_s": round(t1 - per_round, 2),
           "host_s_per_round_all_bands": round(per_round, 3), "ratios_device": rd[0].round(6).tolist()})


if __name__ == "__main__":
    main()
