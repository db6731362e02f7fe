#!/usr/bin/env python
"""CPU baseline for bench.py: time the C restatement (oracle/loglike_ref.c) on
the host cores, one serial star per worker PROCESS (separate address spaces:
128 threads page-faulting 0.3 GB each inside one process serialise on the mm
lock).  TEST INFRASTRUCTURE: executed only by bench.py's `cpu_baseline` leg.

    python -m oracle.cpu_bench --config 2 --seconds 20 --procs 64
prints one JSON object.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_G = {}


def _init(nmodel, nfilt, config):
    from brutus_amd import synth
    from oracle import c_oracle
    c_oracle._load().brutus_ref_set_threads(1)
    _G["models"], _, _ = synth.make_mist_like_grid(nmodel, nfilt)
    _G["stars"] = synth.make_stars(_G["models"], 64, seed=1 if config == 2 else 2,
                                   with_parallax=(config == 3))
    _G["kw"] = dict(rvlim=(3.32, 3.32)) if config == 2 else {}


def _work(i):
    from oracle import c_oracle
    st = _G["stars"]
    k = i % len(st["flux"])
    par, pe = st["parallax"][k], st["parallax_err"][k]
    if not np.isfinite(par):
        par, pe = None, None
    t = time.time()
    c_oracle.loglike(st["flux"][k], st["err"][k], st["mask"][k], _G["models"],
                     parallax=par, parallax_err=pe, **_G["kw"])
    return time.time() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--nmodel", type=int, default=750000)
    ap.add_argument("--nfilt", type=int, default=12)
    ap.add_argument("--seconds", type=float, default=20.)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--single-stars", type=int, default=0,
                    help="instead of the pool: this many stars one after the other in THIS "
                         "process, one thread (the single-core figure)")
    a = ap.parse_args()
    procs = a.procs or max(1, min(os.cpu_count() or 1, 128))
    # the grid is built once in the parent and inherited by fork (copy-on-write)
    _init(a.nmodel, a.nfilt, a.config)
    if a.single_stars > 0:
        t0 = time.time()
        per = [_work(i) for i in range(a.single_stars)]
        dt = time.time() - t0
        print(json.dumps({"value": a.single_stars / dt, "unit": "stars/s", "cores": 1, "kind": "port",
                          "sample": "%d stars x %d models x %d bands in %.1f s, one process, one "
                                    "thread (C restatement oracle/loglike_ref.c; %.2f s per star)"
                                    % (a.single_stars, a.nmodel, a.nfilt, dt, float(np.mean(per)))}))
        return
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        t0 = time.time()
        n, per = 0, []
        while time.time() - t0 < a.seconds * 0.6:
            per += pool.map(_work, range(n, n + procs))
            n += procs
        dt = time.time() - t0
    print(json.dumps({"value": n / dt, "unit": "stars/s", "cores": procs, "kind": "port",
                      "host_cores": os.cpu_count(),
                      "sample": "%d stars x %d models x %d bands in %.1f s (C restatement "
                                "oracle/loglike_ref.c, %d worker processes, one serial star "
                                "each; mean %.2f s per star per core)"
                                % (n, a.nmodel, a.nfilt, dt, procs, float(np.mean(per)))}))


if __name__ == "__main__":
    main()
