"""cProfile of repeated `cluster.isochrone_loglike` calls with a new theta each (GPU box)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from brutus_amd import cluster, synth

iso = synth.TableIsochrone(nbands=12, neep=2000)
phot, err, par, perr = synth.make_cluster(iso, 5000, seed=11)
th = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])
w = np.random.RandomState(1).normal(size=(600, 6))
f = lambda k: cluster.isochrone_loglike(th + 1e-3 * w[k], iso, phot, err, parallax=par,
                                        parallax_err=perr)
for k in range(5):
    f(k)
pr = cProfile.Profile()
pr.enable()
for k in range(5, 405):
    f(k)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
