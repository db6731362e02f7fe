"""cProfile of cluster.isochrone_loglike with a new theta per call (bench configs[4] workload)."""
import cProfile, pstats, sys, os, io
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
import torch
from brutus_amd import cluster, synth
iso = synth.TableIsochrone(nbands=12, neep=2000)
phot, err, par, perr = synth.make_cluster(iso, 5000, seed=11)
theta0 = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])
walk = np.random.RandomState(5).normal(size=(1000, 6))
def call(k):
    th = theta0 + np.array([1e-3, 1e-3, 1e-3, 0., 0.5, 1e-3]) * walk[k]
    return cluster.isochrone_loglike(th, iso, phot, err, parallax=par, parallax_err=perr)
for k in range(20): call(900 + k)
import time
t0 = time.perf_counter()
for k in range(300): call(k)
print("ms per call", (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile()
pr.enable()
for k in range(300, 600): call(k)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
