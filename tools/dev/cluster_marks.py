"""Where a cluster.isochrone_loglike call spends its time on the host (marks of cluster._TRACE),
mean over 300 calls with a new theta each."""
import sys, os, collections
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
import time
from brutus_amd import cluster, synth
iso = synth.TableIsochrone(nbands=12, neep=2000)
phot, err, par, perr = synth.make_cluster(iso, 5000, seed=11)
theta0 = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])
walk = np.random.RandomState(5).normal(size=(1000, 6))
def call(k):
    th = theta0 + np.array([1e-3, 1e-3, 1e-3, 0., 0.5, 1e-3]) * walk[k]
    return cluster.isochrone_loglike(th, iso, phot, err, parallax=par, parallax_err=perr)
for k in range(20): call(900 + k)
acc = collections.OrderedDict()
n = 300
t_all = 0.
for k in range(n):
    cluster._TRACE = []
    t0 = time.perf_counter()
    call(k)
    t1 = time.perf_counter()
    t_all += t1 - t0
    prev = t0
    for lab, t in cluster._TRACE:
        acc[lab] = acc.get(lab, 0.) + (t - prev)
        prev = t
    acc["return"] = acc.get("return", 0.) + (t1 - prev)
cluster._TRACE = None
print("ms per call %.3f" % (t_all / n * 1e3))
for lab, v in acc.items():
    print("  %-14s %.1f us" % (lab, v / n * 1e6))
