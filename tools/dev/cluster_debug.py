import sys, os
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import make_cluster_data, relerr
from brutus_amd import cluster
from oracle import brutus_oracle as O
THETA = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])
iso, phot, err, par, perr = make_cluster_data(5000, 12, 5)
sl = slice(0, 250)
c = O.isochrone_loglike(THETA, iso, phot[sl], err[sl], parallax=par[sl], parallax_err=perr[sl], return_lnls=True)
for n in (250, 1000, 5000):
    a = cluster.isochrone_loglike(THETA, iso, phot[:n], err[:n], parallax=par[:n], parallax_err=perr[:n], return_lnls=True, cache=False)
    d = np.abs(a[1][:250] - c[1]) / np.abs(c[1])
    bad = np.where(d > 1e-9)[0]
    print(n, "max rel", d.max(), "nbad", bad.size, bad[:10], a[1][bad[:4]], c[1][bad[:4]])
    if bad.size:
        i = bad[0]
        print("  object", i, "phot finite", np.isfinite(phot[i]).sum(), "par", par[i], perr[i])
a = cluster.isochrone_loglike(THETA, iso, phot, err, parallax=par, parallax_err=perr, return_lnls=True, cache=False)
for lo in range(0, 5000, 250):
    sl = slice(lo, lo + 250)
    c = O.isochrone_loglike(THETA, iso, phot[sl], err[sl], parallax=par[sl], parallax_err=perr[sl], return_lnls=True)
    d = np.abs(a[1][sl] - c[1]) / np.abs(c[1])
    bad = np.where(~(d < 1e-9))[0]
    if bad.size:
        i = lo + bad[0]
        print(lo, "nbad", bad.size, "first", i, a[1][i], c[1][bad[0]], "finite bands", np.isfinite(phot[i]).sum(), np.isfinite(err[i]).sum(), "par", par[i], perr[i])
        a1 = cluster.isochrone_loglike(THETA, iso, phot[i:i+1], err[i:i+1], parallax=par[i:i+1], parallax_err=perr[i:i+1], return_lnls=True, cache=False)
        print("   alone:", a1[1][0])
