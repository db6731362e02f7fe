"""CPU: default Galactic prior.  The astropy-free pieces are pinned against the
reference (tests/golden/galprior_pieces.npz); the assembled prior is checked
for self-consistency (parity of the assembly is unpinned, see galprior.py)."""
import os

import numpy as np

from helpers import GOLDEN, relerr
from brutus_amd import galprior as G


def test_pieces_match_reference():
    z = np.load(os.path.join(GOLDEN, "galprior_pieces.npz"))
    R, Z, feh, age = z["R"], z["Z"], z["feh"], z["age"]
    assert relerr(z["disk_thin"], G.logn_disk(R, Z)) < 1e-13
    assert relerr(z["disk_thick"], G.logn_disk(R, Z, R_scale=2.0, Z_scale=0.9)) < 1e-13
    assert relerr(z["halo"], G.logn_halo(R, Z)) < 1e-12
    assert relerr(z["feh_thin"], G.logp_feh(feh)) < 1e-13
    assert relerr(z["feh_halo"], G.logp_feh(feh, feh_mean=-1.6, feh_sigma=0.5)) < 1e-13
    assert relerr(z["age_thin"], G.logp_age_from_feh(age, feh_mean=-0.2)) < 1e-12
    assert relerr(z["age_thick"], G.logp_age_from_feh(age, feh_mean=-0.7)) < 1e-12
    assert relerr(z["age_halo"], G.logp_age_from_feh(age, feh_mean=-1.6)) < 1e-12


def test_geometry_and_shapes():
    # towards the Galactic centre the radius shrinks, towards the anticentre it grows
    R0, Z0 = G.galactic_to_RZ(np.array([1.0]), (0., 0.))
    R1, Z1 = G.galactic_to_RZ(np.array([1.0]), (180., 0.))
    assert abs(R0[0] - 7.2) < 1e-12 and abs(R1[0] - 9.2) < 1e-12
    assert abs(Z0[0] - 0.025) < 1e-12
    R2, Z2 = G.galactic_to_RZ(np.array([2.0]), (90., 90.))
    assert abs(R2[0] - 8.2) < 1e-9 and abs(Z2[0] - 2.025) < 1e-12
    # (Nmc, Nsel)-shaped calls with tiled labels, as lnpost makes them
    lab = np.zeros(5, dtype=[("feh", "f8"), ("loga", "f8")])
    lab["feh"] = np.linspace(-2, 0.3, 5)
    lab["loga"] = np.linspace(8.5, 10.1, 5)
    d = np.abs(np.random.RandomState(0).normal(1., 0.3, size=(7, 5)))
    lab_mc = np.tile(lab, 7).reshape(7, 5)
    lp = G.gal_lnprior(d, (204.7, -19.2), labels=lab_mc)
    assert lp.shape == (7, 5) and np.all(np.isfinite(lp))
    lp1, comp = G.gal_lnprior(d[0], (204.7, -19.2), labels=lab, return_components=True)
    assert np.allclose(lp1, lp[0])
    assert set(comp) == {"number_density", "feh", "age"}
