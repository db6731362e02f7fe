"""CPU: default Galactic prior.  The astropy-free pieces are pinned against the
reference (tests/golden/galprior_pieces.npz); the Galactocentric frame the reference
takes from astropy (pdf.py:631-635) is restated from astropy's constants and checked
against closed forms (astropy itself is not installed: the assembly stays unpinned by a
reference RUN, see galprior.py)."""
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


def test_astropy_frame_constants_and_closed_form():
    """astropy >= 4.0 `Galactocentric` defaults (galcen_distance 8.122 kpc, z_sun 20.8 pc,
    roll 0, galcen_coord ICRS (266.4051, -28.936175) deg).  With roll 0 astropy's roll0
    aligns the frame with the Galactic one, so the chain of five rotations must reduce to
    'shift by the centre distance, tilt by asin(z_sun / d)' within the ~1.5e-6 rad the
    two pole definitions differ by."""
    M, off = G.astropy_frame()
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-14)          # a rotation
    th = np.arcsin(0.0208 / 8.122)
    H = np.array([[np.cos(th), 0., np.sin(th)], [0., 1., 0.], [-np.sin(th), 0., np.cos(th)]])
    assert np.max(np.abs(M - H)) < 3e-6                          # Galactic-aligned + tilt
    assert np.allclose(off, [-np.sqrt(8.122 ** 2 - 0.0208 ** 2), 0., 0.0208], atol=1e-12)
    # the Sun sits 20.8 pc above the plane, 8.122 kpc from the centre
    R, Z = G.galactic_to_RZ(np.array([0.]), (123., 45.))
    assert abs(np.hypot(R[0], Z[0]) - 8.122) < 1e-12 and abs(Z[0] - 0.0208) < 1e-12
    # the Galactic centre direction: 8.122 kpc along l = b = 0 ends on the centre
    R, Z = G.galactic_to_RZ(np.array([8.122]), (0., 0.))
    assert R[0] < 2e-4 and abs(Z[0]) < 2e-4
    # hand-computed (closed form: x = d cb cl - D, y = d cb sl, z = d sb, then the tilt)
    # for the Orion sightline of the reference's demo data, 0.4 and 2 kpc
    for d, R_ref, Z_ref in ((0.4, 8.466970, -0.109868), (2.0, 9.871201, -0.632540)):
        l, b = np.deg2rad(204.7), np.deg2rad(-19.2)
        x = d * np.cos(b) * np.cos(l) - 8.122
        y, z = d * np.cos(b) * np.sin(l), d * np.sin(b)
        xt, zt = np.cos(th) * x + np.sin(th) * z, -np.sin(th) * x + np.cos(th) * z
        R, Z = G.galactic_to_RZ(np.array([d]), (204.7, -19.2))
        assert abs(R[0] - np.hypot(xt, y)) < 3e-5 and abs(Z[0] - zt) < 3e-5
        assert abs(R[0] - R_ref) < 1e-5 and abs(Z[0] - Z_ref) < 1e-5
    # the frame's Sun is not where the density model is normalised (R_solar 8.2, Z_solar
    # 0.025): the prior at d -> 0 is not 2 ln d + 0, exactly as in the reference
    assert abs(G.gal_lnprior(np.array([1e-3]), (30., 10.))[0] - 2. * np.log(1e-3)) > 1e-3


def test_geometry_and_shapes():
    # towards the Galactic centre the radius shrinks, towards the anticentre it grows
    R0, Z0 = G.galactic_to_RZ(np.array([1.0]), (0., 0.), frame="simple")
    R1, Z1 = G.galactic_to_RZ(np.array([1.0]), (180., 0.), frame="simple")
    assert abs(R0[0] - 7.2) < 1e-12 and abs(R1[0] - 9.2) < 1e-12
    assert abs(Z0[0] - 0.025) < 1e-12
    R2, Z2 = G.galactic_to_RZ(np.array([2.0]), (90., 90.), frame="simple")
    assert abs(R2[0] - 8.2) < 1e-9 and abs(Z2[0] - 2.025) < 1e-12
    # the two frames differ by the Sun's place: 78 pc in R, 4 pc in Z
    Ra, Za = G.galactic_to_RZ(np.array([1.0]), (180., 0.))
    assert abs((R1[0] - Ra[0]) - 0.078) < 1e-3 and abs((0.025 - Za[0]) - 0.0042) < 3e-3
    fa = G.gal_lnprior_simple.device_params()
    assert fa["frame_off"] == (-8.2, 0., 0.025) and len(fa["frame_mat"]) == 9
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
