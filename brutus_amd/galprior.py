"""Default Galactic prior hook (host side, numpy).

Counterpart of reference `pdf.gal_lnprior` (pdf.py:476-749) and its pieces
`logn_disk` (pdf.py:263-307), `logn_halo` (pdf.py:310-377), `logp_feh`
(pdf.py:380-407), `logp_age_from_feh` (pdf.py:410-473): thin disk + thick disk
+ halo number densities times the r^2 volume factor, with per-component
metallicity and age priors mixed by the component membership probabilities.

The pieces are pinned against the reference (tests/golden/galprior_pieces.npz).
The reference converts (l, b, d) to Galactocentric (R, Z) through astropy's
`Galactocentric` frame (pdf.py:631-635); astropy is not installed here, so the
assembled prior cannot be pinned by running the reference.  Instead the
frame itself is restated (`frame="astropy"`, the default): astropy >= 4.0's default
parameter set -- Galactic -> FK5(J2000) -> ICRS rotation, `galcen_coord` ICRS
(266.4051, -28.936175) deg, `roll` 0 (with astropy's `roll0` = 58.5986320306 deg),
`galcen_distance` 8.122 kpc, `z_sun` 20.8 pc -- see `astropy_frame`.  Note that, exactly
as in the reference, the frame's Sun (8.122 kpc, 20.8 pc) is NOT the `R_solar`/`Z_solar`
= (8.2 kpc, 25 pc) at which the density model is normalised.  `frame="simple"` is the
self-consistent geometry of earlier versions of this package (Sun at `R_solar`,
`Z_solar`, l = 0 towards the centre); HISTORY.md section 2.3c quantifies the difference.
"""
from math import erf, log, sqrt

import numpy as np



def _lse(parts):
    """log(sum_k exp(parts[k])) over a short list of equally shaped arrays,
    the usual max-shifted form; lean replacement for scipy's `logsumexp` (whose
    array-API plumbing dominates the runtime of this prior)."""
    m = parts[0]
    for q in parts[1:]:
        m = np.maximum(m, q)
    safe = np.where(np.isfinite(m), m, 0.)
    acc = np.exp(parts[0] - safe)
    for q in parts[1:]:
        acc = acc + np.exp(q - safe)
    return np.log(acc) + safe

__all__ = ["galactic_to_RZ", "astropy_frame", "simple_frame", "logn_disk", "logn_halo",
           "logp_feh", "logp_age_from_feh", "gal_lnprior", "gal_lnprior_simple"]


def _rot(angle, axis):
    """astropy.coordinates.matrix_utilities.rotation_matrix(angle [rad], axis): the
    passive rotation (of the frame) about a coordinate axis."""
    c, s = np.cos(angle), np.sin(angle)
    if axis == "x":
        return np.array([[1., 0., 0.], [0., c, s], [0., -s, c]])
    if axis == "y":
        return np.array([[c, 0., -s], [0., 1., 0.], [s, 0., c]])
    return np.array([[c, s, 0.], [-s, c, 0.], [0., 0., 1.]])


def astropy_frame(galcen_distance=8.122, z_sun=0.0208, roll=0.,
                  galcen_ra=266.4051, galcen_dec=-28.936175):
    """`(M (3, 3), offset (3,))` with  x_galactocentric [kpc] = M @ x_galactic + offset,
    x_galactic = d (cos b cos l, cos b sin l, sin b): astropy's Galactic -> ICRS ->
    `Galactocentric` chain with the defaults of astropy >= 4.0 (`galactocentric_frame_
    defaults` 'v4.0': galcen_distance 8.122 kpc -- GRAVITY 2018 --, z_sun 20.8 pc --
    Bennett & Bovy 2019 --, galcen_coord ICRS (17h45m37.224s, -28d56m10.23s) -- Reid &
    Brunthaler 2004 --, roll 0).  Constants as in astropy's sources:
      builtin_frames/galactic.py         NGP (FK5 J2000) ra 192.8594812065348 deg,
                                         dec 27.12825118085622 deg, lon0 122.9319185680026 deg
      builtin_frames/icrs_fk5_transforms.py   eta0 -19.9 mas, xi0 9.1 mas, da0 -22.9 mas
      builtin_frames/galactocentric.py   roll0 58.5986320306 deg; R = Rx(roll0 - roll)
                                         Ry(-dec_gc) Rz(ra_gc); H = Ry(-asin(z_sun / d_gc));
                                         x' = H R x_icrs - H (d_gc, 0, 0)
    With roll 0 the frame is the Galactic frame shifted to the centre up to ~1.5e-6 rad
    (that is what roll0 is for); what differs from the `simple` geometry are the Sun's
    distance and height and the 0.147 deg tilt H."""
    rad = np.deg2rad
    fk5_to_gal = (_rot(rad(180. - 122.9319185680026), "z")
                  @ _rot(rad(90. - 27.12825118085622), "y") @ _rot(rad(192.8594812065348), "z"))
    mas = np.pi / 180. / 3600e3
    icrs_to_fk5 = _rot(-22.9 * mas, "z") @ _rot(9.1 * mas, "y") @ _rot(19.9 * mas, "x")
    gal_to_icrs = icrs_to_fk5.T @ fk5_to_gal.T
    R = (_rot(rad(58.5986320306 - roll), "x") @ _rot(-rad(galcen_dec), "y")
         @ _rot(rad(galcen_ra), "z"))
    H = _rot(-np.arcsin(z_sun / galcen_distance), "y")
    return H @ R @ gal_to_icrs, -H @ np.array([galcen_distance, 0., 0.])


def simple_frame(R_solar=8.2, Z_solar=0.025):
    """The same pair for the self-consistent geometry: Sun at radius `R_solar`, height
    `Z_solar`, l = 0 towards the centre, no tilt."""
    return np.diag([1., 1., 1.]), np.array([-R_solar, 0., Z_solar])


_ASTROPY_FRAME = astropy_frame()


def _frame(frame, R_solar, Z_solar):
    if isinstance(frame, str):
        if frame == "astropy":
            return _ASTROPY_FRAME
        if frame == "simple":
            return simple_frame(R_solar, Z_solar)
        raise ValueError("frame must be 'astropy', 'simple' or a (matrix, offset) pair")
    return np.asarray(frame[0], dtype=np.float64), np.asarray(frame[1], dtype=np.float64)


def galactic_to_RZ(dists, coord, R_solar=8.2, Z_solar=0.025, frame="astropy"):
    """Heliocentric Galactic (l, b) [deg] and distance [kpc] -> cylindrical
    Galactocentric radius R and height Z [kpc] (reference pdf.py:631-635).  `frame`:
    "astropy" (the reference's route, see `astropy_frame`), "simple" (Sun at `R_solar`,
    `Z_solar`) or an explicit `(M, offset)` pair."""
    M, off = _frame(frame, R_solar, Z_solar)
    ell, b = np.deg2rad(coord[0]), np.deg2rad(coord[1])
    d = np.asarray(dists, dtype=np.float64)
    n = np.array([np.cos(b) * np.cos(ell), np.cos(b) * np.sin(ell), np.sin(b)])
    u = M @ n                                   # direction of the sightline in the frame
    x = off[0] + d * u[0]
    y = off[1] + d * u[1]
    z = off[2] + d * u[2]
    return np.hypot(x, y), z


def logn_disk(R, Z, R_solar=8.2, Z_solar=0.025, R_scale=2.6, Z_scale=0.3,
              R_smooth=2.0):
    """Exponential disk, normalised to zero at the solar position."""
    R_eff = np.sqrt(np.square(R) + R_smooth ** 2)
    return -((R_eff - R_solar) / R_scale
             + (np.abs(Z) - abs(Z_solar)) / Z_scale)


def logn_halo(R, Z, R_solar=8.2, Z_solar=0.025, R_smooth=2.0, eta=4.2,
              q_ctr=0.2, q_inf=0.8, r_q=6.):
    """Power-law halo with radius-dependent oblateness, normalised at the Sun."""
    def flattening(r2):
        return q_inf - (q_inf - q_ctr) * np.exp(1. - np.sqrt(r2 + r_q ** 2) / r_q)

    def r_eff(Rc, Zc):
        q = flattening(np.square(Rc) + np.square(Zc))
        return np.sqrt(np.square(Rc) + np.square(Zc / q) + R_smooth ** 2)

    return -eta * np.log(r_eff(R, Z) / r_eff(R_solar, Z_solar))


def logp_feh(feh, feh_mean=-0.2, feh_sigma=0.3):
    """Gaussian ln pdf in [Fe/H]."""
    return -0.5 * (np.square(feh_mean - feh) / feh_sigma ** 2
                   + np.log(2. * np.pi * feh_sigma ** 2))


def logp_age_from_feh(age, feh_mean=-0.2, max_age=13.8, min_age=0.,
                      feh_age_ctr=-0.5, feh_age_scale=0.5,
                      nsigma_from_max_age=2., max_sigma=4., min_sigma=1.):
    """Truncated-normal ln pdf in age [Gyr] whose mean follows the component's
    mean metallicity through a logistic age-metallicity relation."""
    mean = ((max_age - min_age)
            / (1. + np.exp((feh_mean - feh_age_ctr) / feh_age_scale)) + min_age)
    sigma = min(max((max_age - mean) / nsigma_from_max_age, min_sigma), max_sigma)
    age = np.asarray(age, dtype=np.float64)
    lo, hi = (min_age - mean) / sigma, (max_age - mean) / sigma
    lnnorm = log(sigma / 2.) + log(erf(hi / sqrt(2.)) - erf(lo / sqrt(2.)))
    out = -log(sqrt(2. * np.pi)) - 0.5 * np.square((age - mean) / sigma) - lnnorm
    outside = (age < min_age) | (age > max_age)
    return np.where(outside, -np.inf, out)


def gal_lnprior(dists, coord, labels=None, R_solar=8.2, Z_solar=0.025,
                R_thin=2.6, Z_thin=0.3, Rs_thin=2.0,
                R_thick=2.0, Z_thick=0.9, f_thick=0.04, Rs_thick=2.0,
                Rs_halo=2.0, q_halo_ctr=0.2, q_halo_inf=0.8, r_q_halo=6.0,
                eta_halo=4.2, f_halo=0.005,
                feh_thin=-0.2, feh_thin_sigma=0.3,
                feh_thick=-0.7, feh_thick_sigma=0.4,
                feh_halo=-1.6, feh_halo_sigma=0.5,
                max_age=13.8, min_age=0., feh_age_ctr=-0.5, feh_age_scale=0.5,
                nsigma_from_max_age=2., max_sigma=4., min_sigma=1.,
                return_components=False, frame="astropy"):
    """ln prior over distance (and, through `labels['feh']` / `labels['loga']`,
    metallicity and age) for a sightline `coord = (l, b)` in degrees.  Same
    signature, defaults and component model as reference pdf.py:476-749; `frame`
    (extension, trailing keyword) selects the Galactocentric geometry, by default the
    reference's astropy route (`galactic_to_RZ`)."""
    dists = np.asarray(dists, dtype=np.float64)
    with np.errstate(all="ignore"):
        volume = 2. * np.log(dists + 1e-300)
        R, Z = galactic_to_RZ(dists, coord, R_solar=R_solar, Z_solar=Z_solar, frame=frame)
        comp = [
            logn_disk(R, Z, R_solar, Z_solar, R_thin, Z_thin, Rs_thin) + volume,
            logn_disk(R, Z, R_solar, Z_solar, R_thick, Z_thick, Rs_thick)
            + volume + np.log(f_thick),
            logn_halo(R, Z, R_solar, Z_solar, Rs_halo, eta_halo, q_halo_ctr,
                      q_halo_inf, r_q_halo) + volume + np.log(f_halo),
        ]
        lnprior = _lse(comp)
        components = {"number_density": comp}
        if labels is not None:
            member = [c - lnprior for c in comp]   # ln P(component | position)
            names = getattr(getattr(labels, "dtype", None), "names", None) or ()
            if "feh" in names:
                feh = labels["feh"]
                parts = [logp_feh(feh, m, s) + w for (m, s), w in
                         zip(((feh_thin, feh_thin_sigma),
                              (feh_thick, feh_thick_sigma),
                              (feh_halo, feh_halo_sigma)), member)]
                lnprior = lnprior + _lse(parts)
                components["feh"] = parts
            if "loga" in names:
                age = 10. ** labels["loga"] / 1e9
                parts = [logp_age_from_feh(age, m, max_age, min_age, feh_age_ctr,
                                           feh_age_scale, nsigma_from_max_age,
                                           max_sigma, min_sigma) + w
                         for m, w in zip((feh_thin, feh_thick, feh_halo), member)]
                lnprior = lnprior + _lse(parts)
                components["age"] = parts
    if return_components:
        return lnprior, components
    return lnprior


def device_params(**kw):
    """The model constants of `gal_lnprior` (defaults of reference pdf.py:476-
    488, overridable by keyword) as the dict of fields `brutus_post_params`
    carries; the truncated-normal age prior is reduced to (mean, sigma,
    ln-normalisation) per component here on the host."""
    d = dict(R_solar=8.2, Z_solar=0.025, R_thin=2.6, Z_thin=0.3, Rs_thin=2.0,
             R_thick=2.0, Z_thick=0.9, f_thick=0.04, Rs_thick=2.0, Rs_halo=2.0,
             q_halo_ctr=0.2, q_halo_inf=0.8, r_q_halo=6.0, eta_halo=4.2,
             f_halo=0.005, feh_thin=-0.2, feh_thin_sigma=0.3, feh_thick=-0.7,
             feh_thick_sigma=0.4, feh_halo=-1.6, feh_halo_sigma=0.5,
             max_age=13.8, min_age=0., feh_age_ctr=-0.5, feh_age_scale=0.5,
             nsigma_from_max_age=2., max_sigma=4., min_sigma=1., frame="astropy")
    d.update(kw)
    means = (d["feh_thin"], d["feh_thick"], d["feh_halo"])
    out = {k: d[k] for k in ("R_solar", "Z_solar", "R_thin", "Z_thin", "Rs_thin",
                             "R_thick", "Z_thick", "f_thick", "Rs_thick",
                             "Rs_halo", "q_halo_ctr", "q_halo_inf", "r_q_halo",
                             "eta_halo", "f_halo", "min_age", "max_age")}
    out["feh_mean"] = means
    out["feh_sigma"] = (d["feh_thin_sigma"], d["feh_thick_sigma"], d["feh_halo_sigma"])
    am, asg, aln = [], [], []
    for m in means:
        mean = ((d["max_age"] - d["min_age"])
                / (1. + np.exp((m - d["feh_age_ctr"]) / d["feh_age_scale"])) + d["min_age"])
        sig = min(max((d["max_age"] - mean) / d["nsigma_from_max_age"], d["min_sigma"]),
                  d["max_sigma"])
        lo, hi = (d["min_age"] - mean) / sig, (d["max_age"] - mean) / sig
        am.append(mean)
        asg.append(sig)
        aln.append(log(sig / 2.) + log(erf(hi / sqrt(2.)) - erf(lo / sqrt(2.))))
    out["age_mean"], out["age_sigma"], out["age_lnnorm"] = tuple(am), tuple(asg), tuple(aln)
    M, off = _frame(d["frame"], d["R_solar"], d["Z_solar"])
    out["frame_mat"] = tuple(float(x) for x in np.asarray(M).ravel())
    out["frame_off"] = tuple(float(x) for x in off)
    return out


#: `lnpost` may pass the (Nsel,) label table for (Nmc, Nsel) distances instead
#: of a tiled copy: every label use above broadcasts.
gal_lnprior.broadcasts_labels = True
#: marks the hook as the built-in model the device `lnpost` implements
gal_lnprior.device_params = device_params


def gal_lnprior_simple(dists, coord, labels=None, **kw):
    """`gal_lnprior` with the self-consistent `simple` geometry (Sun at `R_solar`,
    `Z_solar`); also runs on the device."""
    kw.setdefault("frame", "simple")
    return gal_lnprior(dists, coord, labels=labels, **kw)


gal_lnprior_simple.broadcasts_labels = True
gal_lnprior_simple.device_params = lambda **kw: device_params(**dict(dict(frame="simple"), **kw))
