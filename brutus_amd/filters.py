"""Names of the photometric bands a grid file may carry (HDF5 field names of
the `mag_coeffs` compound dataset), in the order the reference lists them
(`brutus/filters.py:13-29`).  Order matters: it fixes the band axis of the
array `load_models` returns when `filters=None`."""

_FAMILIES = (
    ("Gaia_", ("G_MAW", "BP_MAWf", "RP_MAW")),
    ("SDSS_", tuple("ugriz")),
    ("PS_", ("g", "r", "i", "z", "y", "w", "open")),
    ("DECam_", tuple("ugrizY")),
    ("Bessell_", tuple("UBVRI")),
    ("2MASS_", ("J", "H", "Ks")),
    ("VISTA_", ("Z", "Y", "J", "H", "Ks")),
    ("UKIDSS_", tuple("ZYJHK")),
    ("WISE_W", tuple("1234")),
    ("Tycho_", ("B", "V")),
    ("Hipparcos_", ("Hp",)),
    ("Kepler_", ("D51", "Kp")),
    ("", ("TESS",)),
)

FILTERS = [prefix + band for prefix, bands in _FAMILIES for band in bands]

__all__ = ["FILTERS"]
