"""Container-only helper: import the upstream reference (`/root/reference/brutus`)
under shims for the third-party modules that are absent from this image.

This module is test-infrastructure tooling.  It is used ONLY by
`tools/gen_golden.py` and by the optional `tests/test_oracle_vs_reference.py`
(which skips itself when `/root/reference` is not present, e.g. on the GPU box).
Nothing in the product package imports it.

Shims:
  * `numba.jit` -> identity decorator, so the four `@jit(nopython=True)` loops
    (fitting.py:34,274,430; utils.py:286) run as plain Python.
  * empty stand-in modules for h5py / healpy / pooch / astropy, which are only
    needed by parts of the reference that are out of scope for this path.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BRUTUS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "brutus"))


def import_reference():
    """Return the tuple (fitting, utils, pdf, cluster) of reference modules."""
    if not reference_available():
        raise ImportError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree

    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def jit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f

        numba.jit = jit
        sys.modules["numba"] = numba

    def _stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    _stub("h5py")
    _stub("healpy")

    class _Pooch(object):
        def __init__(self, *a, **k):
            self.base_url = ""
            self.registry = {}
            self.urls = {}

        def fetch(self, *a, **k):
            raise RuntimeError("no network")

    _stub("pooch", create=lambda *a, **k: _Pooch(), os_cache=lambda *a, **k: "/tmp")
    ap = _stub("astropy")
    ap.units = _stub("astropy.units")
    ap.coordinates = _stub("astropy.coordinates", SkyCoord=object,
                           CylindricalRepresentation=object)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from brutus import fitting, utils, pdf, cluster  # noqa: E402
    return fitting, utils, pdf, cluster
