"""Consumers of the fit output on the device:
`utils.photometric_offsets(..., device="cuda")` against the vector generated from the
upstream code and against the host form on the same random stream.  The object counts are
integers (exact); the ratios are medians of float64 quotients whose fluxes come from the
device's `exp10` (1 ulp from numpy's `10.**x`): 1e-11 relative."""
import os

import numpy as np
import pytest

from brutus_amd import utils

pytestmark = pytest.mark.gpu
RTOL = 1e-11


def test_device_offsets_match_reference_vector():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "consumers.npz"))
    r, re_, n = utils.photometric_offsets(
        z["po_phot"], z["po_err"], z["po_mask"], z["po_models"], z["po_idxs"],
        z["po_reds"], z["po_dreds"], z["po_dists"], sel=z["po_sel"],
        weights=z["po_weights"], mask_fit=z["po_mask_fit"], Nmc=20,
        old_offsets=z["po_old"], prior_mean=np.ones(6), prior_std=np.full(6, 0.05),
        verbose=False, rstate=np.random.RandomState(5), device="cuda")
    assert np.array_equal(n, z["po_nratio"])
    assert np.max(np.abs(r / z["po_ratios"] - 1.)) < RTOL
    assert np.max(np.abs(re_ - z["po_ratios_err"])) < RTOL


def _case(seed, Nobj, Ns, Nf, Nm):
    rng = np.random.RandomState(seed)
    models = np.zeros((Nm, Nf, 3), dtype=np.float32)
    models[:, :, 0] = rng.uniform(10, 20, (Nm, Nf))
    models[:, :, 1] = rng.uniform(0.5, 3, (Nm, Nf))
    models[:, :, 2] = rng.uniform(0, 0.3, (Nm, Nf))
    idxs = rng.randint(0, Nm, (Nobj, Ns))
    reds, dreds = rng.uniform(0, 1, (Nobj, Ns)), rng.uniform(3, 3.6, (Nobj, Ns))
    dists = rng.uniform(0.5, 3, (Nobj, Ns))
    seds0 = utils.get_seds(models[idxs[:, 0]], av=reds[:, 0], rv=dreds[:, 0],
                           return_flux=True) / dists[:, 0, None] ** 2
    phot = seds0 * (1 + 0.05 * rng.normal(size=seds0.shape))
    err = 0.05 * phot
    mask = rng.uniform(size=phot.shape) > 0.15
    w = rng.uniform(size=(Nobj, Ns))
    w[5] = 0
    sel = rng.uniform(size=Nobj) > 0.1
    return dict(phot=phot, err=err, mask=mask, models=models, idxs=idxs, reds=reds,
                dreds=dreds, dists=dists, sel=sel, weights=w)


@pytest.mark.parametrize("dim_prior", (True, False))
def test_device_offsets_equal_host_form(dim_prior):
    """Mixed band masks, a zero-weight object, deselected objects, a band outside the fit,
    Nsamps above one scan tile (300 > 256), an odd and an even number of objects per band."""
    for seed, Nobj in ((0, 301), (1, 150)):
        c = _case(seed, Nobj, 300, 6, 500)
        kw = dict(mask_fit=np.array([1, 1, 0, 1, 1, 1], bool), Nmc=12,
                  old_offsets=np.linspace(0.98, 1.02, 6), dim_prior=dim_prior, verbose=False)
        a = utils.photometric_offsets(rstate=np.random.RandomState(3), **c, **kw)
        b = utils.photometric_offsets(rstate=np.random.RandomState(3), device="cuda", **c, **kw)
        assert np.array_equal(a[2], b[2])
        assert np.max(np.abs(b[0] / a[0] - 1.)) < RTOL
        assert np.max(np.abs(b[1] - a[1])) < RTOL


def test_device_offsets_consume_the_same_stream_and_reject_float64_grids():
    c = _case(2, 60, 20, 5, 100)
    r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
    utils.photometric_offsets(rstate=r1, Nmc=5, verbose=False, **c)
    utils.photometric_offsets(rstate=r2, Nmc=5, verbose=False, device="cuda", **c)
    assert r1.random_sample() == r2.random_sample()
    c["models"] = c["models"].astype(np.float64) + 1e-9
    with pytest.raises(ValueError):
        utils.photometric_offsets(Nmc=2, verbose=False, device="cuda", **c)
    # no object qualifies: ones, zeros and no device work beyond the weights
    c = _case(3, 10, 8, 5, 50)
    c["sel"] = np.zeros(10, dtype=bool)
    r, e, n = utils.photometric_offsets(Nmc=3, verbose=False, device="cuda", **c)
    assert np.array_equal(r, np.ones(5)) and not e.any() and not n.any()
