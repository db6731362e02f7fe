"""BASELINE configs[0] as written -- "100 synthetic stars, 10k-model mini-grid, 6 bands, numpy CPU
BruteForce.fit (plumbing, no GPU)" -- on the CPU restatement: oracle/cpu_fit.py is `BruteForce.fit`
restated over `brutus_oracle.fit_star` (itself pinned by the reference-generated `_fit` goldens,
tests/test_oracle_golden.py) and writes the reference's HDF5 layout through the package's libhdf5
writer.  The PRODUCT has no CPU path (tests/test_cabi.py); the same shape runs on the GPU against this
restatement row by row in tests/test_gpu_parity.py::test_fit_end_to_end_hdf5."""
import os

import numpy as np
import pytest

from helpers import galprior


def test_config0_cpu_fit_writes_the_reference_layout(tmp_path):
    from brutus_amd import h5io, synth
    from oracle import brutus_oracle as O
    from oracle import cpu_fit
    if not h5io.hdf5_available():
        pytest.skip("libhdf5 not found")
    models, labels, lmask = synth.make_grid(10000, 6, seed=1)
    n = 100
    st = synth.make_stars(models, n, seed=2)
    objid = np.zeros(n, dtype=[("id", "i8"), ("l", "f8"), ("b", "f8")])
    objid["id"] = np.arange(n)
    path = os.path.join(str(tmp_path), "cfg0")
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], data_coords=st["coords"],
              lngalprior=galprior, Nmc_prior=25, Ndraws=60)
    cpu_fit.fit(models, labels, lmask, st["flux"], st["err"], st["mask"], objid, path,
                rstate=np.random.RandomState(862), **kw)
    with pytest.raises(OSError):                  # "w-": never overwrite (fitting.py:1632)
        cpu_fit.fit(models, labels, lmask, st["flux"], st["err"], st["mask"], objid, path, **kw)
    f = path + ".h5"
    spec = {"model_idx": ("int32", (n, 60)), "ml_scale": ("float32", (n, 60)), "ml_av": ("float32", (n, 60)),
            "ml_rv": ("float32", (n, 60)), "ml_cov_sar": ("float32", (n, 60, 3, 3)),
            "obj_log_post": ("float32", (n, 60)), "obj_log_evid": ("float32", (n,)),
            "obj_chi2min": ("float32", (n,)), "obj_Nbands": ("int16", (n,)),
            "samps_dist": ("float32", (n, 60)), "samps_red": ("float32", (n, 60)),
            "samps_dred": ("float32", (n, 60)), "samps_logp": ("float32", (n, 60))}
    assert set(h5io.list_datasets(f)) == set(spec) | {"labels"}
    got = {k: h5io.read_dataset(f, k) for k in spec}
    for k, (dt, shape) in spec.items():
        assert got[k].dtype == np.dtype(dt) and got[k].shape == shape, k
    assert np.array_equal(h5io.read_dataset(f, "labels")["id"], np.arange(n))
    assert got["model_idx"].min() >= 0 and got["model_idx"].max() < 10000      # no row left at the -99 sentinel
    assert np.all(got["obj_Nbands"] >= 4) and np.all(np.isfinite(got["obj_log_evid"]))
    assert np.all(got["samps_dist"] > 0)
    # the rows are the star loop's: the first three objects again, with the stream restarted
    rs = np.random.RandomState(862)
    lnprior = O.static_lnprior(labels, lmask)
    with np.errstate(all="ignore"):
        mag, merr = O.magnitude(st["flux"], st["err"])
    mask = st["mask"] & ~((mag > 50.) | (merr > 0.25))
    for i in range(3):
        row = O.fit_star(st["flux"][i], st["err"][i], mask[i], models, lnprior, labels, st["coords"][i],
                         st["parallax"][i], st["parallax_err"][i], rs, galprior, Nmc_prior=25, Ndraws=60)
        assert np.array_equal(got["model_idx"][i], row[0])
        assert np.allclose(got["obj_log_post"][i], np.asarray(row[6], dtype=np.float32))
        assert got["obj_Nbands"][i] == row[5]
