"""CPU: grid-file / offsets loaders and the libhdf5 binding (h5io)."""
import os

import numpy as np
import pytest

from brutus_amd import h5io, utils


def _write_grid(path, n=50):
    rng = np.random.RandomState(0)
    ctype = np.dtype([("PS_g", "f4", (3,)), ("PS_r", "f4", (3,)),
                      ("2MASS_J", "f4", (3,)), ("WISE_W1", "f4", (3,))])
    coeffs = np.zeros(n, dtype=ctype)
    for name in ("PS_g", "PS_r", "2MASS_J"):       # WISE_W1 stays all-zero
        coeffs[name] = rng.normal(size=(n, 3))
    labels = np.zeros(n, dtype=[("mini", "f8"), ("eep", "f8"), ("feh", "f8"),
                                ("smf", "f8")])
    labels["mini"] = rng.uniform(0.5, 2, n)
    labels["eep"] = np.linspace(300, 600, n)
    labels["feh"] = rng.uniform(-2, 0.5, n)
    labels["smf"] = np.where(np.arange(n) % 5 == 0, 0.5, 0.)
    params = np.zeros(n, dtype=[("loga", "f8"), ("agewt", "f8"), ("junk", "f8")])
    params["loga"] = rng.uniform(8, 10, n)
    params["agewt"] = rng.uniform(0.1, 1, n)
    h5io.write_datasets(path, {"mag_coeffs": coeffs, "labels": labels,
                               "parameters": params})
    return coeffs, labels, params


def test_load_models_roundtrip(tmp_path):
    path = os.path.join(str(tmp_path), "grid.h5")
    coeffs, labels, params = _write_grid(path)
    models, lab, lmask = utils.load_models(path, verbose=False)
    single = labels["smf"] == 0.
    assert models.dtype == np.float32 and models.shape == (single.sum(), 3, 3)
    # band order follows FILTERS: PS_g, PS_r, ..., 2MASS_J; all-zero WISE_W1 dropped
    assert np.array_equal(models[:, 0], coeffs["PS_g"][single])
    assert np.array_equal(models[:, 2], coeffs["2MASS_J"][single])
    assert lab.dtype.names == ("mini", "feh", "eep", "loga", "agewt")
    assert np.array_equal(lab["loga"], params["loga"][single])
    assert bool(lmask["mini"][0]) and not bool(lmask["loga"][0])
    m2, l2, _ = utils.load_models(path, filters=["2MASS_J", "PS_g"],
                                  include_postms=False, include_binaries=True,
                                  verbose=False)
    ms = labels["eep"] <= 454.
    assert m2.shape == (ms.sum(), 2, 3) and "smf" in l2.dtype.names
    assert np.array_equal(m2[:, 0], coeffs["2MASS_J"][ms])
    with pytest.raises(ValueError, match="nothing left"):
        utils.load_models(path, include_ms=False, include_postms=False)


def test_load_offsets(tmp_path):
    path = os.path.join(str(tmp_path), "off.txt")
    with open(path, "w") as f:
        f.write("PS_g 1.02\n2MASS_J 0.97\n")
    off = utils.load_offsets(path, filters=["PS_g", "PS_r", "2MASS_J"], verbose=False)
    assert np.allclose(off, [1.02, 1.0, 0.97])


def test_results_file_layout_and_wminus(tmp_path):
    path = os.path.join(str(tmp_path), "res.h5")
    lab = np.zeros(4, dtype=[("id", "i8")])
    rf = h5io.ResultsFile(path, 4, 3, lab, True, flush_every=2)
    # rows not written keep the reference's fill values (fitting.py:1635-1662)
    rf.write_row(1, (np.array([5, 6, 7]), np.ones(3), np.ones(3), np.ones(3),
                     np.ones((3, 3, 3)), 7, np.ones(3), -2.5, 3.5, np.ones(3),
                     np.ones(3), np.ones(3), np.ones(3)))
    rf.close()
    idx = h5io.read_dataset(path, "model_idx")
    assert idx.dtype == np.int32 and np.all(idx[0] == -99) and list(idx[1]) == [5, 6, 7]
    assert np.all(h5io.read_dataset(path, "ml_scale")[0] == 1.)
    assert np.all(h5io.read_dataset(path, "ml_av")[2] == 0.)
    assert h5io.read_dataset(path, "obj_Nbands").dtype == np.int16
    assert h5io.read_dataset(path, "samps_logp").dtype == np.float32
    with pytest.raises(OSError):
        h5io.ResultsFile(path, 4, 3, lab, True)
    # running_io=False writes everything at close
    p2 = os.path.join(str(tmp_path), "res2.h5")
    rf = h5io.ResultsFile(p2, 2, 3, None, False, running_io=False)
    rf.close()
    assert "samps_dist" not in h5io.list_datasets(p2)
    assert "model_idx" in h5io.list_datasets(p2)


def test_read_reference_style_compound():
    """Compound rows with sub-array members (the layout of the reference's
    demo catalogue demos/Orion_l204.7_b-19.2.h5)."""
    import tempfile
    d = np.zeros(3, dtype=[("obj_id", "u8"), ("l", "f8"), ("mag", "f4", (8,)),
                           ("parallax", "f4")])
    d["mag"] = np.arange(24).reshape(3, 8)
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "c.h5")
        h5io.write_datasets(p, {"cat": d})
        back = h5io.read_dataset(p, "cat")
    assert back.dtype.names == d.dtype.names
    assert np.array_equal(back["mag"], d["mag"])


def test_output_consumers_match_reference():
    """get_seds / draw_sar / phot_loglike (reference utils.py:1089-1215, 765-842)
    against vectors generated from the upstream code (tests/golden/consumers.npz)."""
    import os as _os
    z = np.load(_os.path.join(_os.path.dirname(__file__), "golden", "consumers.npz"))
    m, av, rv = z["models"], z["av"], z["rv"]
    assert np.max(np.abs(utils.get_seds(m, av=av, rv=rv) - z["seds_mag"])) == 0.
    got = utils.get_seds(m, av=av, rv=rv, return_flux=True, return_rvec=True,
                         return_drvec=True)
    for a, b in zip(z["seds_flux"], got):
        assert np.max(np.abs(a - b) / np.abs(a)) < 1e-15
    n = len(z["sar_cov"])
    sar = utils.draw_sar(np.ones(n), np.full(n, 0.1), np.full(n, 3.3), z["sar_cov"],
                         ndraws=40, rstate=np.random.RandomState(4))
    assert all(np.array_equal(a, b) for a, b in zip(z["sar"], sar))
    for dp, key in ((True, "pl_dp"), (False, "pl_g")):
        got = utils.phot_loglike(z["pl_d"], z["pl_e"], z["pl_m"], z["pl_models"],
                                 dim_prior=dp)
        assert np.max(np.abs(got - z[key])) < 1e-12


def test_photometric_offsets_match_reference():
    """`photometric_offsets` (reference utils.py:1218-1400) incl. its RNG call
    order, against a vector generated from the upstream code."""
    import os as _os
    z = np.load(_os.path.join(_os.path.dirname(__file__), "golden", "consumers.npz"))
    r, re_, n = utils.photometric_offsets(
        z["po_phot"], z["po_err"], z["po_mask"], z["po_models"], z["po_idxs"],
        z["po_reds"], z["po_dreds"], z["po_dists"], sel=z["po_sel"],
        weights=z["po_weights"], mask_fit=z["po_mask_fit"], Nmc=20,
        old_offsets=z["po_old"], prior_mean=np.ones(6), prior_std=np.full(6, 0.05),
        verbose=False, rstate=np.random.RandomState(5))
    assert np.array_equal(n, z["po_nratio"])
    assert np.max(np.abs(r - z["po_ratios"])) < 1e-12
    assert np.max(np.abs(re_ - z["po_ratios_err"])) < 1e-12


def test_results_file_resume(tmp_path):
    path = os.path.join(str(tmp_path), "r.h5")
    row = lambda k: (np.full(3, k), np.ones(3), np.ones(3), np.ones(3),
                     np.ones((3, 3, 3)), 7, np.ones(3), -2.5, 3.5, np.ones(3),
                     np.ones(3), np.ones(3), np.full(3, float(k)))
    rf = h5io.ResultsFile(path, 5, 3, np.arange(5), True, flush_every=1)
    rf.write_row(0, row(10))
    rf.write_row(1, row(11))
    rf.close()                      # "interrupted" after two objects
    rf = h5io.ResultsFile.resume(path, 5, 3, True)
    assert list(rf.todo) == [2, 3, 4]
    for k in rf.todo:
        rf.write_row(int(k), row(20 + k))
    rf.close()
    idx = h5io.read_dataset(path, "model_idx")
    assert [int(r[0]) for r in idx] == [10, 11, 22, 23, 24]
    assert list(h5io.read_dataset(path, "samps_logp")[:, 0]) == [10., 11., 22., 23., 24.]
    with pytest.raises(ValueError):
        h5io.ResultsFile.resume(path, 6, 3, True)


def test_photometric_offsets_vectorised_equals_call_by_call_form():
    """The bootstrap of `photometric_offsets` is vectorised (one `random_sample(n)` and a
    row-wise searchsorted instead of n `RandomState.choice` calls per round, objects with
    the same band pattern through one `phot_loglike` evaluation): bit-identical to the
    reference's call-by-call form (utils.py:1330-1385) on a case with mixed band masks,
    zero-weight objects, a band outside the fit and both likelihood forms."""
    from scipy.special import logsumexp
    rng = np.random.RandomState(0)
    Nobj, Ns, Nf, Nm = 300, 40, 6, 200
    models = np.zeros((Nm, Nf, 3))
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
    mask_fit = np.array([1, 1, 0, 1, 1, 1], bool)
    old = np.linspace(0.98, 1.02, Nf)

    def call_by_call(dim_prior, rstate, Nmc):
        seds = utils.get_seds(models[idxs.ravel()], av=reds.ravel(), rv=dreds.ravel(),
                              return_flux=True)
        seds = (seds / dists.ravel()[:, None] ** 2).reshape(Nobj, Ns, Nf)
        ratios, errs, nr = np.ones(Nf), np.zeros(Nf), np.zeros(Nf, dtype=int)
        nbands = mask.sum(axis=1)
        usable = w.sum(axis=1) > 0
        for b in range(Nf):
            s = np.where(mask[:, b] & usable & (nbands > 3 + (1 if mask_fit[b] else 0)))[0]
            n = nr[b] = len(s)
            ratio = seds[s, :, b] / phot[s, None, b]
            if mask_fit[b]:
                others = mask[s].copy()
                others[:, b] = False
                lnl = np.array([utils.phot_loglike(p * old, e * old, m, sd, dim_prior=dim_prior)
                                for p, e, m, sd in zip(phot[s], err[s], others, seds[s])])
                wt = np.exp(lnl - logsumexp(lnl, axis=1)[:, None])
            else:
                wt = np.ones((n, Ns))
            wt = wt * w[s]
            wt /= wt.sum(axis=1)[:, None]
            wo = np.array(w[s].sum(axis=1) > 0, dtype=float)
            wo /= wo.sum()
            meds = np.empty(Nmc)
            for j in range(Nmc):
                ridx = rstate.choice(n, size=n, p=wo)
                midx = [rstate.choice(Ns, p=x) for x in wt[ridx]]
                meds[j] = np.median(ratio[ridx, midx])
            ratios[b], errs[b] = np.median(meds), np.std(meds)
        return ratios, errs, nr

    for dp in (True, False):
        a = call_by_call(dp, np.random.RandomState(3), 8)
        b = utils.photometric_offsets(phot, err, mask, models, idxs, reds, dreds, dists,
                                      weights=w, mask_fit=mask_fit, Nmc=8, old_offsets=old,
                                      dim_prior=dp, verbose=False,
                                      rstate=np.random.RandomState(3))
        for x, y in zip(a, b):
            assert np.array_equal(x, y), dp


def _rows(n, ndraws, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        v = rng.normal(size=ndraws)
        out.append((rng.randint(0, 1000, ndraws), v, v + 1, v + 2, rng.normal(size=(ndraws, 3, 3)),
                    6 + i % 3, v - 5, float(i), 2.5 * i, v * 2, v * 3, v * 4,
                    np.where(v > 1, -1e300, v)))       # -1e300 -> -inf in float32, like h5py
    return out


def test_results_writer_async_equals_sync_any_row_order(tmp_path):
    """The background writer (ring of preallocated blocks filled in place) writes the file
    a synchronous writer does, byte for byte -- rows arriving in catalogue order, in a
    scrambled order (sharded / resumed runs) and as whole blocks -- and `h5dump -H` shows the
    reference's layout."""
    import filecmp
    import subprocess
    from brutus_amd import h5io
    n, nd = 1000, 7
    rows = _rows(n, nd)
    lab = np.arange(n)
    paths = []
    for tag, async_io, order in (("sync", False, np.arange(n)), ("async", True, np.arange(n)),
                                 ("scrambled", True, np.random.RandomState(1).permutation(n))):
        p = str(tmp_path / (tag + ".h5"))
        rf = h5io.ResultsFile(p, n, nd, lab, True, flush_every=64, async_io=async_io)
        for i in order:
            rf.write_row(int(i), rows[i])
        rf.close()
        paths.append(p)
    # packed blocks (parallel.fit_sharded's hand-off) through write_block
    rowdt, positions = h5io.ResultsFile.row_dtype(nd, True)
    p = str(tmp_path / "blocks.h5")
    rf = h5io.ResultsFile(p, n, nd, lab, True, flush_every=64)
    for a in range(0, n, 128):
        blk = np.zeros(min(128, n - a), dtype=rowdt)
        with np.errstate(over="ignore"):
            for j in range(len(blk)):
                for name, pos in positions:
                    blk[name][j] = rows[a + j][pos]
        rf.write_block(a, {name: blk[name] for name, _ in positions})
    rf.close()
    paths.append(p)
    for q in paths[1:]:
        assert filecmp.cmp(paths[0], q, shallow=False), q
    assert np.array_equal(h5io.read_dataset(paths[1], "obj_log_evid"), np.arange(n, dtype=np.float32))
    assert np.isneginf(h5io.read_dataset(paths[1], "samps_logp")).any()
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        hdr = subprocess.run([h5dump, "-H", paths[1]], stdout=subprocess.PIPE).stdout.decode()
        for name, shape in (("model_idx", "( %d, %d )" % (n, nd)), ("ml_cov_sar", "( %d, %d, 3, 3 )" % (n, nd)),
                            ("obj_Nbands", "( %d )" % n)):
            assert 'DATASET "%s"' % name in hdr and shape in hdr
        assert "H5T_STD_I32LE" in hdr and "H5T_IEEE_F32LE" in hdr and "H5T_STD_I16LE" in hdr


def test_results_writer_reports_errors_of_the_background_thread(tmp_path):
    """A failure inside the writer thread surfaces at the next call of the fit loop, it is
    not swallowed."""
    from brutus_amd import h5io
    p = str(tmp_path / "err.h5")
    rf = h5io.ResultsFile(p, 10, 3, None, False, flush_every=2)
    rows = _rows(10, 3)
    rf.write_row(0, rows[0])
    rf.file.write_rows = lambda *a, **k: (_ for _ in ()).throw(OSError("disk full"))
    rf.write_row(1, rows[1])            # hands a block to the writer, which fails
    with pytest.raises(OSError):
        rf.flush()
    # sticky: a caller that swallowed the first report still cannot finish cleanly, and the
    # rows that never reached the file are named
    with pytest.raises(OSError, match="disk full") as ei:
        rf.write_row(2, rows[2])
    assert "not written" in str(ei.value.args) and (0, 2) in rf.dropped
    rf._cur = None
    with pytest.raises(OSError):
        rf.close()
    assert rf.file is None


@pytest.mark.parametrize("mode", ["rows", "block"])
def test_results_writer_killed_between_datasets_keeps_the_sentinel(tmp_path, mode):
    """`resume()` trusts `model_idx != -99` alone (reference fitting.py:1635), so within a
    block the sentinel dataset must reach the disk LAST: a writer that dies after any of
    the other datasets leaves the block's rows marked unfitted."""
    from brutus_amd import h5io
    names = list(h5io.ResultsFile.row_dtype(3, True)[0].names)
    rows = _rows(6, 3)
    for die_after in range(len(names)):
        p = str(tmp_path / ("kill_%s_%d.h5" % (mode, die_after)))
        rf = h5io.ResultsFile(p, 6, 3, None, True, flush_every=2, async_io=False)
        rf.write_row(0, rows[0])
        rf.write_row(1, rows[1])            # block 0 complete and written
        real = rf.file.write_rows
        seen = []

        def dying(name, start, arr, real=real, seen=seen):
            if len(seen) == die_after:
                raise KeyboardInterrupt("killed")
            seen.append(name)
            return real(name, start, arr)
        rf.file.write_rows = dying
        with pytest.raises(KeyboardInterrupt):
            if mode == "rows":
                rf.write_row(2, rows[2])
                rf.write_row(3, rows[3])
            else:
                dt, positions = h5io.ResultsFile.row_dtype(3, True)
                blk = np.zeros(2, dtype=dt)
                for name, pos in positions:
                    for j in range(2):
                        blk[name][j] = rows[2 + j][pos]
                rf.write_block(2, {name: blk[name] for name in dt.names})
        assert seen == [n for n in names if n != "model_idx"][:die_after] + \
            (["model_idx"] if die_after == len(names) else [])
        rf.file.write_rows = real
        rf.file.close()
        rf.file = None
        todo = h5io.ResultsFile.resume(p, 6, 3, True)
        assert list(todo.todo) == [2, 3, 4, 5], (die_after, list(todo.todo))
        todo.close()


def test_los_tables_pad_shorter_profiles():
    """Per-batch line-of-sight tables: profiles of different lengths are padded by their
    last node, which leaves numpy.interp (and the device's interpolation) unchanged."""
    from brutus_amd import pdf
    prof = {0: (np.array([0.1, 1., 3.]), np.array([0.1, 0.5, 0.9]), np.array([0.1, 0.1, 0.2])),
            1: (np.array([0.2, 2.]), np.array([0.3, np.nan]), np.array([0.1, 0.1]))}
    q = lambda c: prof[int(c[0])]
    los, ok = pdf.los_tables(q, np.array([[0., 0.], [1., 0.]]))
    assert los.shape == (2, 3, 3) and ok.tolist() == [1, 0]
    assert np.array_equal(los[1, 0], [0.2, 2., 2.])
    d = np.linspace(0.05, 5., 50)
    assert np.array_equal(np.interp(d, los[0, 0], los[0, 1]), np.interp(d, *prof[0][:2]))
    # one node (or scalars): a constant profile for numpy.interp, and so for the table
    for one in ((np.array([1.]), np.array([0.7]), np.array([0.2])), (1., 0.7, 0.2)):
        los, ok = pdf.los_tables(lambda c: one, np.zeros((1, 2)))
        assert ok.tolist() == [1] and los.shape == (1, 3, 2)
        assert np.array_equal(np.interp(d, los[0, 0], los[0, 1]), np.interp(d, [1.], [0.7]))
    with pytest.raises(ValueError):
        pdf.los_tables(lambda c: (np.array([1., 2.]), np.array([1.]), np.array([1.])), np.zeros((1, 2)))
