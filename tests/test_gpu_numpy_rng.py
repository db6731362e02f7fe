"""GPU: numpy's own random stream on the device (csrc/mt_kernels.hpp).

The reference draws from a legacy `numpy.random.RandomState` (MT19937 + polar
Box-Muller with a cached deviate + `choice` via random_sample).  These
tests pin (1) the stream kernel against numpy itself, word for word, including the
state it hands back, and (2) `BruteForce._fit` with a plain `RandomState` / the global
`numpy.random` / per-object seeds running `lnpost` on the device: resampled indices
bit-exact against the reference-generated goldens and against the oracle."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, galprior, relerr

pytestmark = pytest.mark.gpu


def _walk(states, nnorm, nuni):
    """brutus_debug_mt_stream -> (list of normals per object, uniforms (nobj, nuni))"""
    import torch
    from brutus_amd import _lib
    L = _lib.lib()
    nobj = len(nnorm)
    offs = np.concatenate([[0], np.cumsum(((np.asarray(nnorm) + 1) & ~1) + 2)])
    z = torch.zeros(int(offs[-1]) + 8, dtype=torch.float64, device="cuda")
    u = torch.zeros((nobj, max(nuni, 1)), dtype=torch.float64, device="cuda")
    nn = np.ascontiguousarray(nnorm, dtype=np.int64)
    _lib.check(L.brutus_debug_mt_stream(nobj, states.shape[0], states.ctypes.data, nn.ctypes.data,
                                        nuni, z.data_ptr(), u.data_ptr(), None))
    z = z.cpu().numpy()
    return [z[offs[o]:offs[o] + nnorm[o]] for o in range(nobj)], u.cpu().numpy()[:, :nuni]


def test_mt_stream_matches_numpy_word_for_word():
    from brutus_amd.rng import state_to_words, words_to_state
    rng = np.random.RandomState(77)
    for trial in range(6):
        rs = np.random.RandomState(1000 + trial)
        # arbitrary position inside a block, odd word offsets, a cached deviate pending
        rs.randint(0, 2 ** 31, size=int(rng.randint(0, 700)))
        if trial % 2:
            rs.normal(size=3)
        nnorm = [0, 1, 2, 7, 150 * 37, 75 * 1001 + 1, 0, 3 * 50 * 4000][: 3 + trial]
        nuni = [0, 1, 500, 40, 7, 500][trial]
        ref = np.random.RandomState()
        ref.set_state(rs.get_state())
        want_z, want_u = [], []
        for n in nnorm:
            want_z.append(ref.normal(size=n))
            want_u.append(ref.random_sample(nuni))
        st = state_to_words(rs.get_state()).reshape(1, -1).copy()
        got_z, got_u = _walk(st, nnorm, nuni)
        for o, n in enumerate(nnorm):
            assert np.array_equal(got_u[o], want_u[o]), (trial, o)
            if n:
                err = np.abs(got_z[o] - want_z[o]) / np.abs(want_z[o])
                # same accept / reject decisions (a wrong one shifts every later deviate);
                # values agree to the last bit or two (ln is not correctly rounded anywhere)
                assert err.max() < 1e-15, (trial, o, err.max())
        # the state handed back continues numpy's own stream
        back = np.random.RandomState()
        back.set_state(words_to_state(st[0]))
        a, b = back.random_sample(9), ref.random_sample(9)
        assert np.array_equal(a, b), trial
        za, zb = back.normal(size=5), ref.normal(size=5)
        assert relerr(zb, za) < 1e-15, trial


def test_mt_stream_per_object_streams():
    from brutus_amd.rng import state_to_words
    nnorm = [150 * 200, 0, 151, 75 * 333]
    st = np.stack([state_to_words(np.random.RandomState(500 + i).get_state()) for i in range(4)])
    got_z, got_u = _walk(st, nnorm, 20)
    for i, n in enumerate(nnorm):
        ref = np.random.RandomState(500 + i)
        z = ref.normal(size=n)
        u = ref.random_sample(20)
        assert np.array_equal(got_u[i], u)
        if n:
            assert relerr(z, got_z[i]) < 1e-15


NAMES = ("sidxs scales avs rvs cov Ndim lnprob levid chi2min dists reds dreds "
         "logwts").split()


def _bf(seed=31, nmodel=6000):
    from brutus_amd import fitting, synth
    from oracle import brutus_oracle as O
    models, labels, lmask = synth.make_mist_like_grid(nmodel, 8, seed=seed)
    st = synth.make_stars(models, 9, seed=seed + 1)
    st["mask"][1, 2] = False
    BF = fitting.BruteForce(models, labels, lmask)
    return BF, models, labels, st, O.static_lnprior(labels, lmask)


def _device_path_taken(BF, monkeypatch_calls):
    return monkeypatch_calls["n"] > 0


def test_fit_with_shared_randomstate_on_device_vs_oracle(monkeypatch):
    """One sequential RandomState over all objects (the reference's semantics), batches of
    4: device `lnpost` with numpy's stream == oracle with the same RandomState, and the
    caller's generator ends in the same state."""
    from brutus_amd import fitting
    from brutus_amd.galprior import gal_lnprior
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _bf()
    BF.batch_size = 4
    calls = {"n": 0}
    orig = fitting._Engine.post_batch_device

    def spy(self, *a, **k):
        if k.get("np_states") is not None:
            calls["n"] += 1
        return orig(self, *a, **k)
    monkeypatch.setattr(fitting._Engine, "post_batch_device", spy)
    # ... or through its two-phase form (phase 2 of a batch beside phase 1 of the next)
    orig_begin = fitting._Engine.post_numpy_begin

    def spy_begin(self, *a, **k):
        ok = orig_begin(self, *a, **k)
        calls["n"] += 1 if ok else 0
        return ok
    monkeypatch.setattr(fitting._Engine, "post_numpy_begin", spy_begin)
    for pipeline in (True, False):
        BF.post_pipeline = pipeline
        calls["n"] = 0
        rs = np.random.RandomState(2024)
        rs.normal(size=1)               # a cached deviate is pending when the fit starts
        ro = np.random.RandomState(2024)
        ro.normal(size=1)
        dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                           parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                           lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60,
                           rstate=rs))
        assert calls["n"] == 3       # 9 objects in batches of 4: all through the device stage
        for i in range(len(dev)):
            ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                             labels, st["coords"][i], st["parallax"][i],
                             st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=20, Ndraws=60)
            assert np.array_equal(dev[i][0], ref[0]), (pipeline, i)
            for n, a, b in zip(NAMES[1:], ref[1:], dev[i][1:]):
                assert relerr(a, b) < 1e-8, (pipeline, i, n, relerr(a, b))
        assert np.array_equal(rs.random_sample(5), ro.random_sample(5))
    BF.post_pipeline = True
    assert relerr(ro.normal(size=3), rs.normal(size=3)) < 1e-15


def test_fit_with_global_numpy_random_on_device():
    """rstate=None: the reference draws from the global numpy.random (fitting.py:937-944;
    the notebooks call np.random.seed first)."""
    from brutus_amd.galprior import gal_lnprior
    from oracle import brutus_oracle as O
    BF, models, labels, st, lnprior = _bf(seed=41)
    np.random.seed(862)
    dev = list(BF._fit(st["flux"][:3], st["err"][:3], st["mask"][:3], parallax=st["parallax"][:3],
                       parallax_err=st["parallax_err"][:3], Nmc_prior=15, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"][:3], Ndraws=40))
    after = np.random.random_sample(4)
    ro = np.random.RandomState(862)
    for i in range(3):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                         labels, st["coords"][i], st["parallax"][i],
                         st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=15, Ndraws=40)
        assert np.array_equal(dev[i][0], ref[0]), i
        assert relerr(ref[6], dev[i][6]) < 1e-8
    assert np.array_equal(after, ro.random_sample(4))


def test_device_numpy_rng_equals_host_stage_full_size():
    """750k x 12 (the bench's grid and stars): `_fit` with per-object numpy seeds, device
    stage vs the host stage of the same package (numpy itself drawing), which the goldens
    pin to the reference."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    models, labels, lmask = synth.make_mist_like_grid(750000, 12)
    st = synth.make_stars(models, 3, seed=5)
    BF = fitting.BruteForce(models, labels, lmask)
    lnprior = BF._setup(st["flux"], st["err"], st["mask"], None, data_coords=st["coords"],
                        lngalprior=gal_lnprior)[5]
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], Nmc_prior=50,
              lnprior=lnprior, lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=250,
              seed0=4242)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], **kw))
    BF.device_numpy_rng = False
    host = list(BF._fit(st["flux"], st["err"], st["mask"], **kw))
    for i in range(3):
        assert np.array_equal(dev[i][0], host[i][0]), i
        for n, a, b in zip(NAMES[1:], host[i][1:], dev[i][1:]):
            assert relerr(a, b) < 1e-6, (i, n, relerr(a, b))


def test_mt_stream_many_workgroups_equals_one_and_numpy():
    """Streams long enough to be cut into many sub-streams (jump-ahead, two levels: more
    than 128 sub-streams of 2 096 640 words): the parallel walk must give the same normals,
    uniforms and final state as the single-workgroup walk, and as numpy."""
    from brutus_amd.rng import state_to_words, words_to_state
    cases = [([21000000, 0, 3, 5000001, 150], 500, 1), ([75 * 140000, 75 * 90001], 500, 2),
             ([120000000, 7], 20, 1)]
    for nnorm, nuni, nstream in cases:
        outs = {}
        for par in ("1", "0"):
            os.environ["BRUTUS_MT_PARALLEL"] = par
            if nstream == 1:
                rs = np.random.RandomState(99)
                rs.randint(0, 2 ** 31, size=101)
                rs.normal(size=1)
                st = state_to_words(rs.get_state()).reshape(1, -1).copy()
            else:
                st = np.stack([state_to_words(np.random.RandomState(700 + i).get_state())
                               for i in range(len(nnorm))])
            z, u = _walk(st, nnorm, nuni)
            outs[par] = (z, u, st.copy())
        os.environ.pop("BRUTUS_MT_PARALLEL", None)
        for o in range(len(nnorm)):
            assert np.array_equal(outs["1"][0][o], outs["0"][0][o]), (nnorm, o)
            assert np.array_equal(outs["1"][1][o], outs["0"][1][o]), (nnorm, o)
        for g in range(nstream if nstream > 1 else 1):
            a = np.random.RandomState()
            a.set_state(words_to_state(outs["1"][2][g]))
            b = np.random.RandomState()
            b.set_state(words_to_state(outs["0"][2][g]))
            assert np.array_equal(a.random_sample(7), b.random_sample(7))
            assert np.array_equal(a.normal(size=3), b.normal(size=3))
        # and numpy itself
        if nstream == 1:
            ref = np.random.RandomState(99)
            ref.randint(0, 2 ** 31, size=101)
            ref.normal(size=1)
            for o, n in enumerate(nnorm):
                zr = ref.normal(size=n)
                ur = ref.random_sample(nuni)
                assert np.array_equal(outs["1"][1][o], ur), (nnorm, o)
                if n:
                    assert relerr(zr, outs["1"][0][o]) < 1e-15


def test_small_normal_buffer_falls_back_to_groups(monkeypatch):
    """A normal buffer that cannot hold a batch as ONE group: phase 1 of the two-phase call
    declines (nothing consumed), the whole-call form serves the objects in groups -- and,
    with no room for the parallel walk's scratch, through the one-workgroup stream walker;
    results and the generator's end state still equal the oracle's."""
    from brutus_amd import fitting
    from brutus_amd.galprior import gal_lnprior
    from oracle import brutus_oracle as O
    monkeypatch.setenv("BRUTUS_AMD_ZBUF_GB", "0.003")          # ~4e5 doubles
    BF, models, labels, st, lnprior = _bf(seed=57)
    BF.batch_size = 5
    seen = {"begin": [], "whole": 0}
    ob, ow = fitting._Engine.post_numpy_begin, fitting._Engine.post_batch_device

    def spy_begin(self, *a, **k):
        ok = ob(self, *a, **k)
        seen["begin"].append(ok)
        return ok

    def spy_whole(self, *a, **k):
        seen["whole"] += 1
        return ow(self, *a, **k)
    monkeypatch.setattr(fitting._Engine, "post_numpy_begin", spy_begin)
    monkeypatch.setattr(fitting._Engine, "post_batch_device", spy_whole)
    rs, ro = np.random.RandomState(5), np.random.RandomState(5)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], Nmc_prior=20, lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=60, rstate=rs))
    # (the first batch of five does not fit; the last four objects do and take the
    # two-phase form: both forms in one run, in stream order)
    assert seen["begin"] and seen["begin"][0] is False and seen["whole"] >= 1
    for i in range(len(dev)):
        ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior, labels,
                         st["coords"][i], st["parallax"][i], st["parallax_err"][i], ro,
                         gal_lnprior, Nmc_prior=20, Ndraws=60)
        assert np.array_equal(dev[i][0], ref[0]), i
        for n, a, b in zip(NAMES[1:], ref[1:], dev[i][1:]):
            assert relerr(a, b) < 1e-8, (i, n, relerr(a, b))
    assert np.array_equal(rs.random_sample(5), ro.random_sample(5))


def test_full_size_fit_with_numpy_stream_vs_oracle():
    """The bench's size and fit()'s defaults (750k x 12, Nmc_prior=50, Ndraws=250: ~10^5
    kept models and ~2 10^7 normals per object, i.e. many jump-ahead sub-streams) with ONE
    numpy RandomState over three objects in batches of two (two-phase form, a stream that
    continues across batches): resampled indices bit-exact against the oracle -- C `loglike`
    followed by the numpy `lnpost` drawing from numpy itself -- and the same end state."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from oracle import brutus_oracle as O
    from oracle import c_oracle
    models, labels, lmask = synth.make_mist_like_grid(750000, 12)
    st = synth.make_stars(models, 3, seed=2)
    lnprior = O.static_lnprior(labels, lmask)
    BF = fitting.BruteForce(models, labels, lmask)
    BF.batch_size = 2
    rs, ro = np.random.RandomState(99), np.random.RandomState(99)
    dev = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                       parallax_err=st["parallax_err"], lnprior=lnprior,
                       lngalprior=gal_lnprior, data_coords=st["coords"], rstate=rs,
                       Nmc_prior=50, Ndraws=250))
    py_loglike = O.loglike
    O.loglike = lambda *a, return_vals=True, **k: c_oracle.loglike(*a, **k)
    try:
        for i in range(3):
            ref = O.fit_star(st["flux"][i], st["err"][i], st["mask"][i], models, lnprior,
                             labels, st["coords"][i], st["parallax"][i],
                             st["parallax_err"][i], ro, gal_lnprior, Nmc_prior=50, Ndraws=250)
            assert np.array_equal(dev[i][0], ref[0]), "resampled indices, object %d" % i
            for n, a, b in zip(NAMES[1:], ref[1:], dev[i][1:]):
                assert relerr(a, b) < 1e-6, (i, n, relerr(a, b))
    finally:
        O.loglike = py_loglike
    assert np.array_equal(rs.random_sample(5), ro.random_sample(5))


_ONE_WALK_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
from oracle import brutus_oracle as O
models, labels, lmask = synth.make_mist_like_grid(120000, 8, seed=7)
st = synth.make_stars(models, 6, seed=8)
BF = fitting.BruteForce(models, labels, lmask)
BF.batch_size = 3
lnprior = O.static_lnprior(labels, lmask)
rs = np.random.RandomState(99)
rs.normal(size=1)                       # a cached deviate is pending when the fit starts
out = list(BF._fit(st["flux"], st["err"], st["mask"], parallax=st["parallax"],
                   parallax_err=st["parallax_err"], Nmc_prior=50, lnprior=lnprior,
                   lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=100, rstate=rs))
np.savez(sys.argv[2], state=rs.get_state()[1], pos=rs.get_state()[2],
         **{"o%d_%d" % (i, k): np.asarray(v) for i, o in enumerate(out) for k, v in enumerate(o)})
"""


def test_one_walk_of_the_stream_equals_two_walks_bit_for_bit(tmp_path):
    """BRUTUS_MT_ONE_WALK=1 (pass 1 leaves the normals as pairs per sub-stream, the consumers
    read through segment lists) against =0 (second walk, flat normals): every output of
    `_fit` and the generator state are identical bit for bit.  120k models: objects of
    10^4..10^5 kept models = 10^6..10^7 normals each span several sub-streams (4.1e5 pairs
    each), batches of 3 share one stream, a cached deviate is pending at the start."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, BRUTUS_MT_ONE_WALK=mode)
        f = str(tmp_path / ("walk%s.npz" % mode))
        subprocess.run([sys.executable, "-c", _ONE_WALK_SCRIPT, root, f], check=True, env=env,
                       timeout=600)
        res[mode] = np.load(f)
    assert set(res["0"].files) == set(res["1"].files) and len(res["0"].files) > 20
    nsel = 0
    for k in res["0"].files:
        a, b = res["0"][k], res["1"][k]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), k
    for i in range(6):
        nsel = max(nsel, len(np.unique(res["1"]["o%d_0" % i])))
    assert nsel > 10            # (the draws are spread over many models: broad posteriors)


def test_pipelined_numpy_fit_can_be_abandoned_and_repeated():
    """The `_fit` generator of the two-phase numpy-stream pipeline (phase 2 of a batch held back
    until the next batch's jump-ahead is through) is closed after 1, 9, 17, 33 objects -- in
    the first batch, at a batch boundary, inside a later batch -- and run again: what it
    yielded equals the full run, nothing hangs, and a complete run afterwards is identical."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    models, labels, lmask = synth.make_mist_like_grid(60000, 8, seed=3)
    st = synth.make_stars(models, 40, seed=4)
    bf = fitting.BruteForce(models, labels, lmask)
    bf.batch_size = 8
    lnprior = bf._setup(st["flux"], st["err"], st["mask"], None, data_coords=st["coords"],
                        lngalprior=gal_lnprior)[5]
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], Nmc_prior=20,
              lnprior=lnprior, lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=50)
    full = list(bf._fit(st["flux"], st["err"], st["mask"], rstate=np.random.RandomState(1), **kw))
    assert len(full) == 40
    for stop in (1, 9, 17, 33):
        g = bf._fit(st["flux"], st["err"], st["mask"], rstate=np.random.RandomState(1), **kw)
        got = [next(g) for _ in range(stop)]
        g.close()
        assert all(np.array_equal(a[0], b[0]) for a, b in zip(got, full)), stop
    again = list(bf._fit(st["flux"], st["err"], st["mask"], rstate=np.random.RandomState(1), **kw))
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[6], b[6])
               for a, b in zip(again, full))


def test_pipeline_with_a_late_phase_two_equals_the_sequential_form():
    """Phase 2 of batch k reads the scan records of batch k while batch k + 1 is walked and
    batch k + 2 is scanned; the engine that scans batch k + 3 is the one that holds batch k.
    With phase 2 artificially late (test hook) every hand-over in that chain is exercised:
    the pipelined run must equal the run without the pipeline, row for row."""
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    models, labels, lmask = synth.make_mist_like_grid(60000, 8, seed=5)
    st = synth.make_stars(models, 48, seed=6)
    bf = fitting.BruteForce(models, labels, lmask)
    bf.batch_size = 6
    lnprior = bf._setup(st["flux"], st["err"], st["mask"], None, data_coords=st["coords"],
                        lngalprior=gal_lnprior)[5]
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"], Nmc_prior=20,
              lnprior=lnprior, lngalprior=gal_lnprior, data_coords=st["coords"], Ndraws=50)
    bf.post_pipeline = False
    ref = list(bf._fit(st["flux"], st["err"], st["mask"], rstate=np.random.RandomState(9), **kw))
    bf.post_pipeline = True
    for delay in (0., 0.03):
        bf._test_phase2_delay = delay
        got = list(bf._fit(st["flux"], st["err"], st["mask"], rstate=np.random.RandomState(9), **kw))
        assert len(got) == len(ref) == 48
        for i, (a, b) in enumerate(zip(got, ref)):
            assert np.array_equal(a[0], b[0]), (delay, i)
            for x, y in zip(a[1:], b[1:]):
                assert np.array_equal(np.asarray(x), np.asarray(y)), (delay, i)
