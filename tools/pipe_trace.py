import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, tempfile
from brutus_amd import fitting, synth
from brutus_amd.galprior import gal_lnprior
ev = []
ob, oe = fitting._Engine.post_numpy_begin, fitting._Engine.post_numpy_end
orr = fitting._Engine.records_device
def tb(self, *a, **k):
    t = time.perf_counter(); r = ob(self, *a, **k); ev.append(("begin", t, time.perf_counter())); return r
def te(self, *a, **k):
    t = time.perf_counter(); r = oe(self, *a, **k); ev.append(("end", t, time.perf_counter())); return r
def tr(self, *a, **k):
    t = time.perf_counter(); r = orr(self, *a, **k); ev.append(("scan", t, time.perf_counter())); return r
fitting._Engine.post_numpy_begin = tb; fitting._Engine.post_numpy_end = te; fitting._Engine.records_device = tr
models, labels, lmask = synth.make_mist_like_grid(750000, 12)
st = synth.make_stars(models, 1024, seed=4242, with_parallax=False)
bf = fitting.BruteForce(models, labels, lmask); bf.batch_size = 128
def run():
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(st["flux"], st["err"], st["mask"], np.arange(1024), os.path.join(tmp, "x"), data_coords=st["coords"],
               lngalprior=gal_lnprior, rv_gauss=(3.32, 1e-6), rstate=np.random.RandomState(862), verbose=False)
        return time.perf_counter() - t0
run(); ev.clear(); t0 = time.perf_counter(); dt = run()
print("fit %.3f s" % dt)
for name, a, b in sorted(ev, key=lambda e: e[1]):
    print("%-6s %7.1f -> %7.1f  (%.1f ms)" % (name, (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3))
