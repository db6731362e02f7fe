import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, "tests")
from brutus_amd import _lib
from brutus_amd.galprior import _frame
from test_gpu_lnpost import _post_params
L = _lib.lib()
rng = np.random.RandomState(11)
n = 1 << 20
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
feh, loga = rng.uniform(-3, 0.6, n), rng.uniform(7.5, 10.2, n)
if len(sys.argv) > 1:
    feh[:] = -0.2; loga[:] = 9.5
tf, tl = t(feh), t(loga)
def both(d, coord):
    td, tc = t(d), t(np.array(coord))
    ref = torch.empty(n, dtype=torch.float64, device="cuda"); out = torch.empty(n, dtype=torch.float64, device="cuda")
    used = torch.zeros(n, dtype=torch.int32, device="cuda")
    _lib.check(L.brutus_debug_galprior_mc(_post_params(), n, td.data_ptr(), tc.data_ptr(), tf.data_ptr(), tl.data_ptr(), ref.data_ptr(), None))
    _lib.check(L.brutus_debug_galprior_sl(_post_params(), n, td.data_ptr(), tc.data_ptr(), tf.data_ptr(), tl.data_ptr(), out.data_ptr(), used.data_ptr(), None))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy(), used.cpu().numpy().astype(bool)
M, off = _frame("astropy", 8.2, 0.025)
for coord in ((204.7, -19.2), (0., 90.), (0., -90.), (33., 2.), (0.02, -0.01), (0., -0.17), (180., -5.), (90., -30.)):
    ell, b = np.deg2rad(coord)
    uz = (M @ np.array([np.cos(b) * np.cos(ell), np.cos(b) * np.sin(ell), np.sin(b)]))[2]
    dk = -off[2] / uz
    for name, d in (("sorted", np.sort(10. ** rng.uniform(-2.5, 2.3, n))), ("kink", np.sort(abs(dk) * (1. + rng.uniform(-0.02, 0.02, n)))),
                    ("unsorted", 10. ** rng.uniform(-3, 3, n))):
        ref, out, used = both(d, coord)
        fin = np.isfinite(ref) & np.isfinite(out)
        e = np.abs(out - ref); e[~fin] = 0
        k = int(np.argmax(e))
        print(coord, "dk %.4f" % dk, name, "worst %.2e at d=%.5g feh %.2f loga %.2f used %d ref %.4f; frac>1e-10: %.2e" % (e[k], d[k], feh[k], loga[k], used[k], ref[k], (e > 1e-10).mean()))
