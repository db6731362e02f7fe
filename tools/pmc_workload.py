#!/usr/bin/env python
"""Workload for the PMC passes: a few calibration streams of known size, then a
few bench steps.  Run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE` (the two do not fit one pass on gfx950)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from brutus_amd import _lib, fitting, synth  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = _lib.lib()
dev = torch.device("cuda:0")
n = 256 << 20      # 1 GiB read, 2 GiB written: well past the 256 MiB Infinity Cache
src = torch.rand(n, dtype=torch.float32, device=dev)
dst = torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(3):
    _lib.check(L.brutus_calibrate_traffic(src.data_ptr(), dst.data_ptr(), n, None))
torch.cuda.synchronize()
del src, dst
models, _, _ = synth.make_mist_like_grid(750000, 12)
grid = fitting.DeviceGrid(models, device=dev)
st = synth.make_stars(models, B, seed=1 if config == 2 else 2, with_parallax=(config == 3))
params = fitting._make_params((0., 20.), (0., 1e6), (3.32, 3.32) if config == 2 else (1., 8.),
                              (3.32, 0.18), 3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
eng = fitting._Engine(grid, max_batch=B, mem_budget=64e9)
up = eng._upload(st["flux"], st["err"], st["mask"],
                 st["parallax"] if config == 3 else None,
                 st["parallax_err"] if config == 3 else None)
bufs = eng._record_buffers(max(32 << 20, int(B * 750000 * 0.62)))
for _ in range(3):
    eng.fit_batch_device(*up, params, buffers=bufs, grow=False)
torch.cuda.synchronize()
print("pmc workload done: calibration n=%d (read %d B, write %d B), batch %d" % (n, 4 * n, 8 * n, B))
