"""Same-box timing of the float32 pass alone (brutus_debug_pre32_time): one fit batch of the bench
workload sets the workspace up, then each form is re-launched `reps` times.
    python tools/pre32_forms.py [forms, e.g. 0,1,2] [configs, e.g. 2,3] [nstar] [grid: mist|survey|sharp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from brutus_amd import _lib, fitting, synth  # noqa: E402

forms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
cfgs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,3").split(",")]
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
gname = sys.argv[4] if len(sys.argv) > 4 else "mist"
L = _lib.lib()
mk = {"mist": synth.make_mist_like_grid, "survey": synth.make_grid, "sharp": synth.make_sharp_grid}[gname]
models, _, _ = mk(750000, 12)
grid = fitting.DeviceGrid(models, device="cuda:0")
for cfg in cfgs:
    kw = dict(rvlim=(3.32, 3.32)) if cfg == 2 else dict()
    st = synth.make_stars(models, S, seed={2: 1, 3: 2}[cfg], with_parallax=cfg == 3)
    params = fitting._make_params((0., 20.), (0., 1e6), kw.get("rvlim", (1., 8.)), (3.32, 0.18),
                                  3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    eng = fitting._Engine(grid, max_batch=S, mem_budget=64e9)
    eng._rec_bufs = eng._record_buffers(max(32 << 20, int(S * 750000 * 0.62)))
    up = eng._upload(st["flux"], st["err"], st["mask"], st["parallax"] if cfg == 3 else None,
                     st["parallax_err"] if cfg == 3 else None)
    eng.fit_batch_device(*up, params)
    torch.cuda.synchronize()
    ws = eng._workspace(S)
    for rep in range(2):
        for form in forms:
            ms = C.c_float(0.)
            _lib.check(L.brutus_debug_pre32_time(ws.data_ptr(), ws.numel(), grid.soa.data_ptr(), grid.nmodel,
                                                 grid.nfilt, S, params, form, 20, C.byref(ms), None))
            print("grid %-6s cfg %d  %3d stars  form %d  rep %d   %.4f ms" % (gname, cfg, S, form, rep, ms.value),
                  flush=True)
