"""Work-item balance of the flux phase over the 8 XCDs (GPU box): survivors per (star, chunk)
from the tag plane of one bench batch -> items per chunk -> per-XCD totals under the static
ownership rule (chunk c -> XCD c % 8).   python tools/chunk_balance.py [config] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from brutus_amd import _lib, fitting, synth

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = _lib.lib()
dev = torch.device("cuda:0")
nmodel, nfilt = 750000, 12
models, _, _ = synth.make_mist_like_grid(nmodel, nfilt)
grid = fitting.DeviceGrid(models, device=dev)
for seed in (1, 7, 8):
    st = synth.make_stars(models, B, seed=seed if config == 2 else seed + 1, with_parallax=(config == 3))
    params = fitting._make_params((0., 20.), (0., 1e6), (3.32, 3.32) if config == 2 else (1., 8.),
                                  (3.32, 0.18), 3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    eng = fitting._Engine(grid, max_batch=B, mem_budget=64e9)
    up = eng._upload(st["flux"], st["err"], st["mask"], st["parallax"] if config == 3 else None,
                     st["parallax_err"] if config == 3 else None)
    bufs = eng._record_buffers(max(32 << 20, int(B * nmodel * 0.62)))
    eng.fit_batch_device(*up, params, buffers=bufs, grow=False)
    torch.cuda.synchronize()
    ws = eng._workspace(B)
    plane = torch.empty((B, nmodel), dtype=torch.int32, device=dev)
    _lib.check(L.brutus_debug_copy(ws.data_ptr(), ws.numel(), nmodel, nfilt, B, 3, plane.data_ptr(),
                                   plane.numel() * 4, None))
    torch.cuda.synchronize()
    surv = ((plane > 0) & (plane < 0x7F800000)).cpu().numpy()
    ntile = (nmodel + 255) // 256
    edges = [min(nmodel, 256 * (ntile * c // 64)) for c in range(65)]
    cnt = np.stack([surv[:, edges[c]:edges[c + 1]].sum(axis=1) for c in range(64)], axis=1)   # (B, 64)
    items = (cnt + 255) // 256
    per_chunk = items.sum(axis=0)
    per_xcd = np.array([per_chunk[x::8].sum() for x in range(8)])
    fill = cnt.sum() / (items.sum() * 256.)
    print("seed %d: survivors %d, items %d (lane fill %.3f); per XCD %s; max / mean = %.3f; per chunk max / mean = %.2f"
          % (seed, cnt.sum(), items.sum(), fill, per_xcd.tolist(), per_xcd.max() / per_xcd.mean(),
             per_chunk.max() / per_chunk.mean()))
    # greedy rebalancing: chunks sorted by load, each to the least loaded XCD
    load = np.zeros(8)
    for c in np.argsort(-per_chunk):
        load[np.argmin(load)] += per_chunk[c]
    print("   greedy assignment of chunks to XCDs: max / mean = %.3f" % (load.max() / load.mean()))
