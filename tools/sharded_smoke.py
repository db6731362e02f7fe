"""End-to-end check of `parallel.fit_sharded` with real kernels: N ranks (one
GPU each, or all on cuda:0 over gloo with BRUTUS_BENCH_ONE_DEVICE=1
BRUTUS_BENCH_BACKEND=gloo on a one-GPU box) fit shards of one catalogue with the
device `lnpost`; rank 0 checks the gathered HDF5 file against an unsharded run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29520 tools/sharded_smoke.py
"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from brutus_amd import fitting, h5io, parallel, synth
from brutus_amd.galprior import gal_lnprior


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("BRUTUS_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BRUTUS_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    nmodel, nfilt, nstar = 40000, 8, 37          # 37: uneven shards
    models, labels, lmask = synth.make_mist_like_grid(nmodel, nfilt, seed=5)
    st = synth.make_stars(models, nstar, seed=9)
    grid = fitting.DeviceGrid(models, device=dev) if rank == 0 else None
    grid = parallel.broadcast_grid(grid, nmodel, nfilt, dev, src=0)
    bf = fitting.BruteForce(models, labels, lmask)
    bf.use_device_grid(grid)
    bf.batch_size = 8
    kw = dict(parallax=st["parallax"], parallax_err=st["parallax_err"],
              data_coords=st["coords"], lngalprior=gal_lnprior, Nmc_prior=20, Ndraws=40)
    tmp = tempfile.mkdtemp() if rank == 0 else None
    box = [tmp]
    dist.broadcast_object_list(box, src=0)
    path = os.path.join(box[0], "sharded")
    writer = os.environ.get("SHARDED_WRITER", "rank0")       # "per_rank": part files + virtual index
    n = parallel.fit_sharded(bf, st["flux"], st["err"], st["mask"], np.arange(nstar), path,
                             seed0=500, writer=writer, **kw)
    lo, hi = parallel.shard_range(nstar, rank, world)
    assert n == hi - lo
    if rank == 0:
        f = path + ".h5"
        idx = h5io.read_dataset(f, "model_idx")
        post = h5io.read_dataset(f, "obj_log_post")
        dist_s = h5io.read_dataset(f, "samps_dist")
        # unsharded reference on this rank, same per-object seeds
        (d, e, m, _, coords, lnprior, lng, lnd, avg, wt, _) = bf._setup(
            st["flux"], st["err"], st["mask"], np.arange(nstar), parallax=st["parallax"],
            parallax_err=st["parallax_err"], data_coords=st["coords"], lngalprior=gal_lnprior)
        rows = list(bf._fit(d, e, m, parallax=st["parallax"], parallax_err=st["parallax_err"],
                            lnprior=lnprior, lngalprior=lng, lndustprior=lnd, av_gauss=avg,
                            wt_thresh=wt, data_coords=coords, Nmc_prior=20, Ndraws=40,
                            return_distreds=True, seed0=500, rstate_per_object="philox",
                            logl_dim_prior=True))
        assert idx.shape == (nstar, 40) and idx.min() >= 0
        for i, r in enumerate(rows):
            assert np.array_equal(idx[i], r[0]), i
            assert np.allclose(post[i], r[6].astype(np.float32), rtol=1e-6, atol=0), i
            assert np.allclose(dist_s[i], r[9].astype(np.float32), rtol=1e-6, atol=0), i
        if writer == "per_rank":
            assert len([x for x in os.listdir(box[0]) if x.startswith("sharded.r")]) == world
        print("sharded_smoke ok: %d ranks, %d objects, file == unsharded run (writer %s)" % (world, nstar, writer))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
