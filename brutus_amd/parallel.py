"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by stars: objects are independent (the reference's star loop,
fitting.py:1980, has no cross-star coupling except its RNG stream), so each
rank fits a contiguous range of the catalogue and there is NO collective on the
data path.  The only collective is the one-off broadcast of the model grid in
kernel layout (108 MB at 750k x 12) and of the static prior vector.
"""
import numpy as np

__all__ = ["shard_range", "broadcast_grid", "broadcast_array", "gather_rows",
           "fit_sharded"]


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of `n` objects for `rank` (keeps output row order;
    sizes differ by at most one)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_grid(grid, nmodel, nfilt, device, src=0):
    """Broadcast the SoA grid tensor from `src`; returns a DeviceGrid on every
    rank.  `grid` is the source rank's DeviceGrid (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from . import _lib
    from .fitting import DeviceGrid
    nbytes = _lib.lib().brutus_grid_soa_bytes(nmodel, nfilt)
    if grid is not None:
        soa = grid.soa
    else:
        soa = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    dist.broadcast(soa, src=src)
    return DeviceGrid.from_soa(soa, nmodel, nfilt)


def broadcast_array(arr, src=0, device=None):
    """Broadcast a numpy array (shape/dtype known on `src` only)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        arr = np.ascontiguousarray(arr)
        meta = [(arr.shape, arr.dtype.str)]
    dist.broadcast_object_list(meta, src=src)
    shape, dt = meta[0]
    if rank != src:
        arr = np.empty(shape, dtype=np.dtype(dt))
    t = torch.from_numpy(arr.view(np.uint8).reshape(-1))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    if device is not None:
        arr = t.cpu().numpy().view(np.dtype(dt)).reshape(shape)
    return arr


def gather_rows(local_rows, dst=0):
    """Gather per-rank lists of result rows on `dst` in rank order (the row
    order of the sharded catalogue).  Host-side; results are small."""
    import torch.distributed as dist
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local_rows, out, dst=dst)
    if out is None:
        return None
    rows = []
    for part in out:
        rows.extend(part)
    return rows


def fit_sharded(bf, data, data_err, data_mask, data_labels, save_file,
                seed0=0, rng="philox", chunk=256, **fit_kwargs):
    """`BruteForce.fit` over all ranks of the default process group.

    Every rank fits the contiguous shard `shard_range(Ndata, rank, world)` on
    its own GPU (no collective on the data path).  The rows go to rank 0 in
    bounded pieces: in round k every rank packs its next `chunk` objects into ONE
    fixed-size byte block -- a structured array with the file's own dtypes
    (`h5io.ResultsFile.row_dtype`), 18 KB per object at the defaults -- and the blocks
    are gathered on rank 0 over a gloo SIDE GROUP (`dist.new_group(backend="gloo")`:
    host memory to host memory; nothing is pickled and nothing passes through RCCL,
    whose default group would stage pickled bytes in device buffers).  Rank 0 hands the
    blocks to the asynchronous `ResultsFile` writer at their catalogue positions -- row
    `lo_r + k * chunk + j`, the mapping of reference fitting.py:1734-1748 -- and nothing
    else is kept: no rank ever holds more than `chunk` finished rows, rank 0 no more
    than a few rounds of `world * chunk`, whatever the catalogue size.  With
    `running_io=True` (default) the file on disk is current up to the last flush, so a
    crash loses at most the rounds in flight.

    Object `i` draws from its own stream keyed `seed0 + i`, so the file is
    identical for any number of ranks (the reference's single sequential
    stream, fitting.py:2039-2053, would make results depend on the sharding):
    `rng="philox"` (default) uses `rng.PhiloxRandomState(seed0 + i)`, which
    lets `lnpost` run on the GPU for the built-in priors; `rng="numpy"` uses
    `numpy.random.RandomState(seed0 + i)` and the host stage (optionally
    spread over `bf.host_workers` processes).

    `fit_kwargs` are `BruteForce.fit` keyword arguments (`lnprior_ext` arrays
    are sliced to each rank's shard; `resume` is not supported here).  Returns
    the number of objects this rank fitted.
    """
    import torch.distributed as dist
    from . import h5io
    rank, world = dist.get_rank(), dist.get_world_size()
    kw = dict(fit_kwargs)
    if kw.pop("resume", False):
        raise ValueError("fit_sharded: `resume` is not supported (rows are written by "
                         "rank 0 in catalogue order; re-run the missing range instead)")
    Ndraws = kw.pop("Ndraws", 250)
    save_dar_draws = kw.pop("save_dar_draws", True)
    running_io = kw.pop("running_io", True)
    lnprior_ext = kw.pop("lnprior_ext", None)
    kw.pop("verbose", None)
    kw.pop("rstate", None)
    chunk = max(1, int(chunk))
    setup_keys = ("phot_offsets", "parallax", "parallax_err", "av_gauss",
                  "lnprior", "wt_thresh", "cdf_thresh", "apply_agewt",
                  "apply_grad", "lngalprior", "lndustprior", "dustfile",
                  "data_coords", "ltol_subthresh", "logl_initthresh", "mag_max",
                  "merr_max")
    skw = {k: kw[k] for k in setup_keys if k in kw}
    (data, data_err, data_mask, data_labels, data_coords, lnprior, lngalprior,
     lndustprior, av_gauss, wt_thresh, _) = bf._setup(
        data, data_err, data_mask, data_labels, **skw)
    Ndata = data.shape[0]
    lo, hi = shard_range(Ndata, rank, world)
    fkw = {k: v for k, v in kw.items()
           if k not in ("phot_offsets", "apply_agewt", "apply_grad", "mag_max",
                        "merr_max", "parallax", "parallax_err", "data_coords",
                        "lnprior", "lngalprior", "lndustprior", "av_gauss",
                        "wt_thresh")}
    if "logl_dim_prior" not in fkw:
        fkw["logl_dim_prior"] = True
    if lnprior_ext is not None:
        # per-object constraints: this rank sees objects lo..hi as 0..hi-lo
        fkw["lnprior_ext"] = {k: np.asarray(v)[lo:hi] for k, v in lnprior_ext.items()}
    par = kw.get("parallax")
    perr = kw.get("parallax_err")
    gen = bf._fit(
        data[lo:hi], data_err[lo:hi], data_mask[lo:hi],
        parallax=None if par is None else np.asarray(par)[lo:hi],
        parallax_err=None if perr is None else np.asarray(perr)[lo:hi],
        lnprior=lnprior, lngalprior=lngalprior, lndustprior=lndustprior,
        av_gauss=av_gauss, wt_thresh=wt_thresh, data_coords=data_coords[lo:hi],
        Ndraws=Ndraws, return_distreds=save_dar_draws,
        seed0=seed0 + lo,
        rstate_per_object="philox" if rng == "philox" else None, **fkw)
    bounds = [shard_range(Ndata, r, world) for r in range(world)]
    nround = (max(b - a for a, b in bounds) + chunk - 1) // chunk
    import torch
    out = None
    if rank == 0:
        out = h5io.ResultsFile("{0}.h5".format(save_file), Ndata, Ndraws,
                               data_labels, save_dar_draws,
                               running_io=running_io)
    rowdt, positions = h5io.ResultsFile.row_dtype(Ndraws, save_dar_draws)
    side = dist.new_group(backend="gloo") if world > 1 else None     # host-to-host hand-off
    nbytes = 8 + chunk * rowdt.itemsize
    try:
        for k in range(nround):
            # this rank's next piece (may be empty), packed: [count i64][chunk rows]
            block = np.zeros(nbytes, dtype=np.uint8)
            rows = block[8:].view(rowdt)
            n = 0
            with np.errstate(over="ignore"):
                for n in range(chunk + 1):
                    if n == chunk:
                        break
                    try:
                        res = next(gen)
                    except StopIteration:
                        break
                    for name, pos in positions:
                        rows[name][n] = res[pos]
            block[:8].view(np.int64)[0] = n
            if world == 1:
                parts = [block]
            else:
                mine = torch.from_numpy(block)
                got = ([torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                       if rank == 0 else None)
                dist.gather(mine, got, dst=0, group=side)
                parts = [g.numpy() for g in got] if rank == 0 else ()
            if rank == 0:
                for r, part in enumerate(parts):
                    cnt = int(part[:8].view(np.int64)[0])
                    if cnt:
                        prow = part[8:].view(rowdt)[:cnt]
                        out.write_block(bounds[r][0] + k * chunk,
                                        {name: prow[name] for name, _ in positions})
    finally:
        if hasattr(gen, "close"):
            gen.close()                     # shuts the scan-ahead helper thread down
        if out is not None:
            out.close()
    dist.barrier()
    return hi - lo
