"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by stars: objects are independent (the reference's star loop,
fitting.py:1980, has no cross-star coupling except its RNG stream), so each
rank fits a contiguous range of the catalogue and there is NO collective on the
data path.  The only collective is the one-off broadcast of the model grid in
kernel layout (108 MB at 750k x 12) and of the static prior vector.
"""
import numpy as np

__all__ = ["shard_range", "broadcast_grid", "broadcast_array", "gather_rows"]


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
