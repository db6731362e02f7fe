"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by stars: objects are independent (the reference's star loop,
fitting.py:1980, has no cross-star coupling except its RNG stream), so each
rank fits a contiguous range of the catalogue and there is NO collective on the
data path.  The only collective is the one-off broadcast of the model grid in
kernel layout (108 MB at 750k x 12) and of the static prior vector.
"""
import numpy as np

__all__ = ["shard_range", "shard_bounds", "broadcast_grid", "broadcast_array", "gather_rows",
           "fit_sharded"]


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of `n` objects for `rank` (keeps output row order;
    sizes differ by at most one)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_bounds(n, world, rank0_share=1.0):
    """[lo, hi) of every rank.  `rank0_share` < 1 gives rank 0 -- which also receives and
    writes everybody's rows -- that fraction of an equal share (0: rank 0 only writes);
    1.0 is `shard_range`."""
    n, world = int(n), int(world)
    if world == 1 or rank0_share >= 1.:
        return [shard_range(n, r, world) for r in range(world)]
    n0 = int(round(max(0., float(rank0_share)) * n / (world - 1 + max(0., float(rank0_share)))))
    rest = [shard_range(n - n0, r, world - 1) for r in range(world - 1)]
    return [(0, n0)] + [(n0 + a, n0 + b) for a, b in rest]


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
                seed0=0, rng="philox", chunk=256, rank0_share=1.0, queue_depth=4, writer="rank0",
                **fit_kwargs):
    """`BruteForce.fit` over all ranks of the default process group.

    Every rank fits the contiguous shard `shard_bounds(Ndata, world, rank0_share)[rank]` on
    its own GPU (no collective on the data path) and hands its finished rows to rank 0 in
    bounded pieces, WITHOUT the ranks ever waiting for each other's fits:

    * the fit (`BruteForce._fit`, this thread) packs every `chunk` finished objects into one
      block -- a structured array with the file's own dtypes (`h5io.ResultsFile.row_dtype`,
      18 KB per object at the defaults) -- and puts it into a bounded queue
      (`queue_depth` blocks: the only back-pressure is a writer slower than the fits);
    * a hand-off thread per rank drains that queue over a gloo SIDE GROUP
      (`dist.new_group(backend="gloo")`: host memory to host memory, nothing is pickled and
      nothing passes through RCCL): a round is one tiny `all_gather` of headers
      `(rows, first row, done, failed)` followed by point-to-point sends of exactly the
      rows that are ready -- a rank with nothing ready sends nothing and holds nobody up,
      so rank 0's shard, its unpacking and the writer's thread no longer sit in every other
      rank's critical path (round 3 gathered fixed blocks in lock-step from the fit loop);
    * rank 0's hand-off thread gives the blocks to the asynchronous `ResultsFile` writer at
      their catalogue positions -- row `lo_r + ...`, the mapping of reference
      fitting.py:1734-1748.  No rank holds more than `queue_depth * chunk` finished rows,
      rank 0 no more than a round of `world * chunk` on top, whatever the catalogue size.

    A failure anywhere -- a fit, the writer (its errors are sticky), a hand-off -- is
    announced in the next round's headers; every rank then stops its generator and raises,
    none is left waiting in a collective.  That includes the writer's LAST blocks: when every
    rank has reported "done", rank 0 flushes and closes the file inside the protocol and a
    closing round of headers carries the outcome.  The side group is destroyed on the way out.

    With `running_io=True` (default) the file on disk is current up to the last flush and
    `model_idx` is the last dataset of a block to be written, so a crash leaves the rows in
    flight unfitted (-99), never half-written.

    The bands that enter a fit are chosen once, from the whole catalogue's mask (a rank whose
    shard happens to lack a band must not run other kernel instantiations than its peers; up
    to three engines -- band sets -- stay resident per process).  Object `i` draws from its
    own stream keyed `seed0 + i`, so the file is
    identical for any number of ranks (the reference's single sequential
    stream, fitting.py:2039-2053, would make results depend on the sharding):
    `rng="philox"` (default) uses `rng.PhiloxRandomState(seed0 + i)`, which
    lets `lnpost` run on the GPU for the built-in priors; `rng="numpy"` uses
    `numpy.random.RandomState(seed0 + i)` (device `lnpost` as well; the host stage,
    optionally spread over `bf.host_workers` processes, for user prior hooks).

    `fit_kwargs` are `BruteForce.fit` keyword arguments (`lnprior_ext` arrays
    are sliced to each rank's rows).  `resume=True` completes an interrupted
    `running_io=True` file instead of creating one: rank 0 reads which rows still hold the
    sentinel, every rank fits the unfinished rows of its shard (same per-object streams, so
    the completed file equals an uninterrupted run's, for any number of ranks before and
    after).  Returns the number of objects this rank fitted; `fit_sharded.last_stats` holds this rank's
    timings (`fit_s`: until its last row was packed, `total_s`).

    `writer="per_rank"` takes the funnel out altogether: every rank writes the rows of ITS shard
    to `{save_file}.rNN.h5` (same datasets, `hi - lo` rows, its own asynchronous libhdf5 writer;
    nothing but the tiny headers crosses the side group), and when all are closed rank 0 writes
    `{save_file}.h5` as an INDEX: the 13 datasets as HDF5 virtual datasets that map row range
    `[lo_r, hi_r)` onto part `r` (`h5io.write_virtual_index`; the part files must stay beside it,
    `h5io.materialize` copies everything into one plain file when that is wanted).  Readers --
    h5py, `h5io.read_dataset` -- see the reference's layout.  With one writer 8 ranks x 25 k
    objects/s x 18 KB = 3.6 GB/s would pass through one gloo receiver and one libhdf5 thread;
    per rank it is 0.45 GB/s into a file of its own.  `resume=True` re-opens every rank's part
    (same number of ranks as the interrupted run).
    """
    import os
    import queue
    import threading
    import time
    import torch
    import torch.distributed as dist
    from . import h5io
    rank, world = dist.get_rank(), dist.get_world_size()
    kw = dict(fit_kwargs)
    resume = bool(kw.pop("resume", False))
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
    bounds = shard_bounds(Ndata, world, rank0_share)
    lo, hi = bounds[rank]
    fkw = {k: v for k, v in kw.items()
           if k not in ("phot_offsets", "apply_agewt", "apply_grad", "mag_max",
                        "merr_max", "parallax", "parallax_err", "data_coords",
                        "lnprior", "lngalprior", "lndustprior", "av_gauss",
                        "wt_thresh")}
    if "logl_dim_prior" not in fkw:
        fkw["logl_dim_prior"] = True
    par = kw.get("parallax")
    perr = kw.get("parallax_err")
    t_start = time.time()
    if writer not in ("rank0", "per_rank"):
        raise ValueError("fit_sharded: writer must be 'rank0' or 'per_rank'")
    per_rank = writer == "per_rank"
    side = dist.new_group(backend="gloo") if world > 1 else None     # host-to-host hand-off
    out = None
    todo = None
    # Who writes what: one file on rank 0 (rows arrive over the side group), or a part per rank.
    owner = rank == 0 or (per_rank and hi > lo)
    my_path = ("{0}.r{1:02d}.h5".format(save_file, rank) if per_rank else "{0}.h5".format(save_file))
    my_rows = (hi - lo) if per_rank else Ndata
    my_first = lo if per_rank else 0
    if per_rank and rank == 0 and hi == lo:
        owner = False
    open_error = None
    if per_rank and rank == 0 and not resume and os.path.exists("{0}.h5".format(save_file)):
        # the index is written LAST; that it cannot be created ("w-", reference fitting.py:1632)
        # is said now, not after the whole catalogue has been fitted
        open_error = OSError("fit_sharded: %s.h5 exists already" % save_file)
        owner = False
    if owner:
        try:
            if resume:
                # the interrupted file(s): rows whose `model_idx[:, 0]` still holds the sentinel
                # -99 were never fitted (reference fitting.py:1635)
                out = h5io.ResultsFile.resume(my_path, my_rows, Ndraws, save_dar_draws)
            else:
                lab = data_labels
                if per_rank and lab is not None:
                    lab = np.asarray(lab)[lo:hi]
                out = h5io.ResultsFile(my_path, my_rows, Ndraws, lab, save_dar_draws,
                                       running_io=running_io)
        except BaseException as e:
            open_error = e
    # One handshake for every way of opening: a rank that could not create / re-open its file
    # says so before anybody enters the hand-off protocol, and EVERY rank raises (without it
    # the others would wait for the failed rank in the first round of headers).
    ok = torch.tensor([0 if open_error is not None else 1], dtype=torch.int64)
    if world > 1:
        oks = [torch.empty(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(oks, ok, group=side)
        bad = [r for r in range(world) if not int(oks[r])]
    else:
        bad = [] if int(ok) else [0]
    if bad:
        if out is not None:
            try:
                out.close()
                if not resume:
                    os.remove(my_path)          # created a moment ago, holds nothing: a rerun must not trip over it
            except BaseException:
                pass
        if side is not None:
            dist.destroy_process_group(side)
        raise open_error or RuntimeError(
            "fit_sharded: rank(s) %s could not open their results file; this rank stops too"
            % ", ".join(str(r) for r in bad))
    if resume:
        # every rank fits only the unfinished rows of ITS shard, each contiguous run of them as
        # one `_fit` call seeded `seed0 + first row` -- object i draws from stream `seed0 + i` as
        # in the interrupted run, so the completed file equals an uninterrupted one.
        state = torch.zeros(Ndata, dtype=torch.uint8)
        if out is not None:
            state[my_first + torch.from_numpy(np.asarray(out.todo, dtype=np.int64))] = 1
        if world > 1 and not per_rank:
            dist.broadcast(state, src=0, group=side)
        todo = state.numpy().astype(bool)
    # the contiguous runs [a, b) of rows this rank fits
    if todo is None:
        runs = [(lo, hi)] if hi > lo else []
    else:
        mine = np.flatnonzero(todo[lo:hi]) + lo
        cuts = np.flatnonzero(np.diff(mine) != 1) + 1
        runs = [(int(r[0]), int(r[-1]) + 1) for r in np.split(mine, cuts) if r.size]
    nmine = sum(b - a for a, b in runs)
    current = [None]

    def fitted_rows():
        """(catalogue row, 13-tuple) of every row of this rank's runs, in row order."""
        for a, b in runs:
            rkw = dict(fkw)
            if lnprior_ext is not None:
                # per-object constraints: `_fit` sees objects a..b as 0..b-a
                rkw["lnprior_ext"] = {k: np.asarray(v)[a:b] for k, v in lnprior_ext.items()}
            g = bf._fit(
                data[a:b], data_err[a:b], data_mask[a:b],
                parallax=None if par is None else np.asarray(par)[a:b],
                parallax_err=None if perr is None else np.asarray(perr)[a:b],
                lnprior=lnprior, lngalprior=lngalprior, lndustprior=lndustprior,
                av_gauss=av_gauss, wt_thresh=wt_thresh, data_coords=data_coords[a:b],
                Ndraws=Ndraws, return_distreds=save_dar_draws,
                seed0=seed0 + a,
                rstate_per_object="philox" if rng == "philox" else None, **rkw)
            current[0] = g
            for k, res in enumerate(g):
                yield a + k, res
            current[0] = None

    gen = fitted_rows()
    rowdt, positions = h5io.ResultsFile.row_dtype(Ndraws, save_dar_draws)
    rowbytes = rowdt.itemsize
    q = queue.Queue(maxsize=max(1, int(queue_depth)))
    abort = threading.Event()
    failure = {"local": None, "remote": None}

    def hand_off():
        """Drains the queue: header round, then the rows that are ready (see above)."""
        done = False
        try:
            while True:
                item = None
                if not done and failure["local"] is None:
                    try:
                        item = q.get(timeout=0.02)
                    except queue.Empty:
                        pass
                n, start, block = 0, 0, None
                if item is not None:
                    if item[0] == "rows":
                        _, start, n, block = item
                    elif item[0] == "done":
                        done = True
                    else:
                        failure["local"] = failure["local"] or RuntimeError(item[1])
                hdr = torch.tensor([n, start, 1 if done else 0,
                                    0 if failure["local"] is None else 1], dtype=torch.int64)
                if world > 1:
                    hdrs = [torch.empty(4, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(hdrs, hdr, group=side)
                    hdrs = torch.stack(hdrs).numpy()
                else:
                    hdrs = hdr.numpy()[None, :]
                bad = np.flatnonzero(hdrs[:, 3])
                if bad.size:
                    if failure["local"] is None:
                        failure["remote"] = RuntimeError(
                            "fit_sharded: rank(s) %s failed; this rank stops too"
                            % ", ".join(str(int(b)) for b in bad))
                    return
                try:
                    if rank == 0:
                        got = []
                        for r in range(world):       # every payload is received before any is
                            nr = int(hdrs[r, 0])     # written: a failing write strands no sender
                            if nr == 0:
                                continue
                            if r == 0:
                                rows = block[:nr]
                            else:
                                buf = torch.empty(nr * rowbytes, dtype=torch.uint8)
                                dist.recv(buf, src=r, group=side)
                                rows = buf.numpy().view(rowdt)
                            got.append((int(hdrs[r, 1]), rows))
                        for first, rows in got:
                            out.write_block(first, {name: rows[name] for name, _ in positions})
                    elif n:
                        dist.send(torch.from_numpy(block[:n].view(np.uint8).reshape(-1)), dst=0,
                                  group=side)
                except BaseException as e:           # announced in the next round's headers
                    failure["local"] = e
                    continue
                if np.all(hdrs[:, 2] == 1) and not np.any(hdrs[:, 0]):
                    # Everything has been handed over -- but the writer is asynchronous: a
                    # failure in its last blocks (a full disk) only shows when the file is
                    # flushed and closed.  Rank 0 does that HERE, and one more round of headers
                    # tells everybody how it went; otherwise rank 0 alone would raise after the
                    # threads have ended and the others would wait in the closing barrier.
                    def closing_round():
                        """Everybody's verdict on the step just taken; True if all went well."""
                        if world == 1:
                            return failure["local"] is None
                        last = torch.tensor([0, 0, 1, 0 if failure["local"] is None else 1],
                                            dtype=torch.int64)
                        lasts = [torch.empty(4, dtype=torch.int64) for _ in range(world)]
                        dist.all_gather(lasts, last, group=side)
                        badr = [r for r in range(world) if int(lasts[r][3])]
                        if badr and failure["local"] is None:
                            failure["remote"] = RuntimeError(
                                "fit_sharded: rank %s could not finish the results file; "
                                "this rank stops too" % ", ".join(str(r) for r in badr))
                        return not badr
                    if out is not None:
                        try:
                            out.close()
                        except BaseException as e:
                            failure["local"] = e
                    all_closed = closing_round()
                    if per_rank and all_closed:
                        # every part is complete and closed: the index that presents them as
                        # the reference's one file
                        if rank == 0:
                            try:
                                h5io.write_virtual_index(
                                    "{0}.h5".format(save_file), Ndata, Ndraws, save_dar_draws,
                                    data_labels,
                                    [(a, b, os.path.basename("{0}.r{1:02d}.h5".format(save_file, r)))
                                     for r, (a, b) in enumerate(bounds) if b > a],
                                    overwrite=resume)
                            except BaseException as e:
                                failure["local"] = e
                        closing_round()
                    return
        except BaseException as e:                   # the collective itself failed
            failure["local"] = failure["local"] or e
        finally:
            if failure["local"] is not None or failure["remote"] is not None:
                abort.set()

    def put(item):
        while True:
            if abort.is_set():
                raise failure["remote"] or failure["local"] or RuntimeError("fit_sharded aborted")
            try:
                q.put(item, timeout=0.05)
                return
            except queue.Full:
                continue

    th = threading.Thread(target=hand_off, name="brutus-shard-handoff", daemon=True)
    th.start()
    fit_error = None
    t_fit = None
    try:
        bf._catalogue_mask = data_mask       # one band set for all ranks and runs (BruteForce._bands_for)
        try:
            exhausted = False
            pending = None                           # a row of the next run, already fitted
            while not exhausted:
                block = np.zeros(chunk, dtype=rowdt)
                n, start = 0, None
                with np.errstate(over="ignore"):
                    while n < chunk:
                        if pending is not None:
                            row, res = pending
                            pending = None
                        else:
                            try:
                                row, res = next(gen)
                            except StopIteration:
                                exhausted = True
                                break
                        if start is None:
                            start = row
                        elif row != start + n:       # a new run: blocks hold consecutive rows
                            pending = (row, res)
                            break
                        for name, pos in positions:
                            block[name][n] = res[pos]
                        n += 1
                if n and per_rank:
                    if abort.is_set():               # somebody else failed: stop fitting
                        raise failure["remote"] or failure["local"] or RuntimeError("fit_sharded aborted")
                    out.write_block(start - lo, {name: block[name][:n] for name, _ in positions})
                elif n:
                    put(("rows", start, n, block))
            t_fit = time.time() - t_start
            put(("done",))
        except BaseException as e:
            fit_error = e
            if not abort.is_set():                   # tell the others, whatever the queue holds
                failure["local"] = failure["local"] or e
        th.join()
    finally:
        bf._catalogue_mask = None
        if current[0] is not None and hasattr(current[0], "close"):
            current[0].close()              # shuts the scan-ahead helper thread down
        gen.close()
        close_error = None
        if out is not None:
            try:
                out.close()
            except BaseException as e:      # (sticky writer error: reported below)
                close_error = e
        if side is not None:
            dist.destroy_process_group(side)
    err = fit_error or failure["local"] or failure["remote"] or close_error
    if err is not None:
        raise err
    fit_sharded.last_stats = {"fit_s": t_fit, "total_s": time.time() - t_start,
                              "objects": nmine, "writer": writer}
    dist.barrier()
    return nmine
