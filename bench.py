#!/usr/bin/env python
"""Benchmark of the per-star grid-likelihood hot path on MI355X.

Contract (one JSON line on stdout from rank 0):
    python bench.py --gpus N --steps K --warmup W
For N > 1 it is launched by `python -m torch.distributed.run --nproc-per-node N`
(one rank per GPU, RCCL over xGMI): rank 0 builds the synthetic grid, lays it
out on its GPU and broadcasts the SoA tensor once; every rank then fits its own
stars -- stars shard with no data-path collective (weak scaling).

A "step" = one pass of the hot path (brutus_fit_batch: star vectors resident in
HBM -> compact per-star survivor records in HBM) over one batch of `--batch`
synthetic stars against the 750k-model x 12-band grid.  Workload = BASELINE.json
configs[1] (Av-only: rvlim=(3.32, 3.32), no parallax); `--config 3` runs
configs[2] (Av+Rv free, parallax prior).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128,
                    help="stars per step per GPU (64 -> 128 -> 256: +4 %, +6 % stars/s; "
                         "workspace ~87 MB per star at 750k models)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="2 / 3: the grid-likelihood path (BASELINE configs[1] / [2]); "
                         "5: cluster.isochrone_loglike (configs[4], supplementary line)")
    ap.add_argument("--cluster-stars", type=int, default=5000)
    ap.add_argument("--nmodel", type=int, default=750000)
    ap.add_argument("--nfilt", type=int, default=12)
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="budget for the CPU baseline sample (0 = skip)")
    ap.add_argument("--streams", type=int, default=2,
                    help="host threads / HIP streams per GPU, each with its own "
                         "workspace, taking the steps round-robin (kernels of "
                         "consecutive batches overlap on the device)")
    ap.add_argument("--e2e-stars", type=int, default=8,
                    help="stars for the end-to-end BruteForce.fit() rate reported "
                         "beside the metric (0 = skip); rank 0, N=1 only")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def cpu_baseline(config, nmodel, nfilt, budget_s):
    """Time the CPU restatement (oracle/loglike_ref.c) on this host's cores for
    a bounded sample of the same workload (rank 0, N=1 only), in a fresh
    subprocess so that its worker pool never shares the GPU process.
    Baseline, not the target."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--config", str(config),
           "--nmodel", str(nmodel), "--nfilt", str(nfilt), "--seconds", str(budget_s)]
    try:
        out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             timeout=max(120., 10 * budget_s), check=True)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception as e:          # the GPU numbers stay valid without it
        return {"value": None, "unit": "stars/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline failed: %r" % (e,)}


def end_to_end(models, grid, stars, n, kw, with_par):
    """Second number asked for by SURVEY 8d: the whole `BruteForce.fit()` --
    device scan + host `lnpost` stage (user prior hook, numpy RandomState draws,
    resampling) + HDF5 output -- on `n` stars of the same workload.  The host
    stage dominates: it integrates the prior over every selected model with
    Nmc_prior=50 draws each, exactly like the reference."""
    import tempfile
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    _, labels, lmask = synth.make_mist_like_grid(models.shape[0], models.shape[1])
    bf = fitting.BruteForce(models, labels, lmask)
    bf.use_device_grid(grid)
    bf.batch_size = n
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        bf.fit(stars["flux"][:n], stars["err"][:n], stars["mask"][:n],
               np.arange(n), os.path.join(tmp, "e2e"),
               parallax=stars["parallax"][:n] if with_par else None,
               parallax_err=stars["parallax_err"][:n] if with_par else None,
               data_coords=stars["coords"][:n], lngalprior=gal_lnprior,
               # Av-only end to end = Rv pinned by its prior: rvlim=(3.32, 3.32)
               # would reject every Monte Carlo draw in lnpost
               rv_gauss=(3.32, 1e-6) if "rvlim" in kw else (3.32, 0.18),
               rstate=np.random.RandomState(862), verbose=False)
        dt = time.perf_counter() - t0
    res = {"value": n / dt, "unit": "stars/s", "stars": n,
           "note": "BruteForce.fit incl. host lnpost (Nmc_prior=50, Ndraws=250) and HDF5; "
                   "one sequential RandomState like the reference"}
    # same, with one RNG seed per object: the host stage then runs in a pool
    # of worker processes (order-independent results, what fit_sharded uses)
    from brutus_amd import h5io
    workers = int(max(2, min(32, (os.cpu_count() or 4) // 4)))
    n2 = min(len(stars["flux"]), 4 * workers)
    bf.host_workers = workers
    bf.batch_size = min(64, n2)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        out = h5io.ResultsFile(os.path.join(tmp, "e2e.h5"), n2, 250, np.arange(n2), True)
        gen = bf._fit(stars["flux"][:n2], stars["err"][:n2], stars["mask"][:n2],
                      parallax=stars["parallax"][:n2] if with_par else None,
                      parallax_err=stars["parallax_err"][:n2] if with_par else None,
                      data_coords=stars["coords"][:n2], lngalprior=gal_lnprior,
                      rv_gauss=(3.32, 1e-6) if "rvlim" in kw else (3.32, 0.18),
                      lnprior=bf._setup(stars["flux"][:n2], stars["err"][:n2],
                                        stars["mask"][:n2], None,
                                        data_coords=stars["coords"][:n2],
                                        lngalprior=gal_lnprior)[5],
                      Nmc_prior=50, Ndraws=250, seed0=862)
        for i, row in enumerate(gen):
            out.write_row(i, row)
        out.close()
        dt2 = time.perf_counter() - t0
    res["per_object_seeds_pool"] = {"value": n2 / dt2, "unit": "stars/s", "stars": n2,
                                    "host_workers": workers}
    # lnpost + resampling on the device (built-in priors, counter-based rstate)
    from brutus_amd.rng import PhiloxRandomState
    bf.host_workers = 0
    n3 = 4096          # enough objects to amortise file creation and the first batch
    big = synth.make_stars(models, n3, seed=4242, with_parallax=with_par)
    bf.batch_size = 128
    for rep in range(2):          # first pass warms the workspaces
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            bf.fit(big["flux"], big["err"], big["mask"],
                   np.arange(n3), os.path.join(tmp, "e2e"),
                   parallax=big["parallax"] if with_par else None,
                   parallax_err=big["parallax_err"] if with_par else None,
                   data_coords=big["coords"], lngalprior=gal_lnprior,
                   rv_gauss=(3.32, 1e-6) if "rvlim" in kw else (3.32, 0.18),
                   rstate=PhiloxRandomState(862), verbose=False)
            dt3 = time.perf_counter() - t0
    res["device_lnpost"] = {"value": n3 / dt3, "unit": "stars/s", "stars": n3,
                            "note": "same fit() with rstate=PhiloxRandomState: second cut, "
                                    "MC prior integral and resampling on the GPU"}
    return res


def measured_traffic(kernel, batch, config):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes kept under
    profiles/ (FETCH_SIZE and WRITE_SIZE in separate runs, corrected with the
    calibration stream as MI355X_MICROARCH.md prescribes; see
    profiles/README.md).  None when no measurement exists for this
    kernel / batch / config."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        for row in table["rows"]:
            if (row["kernel"] == kernel and row["batch"] == batch
                    and row["config"] == config):
                return row["hbm_bytes_per_launch"]
    except (IOError, ValueError, KeyError):
        pass
    return None


def _local_device(local_rank):
    """BRUTUS_BENCH_ONE_DEVICE=1: every rank on cuda:0 -- lets the N > 1 code
    path be exercised on a one-GPU box (with BRUTUS_BENCH_BACKEND=gloo; RCCL
    refuses two ranks on one device).  Never set for a measurement."""
    return 0 if os.environ.get("BRUTUS_BENCH_ONE_DEVICE") == "1" else local_rank


def _init_pg(dist, rank, world, dev):
    backend = os.environ.get("BRUTUS_BENCH_BACKEND", "nccl")      # nccl = RCCL on ROCm
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


def bench_cluster(args):
    """BASELINE configs[4]: one `isochrone_loglike` evaluation = 5 000 objects x
    12 bands against 15 mass-fraction slices x 2 000 EEP points (SURVEY 8d,
    config 5).  A step is one whole call (host unpacking + isochrone table +
    device block).  Does not shard: with --gpus N every rank evaluates its own
    replica (an MCMC over theta would run one chain per GPU)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from brutus_amd import _lib, cluster, synth
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = _local_device(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_pg(dist, rank, world, dev)
    L = _lib.lib()
    iso = synth.TableIsochrone(nbands=args.nfilt, neep=2000)
    nobj = args.cluster_stars
    phot, err, par, perr = synth.make_cluster(iso, nobj, seed=11 + rank)
    theta0 = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])

    def call(k):
        th = theta0 + np.array([1e-3, 1e-3, 1e-3, 0., 0.5, 0.]) * (k % 7)
        return cluster.isochrone_loglike(th, iso, phot, err, parallax=par,
                                         parallax_err=perr, device=dev)
    for k in range(args.warmup):
        call(k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        val = call(k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # device block alone (HIP events on the launch stream)
    L.brutus_enable_timing(1)
    call(0)
    n = C.c_int(0)
    names = (C.c_char_p * 8)()
    ms = (C.c_float * 8)()
    L.brutus_last_timing(C.byref(n), names, ms, 8)
    L.brutus_enable_timing(0)
    k_ms = dict((names[j].decode(), float(ms[j])) for j in range(n.value)).get("k_cluster")
    npts = 2000 + 14 * int(np.sum(iso.eep_grid <= 480.))     # evolved points only in slice 0
    pairs = float(nobj) * npts
    line = {
        "metric": "isochrone_loglike evaluations/s (5k stars x 12 bands x 15 SMF x 2000 EEP)",
        "value": world * args.steps / dt, "unit": "evaluations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "configs[4]: cluster.isochrone_loglike, %d objects, %d bands, "
                               "15 x 2000 isochrone table (%d live points)" % (nobj, args.nfilt, npts),
                   "parallelism": "replicas only, %d rank(s)" % world,
                   "timed_region": "whole isochrone_loglike call: host arrays in, lnl_tot out"},
        "star_points_per_s": world * pairs * args.steps / dt,
    }
    if k_ms:
        # the block re-reads only the 2.9 MB point table per 64-object workgroup:
        # it is bound by f64 VALU issue, the HBM figure is shown for completeness
        alg = 8. * (npts * (args.nfilt + 1) + nobj * (2 * args.nfilt + 3))
        flops = pairs * (3. * args.nfilt + 40.)
        line["roofline"] = {"bound": "hbm", "kernel": "k_cluster", "achieved": alg / (k_ms * 1e-3) / 1e9,
                            "peak": 8000.0, "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 8e12,
                            "traffic": None, "avg_launch_ms": k_ms,
                            "valu_f64_tflops": flops / (k_ms * 1e-3) / 1e12,
                            "valu_f64_peak_tflops": 78.6,
                            "device_star_points_per_s": pairs / (k_ms * 1e-3)}
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import brutus_oracle as O
        sub = max(16, min(nobj, int(100 * args.cpu_seconds / 10.)))
        t0 = time.perf_counter()
        O.isochrone_loglike(theta0, iso, phot[:sub], err[:sub], parallax=par[:sub],
                            parallax_err=perr[:sub])
        dc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": (sub / dc) / nobj, "unit": "evaluations/s",
                                "cores": 1, "kind": "port",
                                "sample": "oracle numpy restatement on %d of the %d objects "
                                          "(%.1f s), scaled to a full evaluation" % (sub, nobj, dc)}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.config == 5:
        return bench_cluster(args)
    import torch
    import torch.distributed as dist
    from brutus_amd import _lib, fitting, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    local_rank = _local_device(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_pg(dist, rank, world, dev)

    L = _lib.lib()
    nmodel, nfilt = args.nmodel, args.nfilt
    # ---- grid: built once on rank 0, broadcast in kernel layout -------------
    models = None
    if rank == 0:
        models, _, _ = synth.make_mist_like_grid(nmodel, nfilt)
        grid = fitting.DeviceGrid(models, device=dev)
    if world > 1:
        from brutus_amd import parallel
        grid = parallel.broadcast_grid(grid if rank == 0 else None, nmodel,
                                       nfilt, dev, src=0)
    # ---- stars: every rank draws its own shard -------------------------------
    if args.config == 2:
        kw = dict(rvlim=(3.32, 3.32))
        with_par = False
    else:
        kw = dict()
        with_par = True
    seed = {2: 1, 3: 2}[args.config]
    if models is None:
        # ranks > 0 need the f32 coefficients only to synthesise their stars
        models, _, _ = synth.make_mist_like_grid(nmodel, nfilt)
    B = args.batch
    nb_pool = max(1, min(4, args.steps))       # distinct batches cycled through
    stars = synth.make_stars(models, B * nb_pool, seed=seed + 1000 * rank,
                             with_parallax=with_par)
    params = fitting._make_params(
        (0., 20.), (0., 1e6), kw.get("rvlim", (1., 8.)), (3.32, 0.18),
        3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    NS = max(1, args.streams)
    engines = [fitting._Engine(grid, max_batch=B, mem_budget=64e9) for _ in range(NS)]
    eng = engines[0]
    batches = []
    for b in range(nb_pool):
        sl = slice(b * B, (b + 1) * B)
        batches.append(eng._upload(stars["flux"][sl], stars["err"][sl],
                                   stars["mask"][sl],
                                   stars["parallax"][sl] if with_par else None,
                                   stars["parallax_err"][sl] if with_par else None))
    # record buffer: the synthetic stars select up to ~500k models each
    cap = max(32 << 20, B * 600000)
    sel_bufs = [(torch.empty(cap, dtype=torch.int32, device=dev),
                 torch.empty((_lib.NVALS, cap), dtype=torch.float64, device=dev))
                for _ in range(NS)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]

    def step(i, j=0):
        f, e, m, p, pe, hp = batches[i % nb_pool]
        return engines[j].fit_batch_device(f, e, m, p, pe, hp, params,
                                           sel_buffers=sel_bufs[j])

    def run_steps(n):
        """n steps; with --streams > 1 the steps are dealt round-robin to NS
        host threads, each driving its own HIP stream and workspace."""
        if NS == 1:
            out = None
            for i in range(n):
                out = step(i)
            return out
        import threading
        outs = [None] * NS

        def worker(j):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                for i in range(j, n, NS):
                    outs[j] = step(i, j)
                streams[j].synchronize()

        th = [threading.Thread(target=worker, args=(j,)) for j in range(NS)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        return next(o for o in outs if o is not None)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(max(args.warmup, NS if args.warmup else 0))
    fence()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    nsel_total = int(out[2][-1].item())
    if nsel_total > cap:
        raise SystemExit("record buffer overflow (%d > %d): timing would be invalid"
                         % (nsel_total, cap))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel durations (HIP events on the launch stream), after the
    # timed region so that the events do not perturb `value` --------------------
    roofline = None
    ktimes = {}
    if rank == 0 and not args.no_kernel_timing:
        import ctypes as C
        L.brutus_enable_timing(1)
        reps = max(2, min(5, args.steps))
        for i in range(reps):
            step(i)
            n = C.c_int(0)
            names = (C.c_char_p * 16)()
            ms = (C.c_float * 16)()
            L.brutus_last_timing(C.byref(n), names, ms, 16)
            for j in range(n.value):
                ktimes.setdefault(names[j].decode(), []).append(float(ms[j]))
        L.brutus_enable_timing(0)
        avg = {k: float(np.mean(v)) for k, v in ktimes.items()}
        dom = max(avg, key=avg.get)
        bytes_per_star = nmodel * nfilt * 3 * 4      # one f32 grid read
        achieved = B * bytes_per_star / (avg[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": measured_traffic(dom, B, args.config),
                    "avg_launch_ms": avg[dom],
                    "all_kernels_ms": avg,
                    "algorithmic_bytes_per_launch": B * bytes_per_star}
        # measured on this box beside the 8 TB/s spec figure: a plain streaming
        # kernel (1 GiB read with 4 B/lane, 2 GiB written with 8 B/lane)
        n_cal = 256 << 20
        src = torch.empty(n_cal, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty(n_cal, dtype=torch.float64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.check(L.brutus_calibrate_traffic(src.data_ptr(), dst.data_ptr(), n_cal, None))
        e0.record()
        for _ in range(3):
            _lib.check(L.brutus_calibrate_traffic(src.data_ptr(), dst.data_ptr(), n_cal, None))
        e1.record()
        torch.cuda.synchronize()
        roofline["measured_stream_gbs"] = 3 * 12.0 * n_cal / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    stars_per_s = world * args.steps * B / dt
    line = {
        "metric": "stars/sec at 750k-model x 12-band grid; achieved HBM GB/s vs peak",
        "value": stars_per_s, "unit": "stars/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": ("configs[1]: 750k-model x 12-band grid, Av-only solve "
                                "(rvlim=(3.32,3.32)), no parallax" if args.config == 2
                                else "configs[2]: 750k-model x 12-band grid, Av+Rv free, "
                                     "parallax prior"),
                   "nmodel": nmodel, "nfilt": nfilt, "stars_per_step_per_gpu": B,
                   "timed_region": "brutus_fit_batch: device-resident star vectors -> "
                                   "device-resident compact survivor records",
                   "selected_models_last_batch": nsel_total,
                   "streams_per_gpu": NS,
                   "parallelism": "stars sharded, %d rank(s)" % world},
        "hbm_algorithmic_frac_whole_job":
            stars_per_s / world * nmodel * nfilt * 12 / 1e9 / HBM_PEAK_GBS,
    }
    if roofline is not None:
        line["roofline"] = roofline
    if world == 1 and args.e2e_stars > 0:
        line["fit_end_to_end"] = end_to_end(models, grid, stars, args.e2e_stars,
                                            kw, with_par)
    if world == 1 and args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(args.config, nmodel, nfilt, args.cpu_seconds)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
