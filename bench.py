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
(default 512) DISTINCT synthetic stars against the 750k-model x 12-band grid, in
sub-batches of `--sub-batch` stars.  Workload = BASELINE.json configs[1]
(Av-only: rvlim=(3.32, 3.32), no parallax) -> `value`; configs[2] (Av+Rv free,
parallax prior) is timed the same way in the same run -> `other_config`
(`--config 3` swaps them).  `roofline.frac` = stars/s x 108 MB / 8 TB/s for the
whole step; `roofline.kernels` lists every kernel with its own
algorithmic bytes.
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
# RCCL / device-tensor sharing across processes needs dmabuf IPC on this pool's driver
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=512,
                    help="DISTINCT stars per step per GPU: with the driver's 20 steps the "
                         "timed region covers 10 240 different stars (BASELINE configs: "
                         "'10k stars')")
    ap.add_argument("--sub-batch", type=int, default=128,
                    help="stars per brutus_fit_batch call (workspace ~95 MB per star at "
                         "750k models); a step is batch / sub_batch calls")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank fits its own steps x batch stars; strong: ONE "
                         "catalogue of steps x batch stars split by parallel.shard_range "
                         "(BASELINE configs[3]: --scaling strong --steps 250 --batch 4000)")
    ap.add_argument("--single-config", action="store_true",
                    help="skip the second configuration reported under other_config")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="2 / 3: the grid-likelihood path (BASELINE configs[1] / [2]); "
                         "4: configs[3], the sharded 1M-star catalogue = configs[2] with "
                         "--scaling strong --steps 250 --batch 4000 (the same line); "
                         "5: cluster.isochrone_loglike (configs[4], supplementary line)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed repeats of the K steps (SURVEY 8(d): >= 5 repeats of the whole "
                         "star set): every repeat times EXACTLY `--steps` steps between two "
                         "barrier + synchronize fences; `value` is the median, min / max and "
                         "every repeat's time ride along")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the parity block (a few timed stars re-checked against the C "
                         "restatement, 16 end-to-end objects against the oracle)")
    ap.add_argument("--cluster-stars", type=int, default=5000)
    ap.add_argument("--nmodel", type=int, default=750000)
    ap.add_argument("--nfilt", type=int, default=12)
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="budget for the CPU baseline sample (0 = skip)")
    ap.add_argument("--streams", type=int, default=3,
                    help="host threads / HIP streams per GPU, each with its own "
                         "workspace, taking the steps round-robin (kernels of "
                         "consecutive batches overlap on the device)")
    ap.add_argument("--e2e-stars", type=int, default=4096,
                    help="stars of the sequential-RandomState end-to-end BruteForce.fit() "
                         "leg reported beside the metric (0 = skip all end-to-end legs); "
                         "rank 0, N=1 only")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cluster", action="store_true",
                    help="skip the cluster-mode block (BASELINE configs[4]) of the default line")
    ap.add_argument("--no-sharp", action="store_true",
                    help="skip the sharp-posterior block (S/N 50 photometry + parallax at S/N 10 on "
                         "synth.make_sharp_grid: a few per cent of the grid selected per star)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL record (what bench_detail.json holds, ~25 KB) instead of the < 8 KB "
                         "summary line: for the A/B tooling under tools/ab/, never for the driver")
    ap.add_argument("--no-survey-grid", action="store_true",
                    help="skip the third block: the main configuration on SURVEY 8(d)'s own "
                         "grid generator (synth.make_grid, random model order)")
    return ap.parse_args()


def cpu_baseline(config, nmodel, nfilt, budget_s):
    """Time the CPU restatement (oracle/loglike_ref.c) on this host's cores for
    a bounded sample of the same workload (rank 0, N=1 only), in a fresh
    subprocess so that its worker pool never shares the GPU process.
    Baseline, not the target."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--config", str(config),
           "--nmodel", str(nmodel), "--nfilt", str(nfilt), "--seconds", str(budget_s)]

    def run(extra):
        out = subprocess.run(cmd + extra, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             timeout=max(120., 10 * budget_s), check=True)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    try:
        res = run([])            # all cores (up to 128 worker processes)
    except Exception as e:          # the GPU numbers stay valid without it
        return {"value": None, "unit": "stars/s", "cores": 0, "kind": "port",
                "sample": "cpu baseline failed: %r" % (e,)}
    try:
        # ... and ONE core: a few stars on an otherwise idle host
        res["single_core"] = run(["--single-stars", "4"])
    except Exception as e:
        res["single_core"] = {"value": None, "sample": "failed: %r" % (e,)}
    res.setdefault("host_cores", os.cpu_count())
    return res


def end_to_end(models, grid, stars, n, kw, with_par, check=True):
    """Second number asked for by SURVEY 8d: the whole `BruteForce.fit()` -- device scan +
    `lnpost` (second cut, Monte Carlo prior integral with Nmc_prior=50, resampling with
    Ndraws=250) + HDF5 output -- on stars of the same workload, default Galactic prior.

    value                 one sequential numpy RandomState for the whole catalogue, exactly
                          the reference's semantics; `lnpost` on the device with numpy's
                          own stream reproduced word for word (a single stream is
                          sequential by nature: one workgroup walks it)
    numpy_per_object      RandomState(seed0 + i) per object (sharding-independent):
                          the streams of a batch are walked in parallel
    device_lnpost         rstate=PhiloxRandomState (counter-based, random access)
    host_lnpost           the same fit with the host stage (numpy draws), for reference
    """
    import tempfile
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    _, labels, lmask = synth.make_mist_like_grid(models.shape[0], models.shape[1])
    bf = fitting.BruteForce(models, labels, lmask)
    bf.use_device_grid(grid)
    rvg = (3.32, 1e-6) if "rvlim" in kw else (3.32, 0.18)
    # Av-only end to end = Rv pinned by its prior: rvlim=(3.32, 3.32) would reject every
    # Monte Carlo draw in lnpost

    kept = {}

    def run(st, nfit, batch, **fkw):
        """stars/s of `reps` whole fit() calls: the MEDIAN (a best-of-N flatters)."""
        bf.batch_size = batch
        keep = fkw.pop("keep", None)
        mk_rstate = fkw.pop("rstate")
        times = []
        for rep in range(fkw.pop("reps", 1)):
            with tempfile.TemporaryDirectory() as tmp:
                t0 = time.perf_counter()
                bf.fit(st["flux"][:nfit], st["err"][:nfit], st["mask"][:nfit], np.arange(nfit),
                       os.path.join(tmp, "e2e"),
                       parallax=st["parallax"][:nfit] if with_par else None,
                       parallax_err=st["parallax_err"][:nfit] if with_par else None,
                       data_coords=st["coords"][:nfit], lngalprior=gal_lnprior,
                       rv_gauss=rvg, verbose=False, rstate=mk_rstate(), **fkw)
                times.append(time.perf_counter() - t0)
                if keep is not None and keep not in kept:
                    from brutus_amd import h5io
                    kept[keep] = {k: h5io.read_dataset(os.path.join(tmp, "e2e.h5"), k)[:16]
                                  for k in ("model_idx", "obj_log_evid", "obj_log_post")}
        return nfit / float(np.median(times))

    n_shared = max(n, 1)
    big = synth.make_stars(models, 4096, seed=4242, with_parallax=with_par)
    res = {"value": run(big, n_shared, 128, rstate=lambda: np.random.RandomState(862), reps=3),
           "unit": "stars/s", "stars": n_shared, "statistic": "median of 3 whole fit() calls",
           "note": "BruteForce.fit, one sequential numpy RandomState like the reference "
                   "(Nmc_prior=50, Ndraws=250, HDF5); lnpost on the device, numpy's stream "
                   "reproduced word for word"}
    n2 = min(4096, max(n, 1))
    # per-object numpy seeds go through _fit (fit() itself takes one rstate)
    from brutus_amd import h5io
    bf.batch_size = 128
    lnprior = bf._setup(big["flux"][:n2], big["err"][:n2], big["mask"][:n2], None,
                        data_coords=big["coords"][:n2], lngalprior=gal_lnprior)[5]
    times = []
    for rep in range(3):
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            out = h5io.ResultsFile(os.path.join(tmp, "e2e.h5"), n2, 250, np.arange(n2), True)
            gen = bf._fit(big["flux"][:n2], big["err"][:n2], big["mask"][:n2],
                          parallax=big["parallax"][:n2] if with_par else None,
                          parallax_err=big["parallax_err"][:n2] if with_par else None,
                          data_coords=big["coords"][:n2], lngalprior=gal_lnprior,
                          rv_gauss=rvg, lnprior=lnprior, Nmc_prior=50, Ndraws=250, seed0=862)
            for i, row in enumerate(gen):
                out.write_row(i, row)
            out.close()
            times.append(time.perf_counter() - t0)
    res["numpy_per_object"] = {"value": n2 / float(np.median(times)), "unit": "stars/s", "stars": n2,
                               "note": "RandomState(seed0 + i) per object, device lnpost"}
    res["device_lnpost"] = {"value": run(big, 4096, 128, rstate=lambda: PhiloxRandomState(862),
                                         reps=3, keep="philox"),
                            "unit": "stars/s", "stars": 4096,
                            "note": "rstate=PhiloxRandomState: counter-based stream"}
    bf.device_numpy_rng = False
    nh = 6
    res["host_lnpost"] = {"value": run(big, nh, nh, rstate=lambda: np.random.RandomState(862)),
                          "unit": "stars/s", "stars": nh,
                          "note": "same fit with the host stage (numpy draws the normals), "
                                  "what round 1 reported as fit_end_to_end.value"}
    bf.device_numpy_rng = True
    if check and "philox" in kept:
        res["parity"] = parity_end_to_end(models, labels, lmask, big, kept["philox"], rvg, with_par)
    return res


SHARP_STARS = dict(frac_err=0.02, parallax_snr=10., frac_no_parallax=0.)


def end_to_end_sharp(models, grid, n):
    """`BruteForce.fit()` to the HDF5 file on the sharp-posterior workload (counter-based
    stream, device `lnpost`): what a user of the demo notebooks' regime sees end to end."""
    import tempfile
    from brutus_amd import fitting, synth
    from brutus_amd.galprior import gal_lnprior
    from brutus_amd.rng import PhiloxRandomState
    _, labels, lmask = synth.make_sharp_grid(models.shape[0], models.shape[1])
    bf = fitting.BruteForce(models, labels, lmask)
    bf.use_device_grid(grid)
    bf.batch_size = 128
    st = synth.make_stars(models, n, seed=4243, with_parallax=True, **SHARP_STARS)
    times = []
    for rep in range(3):
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            bf.fit(st["flux"], st["err"], st["mask"], np.arange(n), os.path.join(tmp, "e2e"),
                   parallax=st["parallax"], parallax_err=st["parallax_err"],
                   data_coords=st["coords"], lngalprior=gal_lnprior, verbose=False,
                   rstate=PhiloxRandomState(862))
            times.append(time.perf_counter() - t0)
    return {"value": n / float(np.median(times)), "unit": "stars/s", "stars": n,
            "statistic": "median of 3 whole fit() calls",
            "note": "BruteForce.fit, rstate=PhiloxRandomState, Nmc_prior=50, Ndraws=250, HDF5"}


def csrc_sha16():
    """Fingerprint of the kernel sources (brutus_amd/csrc + include): the PMC table under
    profiles/ carries the fingerprint it was measured on, so a kernel edited after the PMC
    pass shows up as `roofline.traffic_stale` (the GPU box has no .git to diff against)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "brutus_amd", "csrc", "*"))
                     + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        if os.path.isfile(fn):
            h.update(os.path.basename(fn).encode())
            h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def traffic_commit():
    """(commit, source fingerprint) the PMC table under profiles/ was measured on."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
            return t.get("commit"), t.get("csrc_sha16")
    except (IOError, ValueError):
        return None, None


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
        # (rows are per kernel; a timer name of the library can cover several kernels)
        tot = [row["hbm_bytes_per_launch"] for row in table["rows"]
               if row["batch"] == batch and row["config"] == config
               and (row["kernel"] == kernel
                    or (kernel != "__total__" and TIMER_OF(row["kernel"]) == kernel))]
        if tot:
            return float(sum(tot))
    except (IOError, ValueError, KeyError):
        pass
    return None


def measure_issue(L, torch, dev):
    """Time one SIMD needs per vector wave-instruction on THIS box, at the clock the device
    holds under that load (brutus_calibrate_issue: every SIMD busy with one kind of
    instruction and nothing else): plain float32, float64, float32 transcendental.  The
    spec-sheet figures (2 / 4 cycles at 2.4 GHz = 0.83 / 1.67 ns) ride along."""
    from brutus_amd import _lib
    out = {}
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    for name, kind, waves in (("f32", 0, 4), ("f64", 1, 2), ("trans32", 2, 4)):
        scratch = torch.empty(ncu * waves * 256, dtype=torch.float32, device=dev)
        iters = 2000
        _lib.check(L.brutus_calibrate_issue(kind, 50, waves, scratch.data_ptr(), scratch.numel(), None))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.brutus_calibrate_issue(kind, iters, waves, scratch.data_ptr(), scratch.numel(), None))
        e1.record()
        torch.cuda.synchronize()
        out[name + "_ns"] = e0.elapsed_time(e1) * 1e6 / (iters * 128. * waves)
    out["spec_ns"] = {"f32": 2 / 2.4, "f64": 4 / 2.4}
    out["simds"] = 4 * ncu
    out["how"] = ("brutus_calibrate_issue: 4 (f32, transcendental) / 2 (f64) waves per SIMD of "
                  "back-to-back instructions of one kind, HIP events")
    return out


def valu_table(batch, config):
    """Rows of profiles/sq_valu.json (tools/sq_to_json.py: SQ_INSTS_VALU per kernel and call,
    static float64 / transcendental shares) for this batch / config, and whether the kernel
    sources changed since."""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_valu.json")) as f:
            t = json.load(f)
        rows = [r for r in t["rows"] if r["batch"] == batch and r["config"] == config]
        return rows, t.get("csrc_sha16") != csrc_sha16(), t.get("commit")
    except (IOError, ValueError, KeyError):
        return [], None, None


def valu_block(rows, stale, commit, issue, call_ms):
    """`roofline.valu`: the ceiling that binds.  Issue time = sum over the call's kernels of
    (vector wave-instructions x the issue time of their kind) / number of SIMDs -- what the
    call would take if every SIMD issued vector instructions back to back, nothing else in
    the way -- against the wall time of a call."""
    if not rows or not issue:
        return None
    per_kernel, tot_n, tot_t = {}, 0., 0.
    for r in rows:
        if r["kernel"] == "__total__":
            continue
        n = r["valu_wave_insts_per_call"]
        f64, tr = r.get("f64_share", 0.), r.get("trans32_share", 0.)
        ns = n * (f64 * issue["f64_ns"] + tr * issue["trans32_ns"] + (1. - f64 - tr) * issue["f32_ns"])
        ms = ns / issue["simds"] * 1e-6
        per_kernel[r["kernel"]] = {"valu_wave_insts_per_call": n, "f64_share": f64,
                                   "trans32_share": tr, "issue_ms_per_call": ms}
        tot_n += n
        tot_t += ms
    return {"bound": "vector instruction issue", "unit": "ms of issue time per call / ms per call",
            "valu_wave_insts_per_call": tot_n, "issue_ms_per_call": tot_t, "call_ms": call_ms,
            "frac": tot_t / call_ms if call_ms else None, "issue_ns_per_wave_inst": issue,
            "kernels": per_kernel, "stale": stale, "from_commit": commit,
            "source": "profiles/sq_valu.json (rocprofv3 --pmc SQ_INSTS_VALU of tools/pmc_workload.py; "
                      "static ISA mix per kernel), issue times measured in this run"}


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


def bench_cluster(args, emit=True):
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
    if world > 1 and emit:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_pg(dist, rank, world, dev)
    L = _lib.lib()
    iso = synth.TableIsochrone(nbands=args.nfilt, neep=2000)
    nobj = args.cluster_stars
    phot, err, par, perr = synth.make_cluster(iso, nobj, seed=11 + rank)
    theta0 = np.array([-0.1, 9.6, 0.2, 3.3, 850., 0.05])

    walk = np.random.RandomState(5).normal(size=(args.steps + args.warmup + 64, 6))

    def call(k):
        # a sampler's walk: every evaluation has a theta of its own (no point table is
        # ever asked for twice; the catalogue-only terms are what the cache keeps)
        th = theta0 + np.array([1e-3, 1e-3, 1e-3, 0., 0.5, 1e-3]) * walk[k]
        return cluster.isochrone_loglike(th, iso, phot, err, parallax=par,
                                         parallax_err=perr, device=dev)
    for k in range(args.warmup):
        call(args.steps + k)

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
    # the same with the plug-in's time taken out, and with a point table that is revisited
    t_plug = 0.
    for k in range(min(args.steps, 16)):
        th = theta0 + np.array([1e-3, 1e-3, 1e-3, 0., 0.5, 1e-3]) * walk[k]
        t1 = time.perf_counter()
        iso.get_seds_grid(smf_grid=iso.smf_grid, feh=th[0], loga=th[1], av=th[2], rv=th[3],
                          eep=np.linspace(202., 808., 2000), dist=th[4])
        t_plug += time.perf_counter() - t1
    t_plug /= min(args.steps, 16)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(args.steps):
        call(0)
    torch.cuda.synchronize()
    dt_hit = (time.perf_counter() - t1) / args.steps
    # device block alone (HIP events on the launch stream)
    L.brutus_enable_timing(1)
    call(1)
    n = C.c_int(0)
    names = (C.c_char_p * 8)()
    ms = (C.c_float * 8)()
    L.brutus_last_timing(C.byref(n), names, ms, 8)
    L.brutus_enable_timing(0)
    k_ms = dict((names[j].decode(), float(ms[j])) for j in range(n.value)).get("k_cluster")
    npts = 2000 + 14 * int(np.sum(iso.eep_grid <= 480.))     # evolved points only in slice 0
    npts_run = 15 * 2000                                     # the kernel walks the dropped ones too
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
        "plugin_ms_per_step": 1e3 * t_plug,
        "library_ms_per_step": 1e3 * (dt / args.steps - t_plug),
        "revisited_table_evaluations_per_s": 1. / dt_hit,
    }
    if k_ms:
        # the block re-reads only the 2.9 MB point table per 64-object workgroup:
        # it is bound by f64 VALU issue, the HBM figure is shown for completeness
        alg = 8. * (npts * (args.nfilt + 1) + nobj * (2 * args.nfilt + 3))
        flops = pairs * (3. * args.nfilt + 40.)
        tf = flops / (k_ms * 1e-3) / 1e12
        line["roofline"] = {"bound": "vector f64 (the block re-reads a 2.9 MB point table per 64-object "
                                     "workgroup: HBM sees next to nothing)",
                            "kernel": "k_cluster", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s",
                            "frac": tf / 78.6, "traffic": None, "avg_launch_ms": k_ms,
                            "hbm_algorithmic_gbs": alg / (k_ms * 1e-3) / 1e9,
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
    if rank == 0 and emit:
        emit_line(line)
    if world > 1 and emit:
        dist.destroy_process_group()
    return line


def parity_scan(recs, stars, picks, models, kw, with_par):
    """CHECKER (never timed, never part of the product path): `picks` stars of the timed set,
    their device records against oracle/loglike_ref.c + the first cut of `lnpost`
    (fitting.py:976-991) -- SURVEY 8(d) "parity gates reported with the metric"."""
    try:
        from oracle import c_oracle
        from brutus_amd.pdf import scale_parallax_lnprior
        if not c_oracle.available():
            return {"error": "oracle/libbrutus_ref.so not built"}
        rec, off, ndim, k1, k2 = recs
        sel_equal = k_equal = True
        max_rel = max_av = 0.
        for s in picks:
            par = float(stars["parallax"][s]) if with_par else np.nan
            perr = float(stars["parallax_err"][s]) if with_par else np.nan
            tr = {}
            lnl, nd, chi2, sc, av, rv, icov = c_oracle.loglike(
                stars["flux"][s], stars["err"][s], stars["mask"][s], models, parallax=par,
                parallax_err=perr, trace=tr, **kw)
            with np.errstate(all="ignore"):
                lnprob = lnl + scale_parallax_lnprior(sc, 1. / np.sqrt(np.abs(icov[:, 0, 0])), par, perr)
            lnprob = np.where(np.isfinite(lnprob), lnprob, -1e300)
            sel = np.where(lnprob > np.log(1e-3) + lnprob.max())[0]
            idx, vals = rec.host(int(off[s]), int(off[s + 1]))
            k_equal = k_equal and int(k1[s]) == tr["K1"] and int(k2[s]) == tr["K2"]
            same = np.array_equal(idx, sel)
            sel_equal = sel_equal and same
            if same:
                with np.errstate(all="ignore"):
                    for got, ref in ((vals[0], lnl[sel]), (vals[1], chi2[sel]), (vals[2], sc[sel])):
                        max_rel = max(max_rel, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300))))
                    max_av = max(max_av, float(np.max(np.abs(vals[3] - av[sel]))))
        return {"stars": len(picks), "sel_equal": bool(sel_equal), "k1_k2_equal": bool(k_equal),
                "max_rel": max_rel, "max_abs_av": max_av,
                "against": "oracle/loglike_ref.c + first cut, same inputs; max_rel over lnlike, chi2, scale"}
    except Exception as e:                  # the parity block never costs the measurement
        return {"error": repr(e)}


def parity_end_to_end(models, labels, lmask, st, got, rvg, with_par, n=16):
    """CHECKER: the first `n` rows fit() wrote in the counter-based leg against C `loglike` +
    the oracle's numpy `lnpost` / resampling driven by the same PhiloxRandomState(862)."""
    try:
        from oracle import brutus_oracle as O
        from oracle import c_oracle
        from brutus_amd.galprior import gal_lnprior
        from brutus_amd.rng import PhiloxRandomState
        lnprior = O.static_lnprior(labels, lmask)
        ro = PhiloxRandomState(862)
        idx_equal, max_rel = True, 0.
        for i in range(n):
            par = float(st["parallax"][i]) if with_par else np.nan
            perr = float(st["parallax_err"][i]) if with_par else np.nan
            # fit()'s band cuts (reference fitting.py:1405-1410: mag_max=50, merr_max=0.25)
            with np.errstate(all="ignore"):
                mag, merr = O.magnitude(st["flux"][i][None, :], st["err"][i][None, :])
            mask = st["mask"][i] & ~((mag[0] > 50.) | (merr[0] > 0.25))
            results = c_oracle.loglike(st["flux"][i], st["err"][i], mask, models,
                                       parallax=par, parallax_err=perr, rv_gauss=rvg)
            ref = O.finish_star(results, lnprior, labels, st["coords"][i], par, perr, ro,
                                gal_lnprior, Nmc_prior=50, Ndraws=250)
            idx_equal = idx_equal and np.array_equal(np.asarray(ref[0]), got["model_idx"][i])
            ev = float(got["obj_log_evid"][i])
            max_rel = max(max_rel, abs(ev - ref[7]) / max(abs(ref[7]), 1e-300))
        return {"objects": n, "model_idx_equal": bool(idx_equal), "max_rel_log_evid": max_rel,
                "note": "file values are float32 (the reference's layout): 6e-8 is rounding"}
    except Exception as e:
        return {"error": repr(e)}


WORKLOADS = {
    2: "configs[1]: 750k-model x 12-band grid, Av-only solve (rvlim=(3.32,3.32)), no parallax",
    3: "configs[2]: 750k-model x 12-band grid, Av+Rv free, parallax prior",
}
# the two synthetic grid generators (brutus_amd/synth.py).  BASELINE.json's configs name a
# "MIST grid": lattice-ordered in (mini, eep, feh) like the real grid files -> `mist_like`,
# the headline; SURVEY 8(d) spells out a generator with models in RANDOM order -> `survey8d`,
# reported as a third block (no index locality: the worst case for the list kernels).
GRIDS = {
    "mist_like": "synth.make_mist_like_grid (lattice-ordered like the MIST grid files)",
    "survey8d": "synth.make_grid (SURVEY 8(d) generator: random model order)",
}

# algorithmic bytes of one kernel launch: what the kernel has to
# move at the very least for the work it is given.  g = grid bytes per star (108 MB at
# 750k x 12), pairs = stars x models, c = (nsel, ncand, nder) of the batch: selected
# records, candidates of the cull, selected models whose values are derived.
KERNEL_ALG_BYTES = {
    "k_pre32": lambda B, g, pairs, c: B * g,                     # the SURVEY 8(d) unit: one grid read per star
    "k_top": lambda B, g, pairs, c: 8. * pairs / 2048.,          # block partials only
    "k_surv_compact": lambda B, g, pairs, c: 4. * pairs,         # the float32 statistic once
    "k_fflux": lambda B, g, pairs, c: 104. * c[1],               # candidates: index in, 11 values + step + lnprob out
    "k_sel_classify": lambda B, g, pairs, c: 4. * pairs,         # one float32 plane
    "k_select": lambda B, g, pairs, c: 8. * c[0] + 4. * c[2],    # (model, slot) per record + derived list
    "k_derive": lambda B, g, pairs, c: 92. * c[2],               # derived records: index in, 11 values out
}


def TIMER_OF(kernel):
    """The library's timer name (`kernels_ms`) a kernel runs under (brutus_kernels.hip, run_fit)."""
    k = kernel.split("<")[0]
    if k == "k_fflux":                              # k_fflux<NB, RVF, FIRST>: the continuation launches
        # (the traffic table names the family without its template arguments: its bytes are the first launch's)
        return "k_fflux" if "<" not in kernel or kernel.rstrip(">").endswith("true") else "k_fflux_cont"
    if k in ("k_pre32", "k_pre32s"):
        return "k_pre32"
    if k in ("k_top", "k_top1", "k_hot_list"):
        return "k_top"
    if k in ("k_cmp_count32", "k_cmp_scatter"):     # (k_offsets / k_items run under two names: left out)
        return "k_surv_compact"
    if k == "k_rec_index":
        return "k_select"
    return k


def run_config(config, args, L, grid, models, dev, world, rank, dist, torch, star_kw=None):
    """Time `args.steps` steps of configs[config - 1] on this rank; returns a dict with
    the whole-job rate and the per-kernel durations.  A step = `args.batch` DISTINCT
    stars, pushed through brutus_fit_batch in sub-batches of `args.sub_batch`."""
    from brutus_amd import _lib, fitting, synth
    nmodel, nfilt = args.nmodel, args.nfilt
    kw = dict(rvlim=(3.32, 3.32)) if config == 2 else dict()
    with_par = config == 3
    seed = {2: 1, 3: 2}[config]
    B, SB = args.batch, args.sub_batch
    nsub = (B + SB - 1) // SB
    strong = args.scaling == "strong"
    # weak: every rank fits its own `steps x batch` stars; strong: ONE catalogue of
    # `steps x batch` stars, rank r takes the contiguous shard parallel.shard_range gives it
    nstars_job = args.steps * B
    if strong:
        from brutus_amd import parallel
        lo, hi = parallel.shard_range(nstars_job, rank, world)
        stars = synth.make_stars(models, nstars_job, seed=seed, with_parallax=with_par, **(star_kw or {}))
        stars = {k: v[lo:hi] for k, v in stars.items()}
        mine = hi - lo
    else:
        stars = synth.make_stars(models, nstars_job, seed=seed + 1000 * rank,
                                 with_parallax=with_par, **(star_kw or {}))
        mine = nstars_job
    params = fitting._make_params(
        (0., 20.), (0., 1e6), kw.get("rvlim", (1., 8.)), (3.32, 0.18),
        3e-2, 1e-2, 5e-3, True, wt_thresh=1e-3)
    NS = max(1, args.streams)
    engines = [fitting._Engine(grid, max_batch=SB, mem_budget=64e9) for _ in range(NS)]
    eng = engines[0]
    # every sub-batch of the job resident in HBM before the clock starts
    subs = []
    for a in range(0, mine, SB):
        sl = slice(a, min(mine, a + SB))
        subs.append(eng._upload(stars["flux"][sl], stars["err"][sl], stars["mask"][sl],
                                stars["parallax"][sl] if with_par else None,
                                stars["parallax_err"][sl] if with_par else None))
    # record buffers sized once, before the clock starts (grow=False: a batch that does not
    # fit fails the run instead of silently repeating work inside the timed region)
    # record buffers sized before the clock starts: generously here, by the warm-up steps
    # where that is not enough; a growth inside the timed region (= a batch done twice)
    # fails the run instead of passing as a slow step
    cap = max(32 << 20, int(SB * nmodel * 0.62))
    for en in engines:
        en._rec_bufs = en._record_buffers(cap)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]

    def one(i, j=0):
        f, e, m, p, pe, hp = subs[i % len(subs)]
        return engines[j].fit_batch_device(f, e, m, p, pe, hp, params)

    def run(n_sub, first=0):
        """n_sub consecutive sub-batches, dealt round-robin to NS host threads, each
        with its own HIP stream and workspace."""
        if NS == 1:
            out = None
            for i in range(first, first + n_sub):
                out = one(i)
            return out
        import threading
        outs = [None] * NS

        def worker(j):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                for i in range(first + j, first + n_sub, NS):
                    outs[j] = one(i, j)
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

    run(max(args.warmup * nsub, NS if args.warmup else 0))
    if any(en.regrown for en in engines):       # buffers grew: once more, now at full size
        run(NS)
    grown = [en.regrown for en in engines]
    # SURVEY 8(d) timing protocol: >= 5 timed repeats of the whole star set; every repeat is
    # EXACTLY `steps` steps (this rank's share of them) between two barrier + synchronize
    # fences, its time the max over ranks; the line reports the MEDIAN repeat, min / max beside
    R = max(1, args.repeats)
    own, job = [], []
    for rep in range(R):
        fence()
        t0 = time.perf_counter()
        out = run(len(subs))
        fence()
        dt = time.perf_counter() - t0
        own.append(dt)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        job.append(dt)
    if [en.regrown for en in engines] != grown:
        raise SystemExit("record buffers grew inside the timed region: timing invalid")
    nsel_total = int(out[0].counts[0])
    med = int(np.argsort(job)[len(job) // 2])          # the median repeat
    dt = job[med]
    per_rank = [own[med]]
    if world > 1:                                      # every rank's OWN time of that repeat
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = own[med]
        dist.all_reduce(t)
        per_rank = [float(x) for x in t.cpu()]
    total_stars = nstars_job if strong else world * nstars_job
    res = {"value": total_stars / dt, "ms_per_step": dt / args.steps * 1e3,
           "repeats": R, "repeat_s": job, "value_min": total_stars / max(job),
           "value_max": total_stars / min(job), "per_rank_s": per_rank,
           "timed_region_s": float(sum(job)),
           "stars_timed_per_rank": mine, "selected_models_last_sub_batch": nsel_total,
           "selected_fraction": nsel_total / float(int(subs[(len(subs) - 1)][0].shape[0]) * nmodel)}
    if rank == 0 and not args.no_parity:
        # parity gate beside the metric: four stars of the timed set, re-fitted now
        rec, ndim_t, k1, k2 = one(0)
        torch.cuda.synchronize()
        n0 = int(subs[0][0].shape[0])
        picks = sorted(set([0, n0 // 3, (2 * n0) // 3, n0 - 1]))
        res["parity"] = parity_scan((rec, rec.off.cpu().numpy(), ndim_t, k1, k2), stars, picks,
                                    models, kw, with_par)

    # ---- per-kernel durations (HIP events on the launch stream), after the timed
    # region and strictly sequential: with two streams the kernels of two sub-batches
    # overlap and every individual duration is stretched
    if rank == 0 and not args.no_kernel_timing:
        import ctypes as C
        ktimes = {}
        L.brutus_enable_timing(1)
        reps = max(2, min(4, len(subs)))
        nsel_k = []
        for i in range(reps):
            o = one(i)
            torch.cuda.synchronize()
            c = o[0].counts
            nsel_k.append((int(c[0]), int(c[1]), int(c[2] - c[1])))
            n = C.c_int(0)
            names = (C.c_char_p * 24)()
            ms = (C.c_float * 24)()
            L.brutus_last_timing(C.byref(n), names, ms, 24)
            for j in range(n.value):
                ktimes.setdefault(names[j].decode(), []).append(float(ms[j]))
        L.brutus_enable_timing(0)
        res["kernels_ms"] = {k: float(np.mean(v)) for k, v in ktimes.items()}
        res["kernel_sub_batch"] = int(subs[0][0].shape[0])
        res["kernel_counts"] = [float(x) for x in np.mean(nsel_k, axis=0)]
    del engines, subs
    torch.cuda.empty_cache()
    return res


def roofline_of(res, args, config, world, with_traffic=True, issue=None):
    """SURVEY 8(d) / BASELINE.md section 4: achieved = stars/s x B_star (one float32
    read of the grid per star = 108.0 MB at 750k x 12) against the 8 TB/s HBM peak,
    for the WHOLE step.  The per-kernel entries carry each kernel's own algorithmic
    bytes and, where a rocprofv3 PMC pass exists under profiles/, its measured traffic."""
    g = float(args.nmodel) * args.nfilt * 12.
    per_gpu = res["value"] / (world if args.scaling == "weak" else world)
    ach = per_gpu * g / 1e9
    rl = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": ach / HBM_PEAK_GBS, "traffic": None,
          "algorithmic_bytes_per_star": g,
          "definition": "stars/s (per GPU) x 108.0 MB (one f32 grid read per star) / 8 TB/s"}
    if ach > HBM_PEAK_GBS:
        # (sharp posteriors: next to nothing survives the float32 pass, whose one grid read
        # serves the 64 stars of a wave -- the unit's bytes are not what the memory moved)
        rl.update(bound="vector issue (float32 pass: one grid read serves 64 stars)", reuse=64.,
                  achieved=ach / 64., frac=ach / 64. / HBM_PEAK_GBS, algorithmic_gbs_before_reuse=ach,
                  definition="stars/s x 108.0 MB / 64 (stars per staged grid row): what the memory "
                             "moves at the very least; the call is bound by vector issue, not by HBM")
    if "kernels_ms" in res:
        SB = res["kernel_sub_batch"]
        pairs = float(SB) * args.nmodel
        kern = {}
        vrows, vstale, vcommit = valu_table(SB, config) if with_traffic else ([], None, None)
        vb = valu_block(vrows, vstale, vcommit, issue, res["ms_per_step"] / max(1., float(args.batch) / SB))
        valu_k = vb["kernels"] if vb else {}
        if vb:
            rl["valu"] = vb
        for name, ms in sorted(res["kernels_ms"].items(), key=lambda kv: -kv[1]):
            alg = KERNEL_ALG_BYTES.get(name.replace("_cont", ""), None)
            e = {"avg_launch_ms": ms}
            tr = measured_traffic(name, SB, config) if with_traffic else None
            e["traffic"] = tr
            # vector issue time of the kernels timed under this name (roofline.valu's table)
            vi = sum(v["issue_ms_per_call"] for k, v in valu_k.items()
                     if TIMER_OF(k) == name) if valu_k else None
            if vi:
                e["valu_issue_ms"] = vi
                e["valu_issue_frac"] = vi / ms
            reused = False
            if alg is not None and not name.endswith("_cont"):
                # (continuation launches walk only the stars still iterating: the candidate
                # count of the call is not their work, no byte figure for them)
                ab = alg(SB, g, pairs, res["kernel_counts"])
                gbs = ab / (ms * 1e-3) / 1e9
                e["algorithmic_bytes"] = ab
                if gbs <= HBM_PEAK_GBS:
                    e["achieved_gbs"] = gbs
                else:
                    # more algorithmic bytes per second than the memory delivers: the kernel
                    # serves several units from one read (the float32 pass stages a grid tile
                    # once for the 64 stars of a wave).  Never shown as an HBM rate: the figure
                    # carries its reuse factor (measured where a PMC pass exists, else the
                    # design's 64 stars per staged row) and the kernel is priced against issue
                    reused = True
                    e["reuse"] = ab / tr if tr else float(min(64, SB))
                    e["reuse_source"] = "PMC bytes" if tr else "design: stars per staged grid row"
                    e["algorithmic_gbs_before_reuse"] = gbs
                    e["hbm_gbs"] = (tr if tr else ab / e["reuse"]) / (ms * 1e-3) / 1e9
            # what bounds it: the vector unit where issue time is most of the duration (or where
            # one read serves a wave of stars), else the memory system (by PMC bytes where measured)
            if vi and vi / ms >= 0.5:
                e["bound"] = "vector issue (%.2f of the duration is issue time)" % (vi / ms)
            elif reused:
                e["bound"] = "vector issue (one read serves %.0f stars)" % e["reuse"]
            elif tr:
                e["bound"] = "memory (%.0f GB/s of PMC traffic)" % (tr / (ms * 1e-3) / 1e9)
            elif "achieved_gbs" in e:
                e["bound"] = "memory / latency"
            else:
                e["bound"] = "latency"
            kern[name] = e
        rl["kernels"] = kern
        dom = max(res["kernels_ms"], key=res["kernels_ms"].get)
        rl["dominant_kernel"] = dom
        rl["sum_of_kernels_ms_per_sub_batch"] = float(sum(res["kernels_ms"].values()))
        # (the PMC table was taken on the mist_like grid: no traffic figure for another grid)
        traffic = measured_traffic("__total__", SB, config) if with_traffic else None
        if traffic:
            # PMC bytes of every kernel of one sub-batch call, scaled to one step
            rl["traffic"] = traffic * (float(args.batch) / SB)
            rl["traffic_over_algorithmic"] = traffic / (SB * g)
            rl["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                    "passes of tools/pmc_workload.py, not measured in this run")
            rl["traffic_from_commit"], sha = traffic_commit()
            rl["traffic_stale"] = sha != csrc_sha16()      # kernels edited since the PMC pass
    return rl


def main():
    global FULL_LINE
    args = parse()
    FULL_LINE = bool(args.full_line)
    if args.config == 5:
        return bench_cluster(args)
    cfg4 = args.config == 4
    if cfg4:
        # BASELINE configs[3]: ONE catalogue of 1M stars (configs[2]'s generator) split over the
        # ranks by parallel.shard_range; one timed pass of it is 30 s at 8 ranks, so one repeat
        args.config, args.scaling, args.steps, args.batch = 3, "strong", 250, 4000
        args.single_config = args.no_survey_grid = args.no_cluster = args.no_sharp = True
        args.e2e_stars, args.cpu_seconds = 0, 0.
        args.repeats = min(args.repeats, 2)
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
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _init_pg(dist, rank, world, dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)                  # RCCL sum of ones = ranks that really joined
        ranks_seen = int(one.item())

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
    if models is None:
        # ranks > 0 need the f32 coefficients only to synthesise their stars
        models, _, _ = synth.make_mist_like_grid(nmodel, nfilt)

    main_cfg = args.config
    other_cfg = 3 if main_cfg == 2 else 2
    res = run_config(main_cfg, args, L, grid, models, dev, world, rank, dist, torch)
    res_other = None
    if not args.single_config:
        res_other = run_config(other_cfg, args, L, grid, models, dev, world, rank, dist, torch)

    # the same configuration on the grid SURVEY 8(d) specifies (random model order)
    res_8d = None
    if not args.no_survey_grid:
        m8, _, _ = synth.make_grid(nmodel, nfilt)
        g8 = fitting.DeviceGrid(m8, device=dev) if rank == 0 or world == 1 else None
        if world > 1:
            from brutus_amd import parallel
            g8 = parallel.broadcast_grid(g8 if rank == 0 else None, nmodel, nfilt, dev, src=0)
        res_8d = run_config(main_cfg, args, L, g8, m8, dev, world, rank, dist, torch)
        del g8, m8
        torch.cuda.empty_cache()

    # the regime of the demo notebooks: sharp posteriors (a few per cent of the grid selected)
    res_sharp = e2e_sharp = None
    if not args.no_sharp and not cfg4:
        msh, _, _ = synth.make_sharp_grid(nmodel, nfilt)
        gsh = fitting.DeviceGrid(msh, device=dev) if rank == 0 or world == 1 else None
        if world > 1:
            from brutus_amd import parallel
            gsh = parallel.broadcast_grid(gsh if rank == 0 else None, nmodel, nfilt, dev, src=0)
        res_sharp = run_config(3, args, L, gsh, msh, dev, world, rank, dist, torch, star_kw=SHARP_STARS)
        if world == 1 and args.e2e_stars > 0:
            try:
                e2e_sharp = end_to_end_sharp(msh, gsh, min(4096, args.e2e_stars))
            except Exception as e:
                e2e_sharp = {"value": None, "error": repr(e)}
        del gsh, msh
        torch.cuda.empty_cache()

    # measured on this box beside the 8 TB/s spec figure: the guide's reference stream
    # (device copy, 16 B per lane; MI355X_MICROARCH.md quotes 6.29 TB/s for it)
    stream_gbs = None
    if rank == 0 and not args.no_kernel_timing:
        nbytes = 1 << 30
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.check(L.brutus_calibrate_copy16(src.data_ptr(), dst.data_ptr(), nbytes, None))
        e0.record()
        for _ in range(5):
            _lib.check(L.brutus_calibrate_copy16(src.data_ptr(), dst.data_ptr(), nbytes, None))
        e1.record()
        torch.cuda.synchronize()
        stream_gbs = 5 * 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    issue = None
    if rank == 0 and not args.no_kernel_timing:
        try:
            issue = measure_issue(L, torch, dev)
        except Exception as e:              # (an older library: the line goes without the block)
            issue = None
    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    def cfg_block(cfg, r, grid_name="mist_like"):
        return {"workload": WORKLOADS[cfg] + "; grid = " + GRIDS[grid_name],
                "grid_generator": grid_name, "selected_fraction": r["selected_fraction"],
                "nmodel": nmodel, "nfilt": nfilt,
                "stars_per_step": args.batch, "sub_batch": args.sub_batch,
                "distinct_stars_timed_per_rank": r["stars_timed_per_rank"],
                "timed_region": "brutus_fit_batch: device-resident star vectors -> "
                                "device-resident compact survivor records",
                "selected_models_last_sub_batch": r["selected_models_last_sub_batch"],
                "streams_per_gpu": max(1, args.streams),
                "parallelism": "stars sharded, %d rank(s), no data-path collective" % world}

    line = {
        "metric": "stars/sec at 750k-model x 12-band grid; achieved HBM GB/s vs peak",
        "value": res["value"], "unit": "stars/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "config": cfg_block(main_cfg, res), "ranks_seen": ranks_seen,
        "statistic": "median of %d timed repeats of the %d steps" % (res["repeats"], args.steps),
        "repeats": res["repeats"], "value_min": res["value_min"], "value_max": res["value_max"],
        "repeat_s": res["repeat_s"], "per_rank_s": res["per_rank_s"],
        "timed_region_s": res["timed_region_s"],
    }
    if cfg4:
        line["config"]["workload"] = ("configs[3]: 1M stars (configs[2]'s generator) sharded over "
                                      "%d rank(s), grid broadcast once; " % world
                                      + line["config"]["workload"])
    if "parity" in res:
        line["parity"] = res["parity"]
    rl = roofline_of(res, args, main_cfg, world, issue=issue)
    if stream_gbs is not None:
        rl["measured_stream_gbs"] = stream_gbs
    line["roofline"] = rl
    if res_other is not None:
        line["other_config"] = {
            "value": res_other["value"], "unit": "stars/s", "ms_per_step": res_other["ms_per_step"],
            "repeats": res_other["repeats"], "value_min": res_other["value_min"], "value_max": res_other["value_max"],
            "parity": res_other.get("parity"),
            "config": cfg_block(other_cfg, res_other),
            "roofline": roofline_of(res_other, args, other_cfg, world, issue=issue)}
    if res_8d is not None:
        line["survey8d_grid"] = {
            "value": res_8d["value"], "unit": "stars/s", "ms_per_step": res_8d["ms_per_step"],
            "repeats": res_8d["repeats"], "value_min": res_8d["value_min"], "value_max": res_8d["value_max"],
            "parity": res_8d.get("parity"),
            "config": cfg_block(main_cfg, res_8d, "survey8d"),
            "roofline": roofline_of(res_8d, args, main_cfg, world, with_traffic=False)}
    if res_sharp is not None:
        line["sharp_posterior"] = {
            "value": res_sharp["value"], "unit": "stars/s", "ms_per_step": res_sharp["ms_per_step"],
            "repeats": res_sharp["repeats"], "value_min": res_sharp["value_min"],
            "value_max": res_sharp["value_max"], "parity": res_sharp.get("parity"),
            "selected_fraction": res_sharp["selected_fraction"],
            "config": {"workload": "750k-model x 12-band grid, Av+Rv free, S/N 50 photometry in every "
                                   "band, parallax at S/N 10 for every star; grid = synth.make_sharp_grid "
                                   "(colours quadratic in the band index: not degenerate with reddening)",
                       "stars_per_step": args.batch, "sub_batch": args.sub_batch,
                       "selected_fraction": res_sharp["selected_fraction"]},
            "roofline": roofline_of(res_sharp, args, 3, world, with_traffic=False),
            "fit_end_to_end": e2e_sharp}
    if world == 1 and args.e2e_stars > 0:
        kw = dict(rvlim=(3.32, 3.32)) if main_cfg == 2 else dict()
        line["fit_end_to_end"] = end_to_end(models, grid, None, args.e2e_stars, kw, main_cfg == 3,
                                            check=not args.no_parity)
    if world == 1 and args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(main_cfg, nmodel, nfilt, args.cpu_seconds)
    if world == 1 and not args.no_cluster:
        # BASELINE configs[4] (cluster mode) rides along: < 1 s, so that the driver's own
        # run times it too (`python bench.py --config 5` prints the same block as a line)
        ca = argparse.Namespace(**vars(args))
        ca.steps, ca.warmup = 200, 20
        ca.cpu_seconds = min(args.cpu_seconds, 10.)     # (its CPU leg: ~1 s of numpy on 100 objects)
        try:
            line["cluster_mode"] = bench_cluster(ca, emit=False)
        except Exception as e:          # never at the expense of the headline line
            line["cluster_mode"] = {"value": None, "error": repr(e)}
    emit_line(line)
    if world > 1:
        dist.destroy_process_group()


# ---- the ONE line on stdout -------------------------------------------------------------------
# The driver parses the last stdout line and keeps an 8 KB tail of it: the line carries the
# contract's keys, the `roofline` / `cpu_baseline` objects, the parity verdicts and ONE number
# per supplementary block; every per-kernel table goes to bench_detail.json beside this file
# (and to gpurun_out/ when that directory exists: the only one that travels back from a GPU box).
LINE_LIMIT = 8192
DETAIL_FILE = "bench_detail.json"


def _r(x, sig=6):
    """Floats to `sig` significant digits (the line is read by people and a size-capped parser)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if np.isfinite(x) else None
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _parity_short(p):
    """A parity block without its prose."""
    if not p:
        return None
    return {k: v for k, v in p.items() if k not in ("against", "note")}


def _kernel_short(name, k):
    """One kernel of roofline.kernels as the line shows it: duration, the ceiling that binds it,
    and -- only with its qualifier -- a bytes-per-second figure.  A kernel whose one read serves
    many stars (the float32 pass: 64 stars per grid tile) is priced against vector issue and
    carries `reuse`; an algorithmic rate above the memory's peak never appears as a plain
    `achieved_gbs`."""
    e = {"ms": k["avg_launch_ms"]}
    vf = k.get("valu_issue_frac")
    if vf:
        e["valu_issue_frac"] = vf
    if k.get("traffic"):
        e["hbm_gbs"] = k["traffic"] / (k["avg_launch_ms"] * 1e-3) / 1e9
    if "reuse" in k:
        e["reuse"] = k["reuse"]
    e["bound"] = k.get("bound", "latency").split(" (")[0]
    return e


def _roofline_short(rl, top=4):
    out = _pick(rl, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                     "traffic_stale", "algorithmic_bytes_per_star", "measured_stream_gbs",
                     "dominant_kernel", "sum_of_kernels_ms_per_sub_batch"))
    out["launch_unit"] = "brutus_fit_batch (one sub-batch call): achieved = stars/s x 108.0 MB"
    if rl.get("traffic"):
        out["traffic_is"] = "PMC HBM bytes (FETCH_SIZE + WRITE_SIZE, corrected) per step = all sub-batch calls of it"
    if rl.get("valu"):
        v = rl["valu"]
        out["valu"] = dict(_pick(v, ("frac", "issue_ms_per_call", "call_ms", "stale")),
                           bound="vector instruction issue",
                           issue_ns=_pick(v["issue_ns_per_wave_inst"], ("f32_ns", "f64_ns", "trans32_ns")))
    if rl.get("kernels"):
        ks = sorted(rl["kernels"].items(), key=lambda kv: -kv[1]["avg_launch_ms"])[:top]
        out["kernels"] = {n: _kernel_short(n, k) for n, k in ks}
    return out


def _block_short(b):
    """ONE-number summary of a supplementary workload block: rate, fraction of the HBM roofline of
    the whole step, selected fraction, parity verdict."""
    if b is None:
        return None
    out = _pick(b, ("value", "unit", "ms_per_step", "value_min", "value_max", "error"))
    rl = b.get("roofline") or {}
    if "frac" in rl:
        out["frac"] = rl["frac"]
    if rl.get("valu"):
        out["valu_frac"] = rl["valu"]["frac"]
    if rl.get("dominant_kernel"):
        out["dominant_kernel"] = rl["dominant_kernel"]
    cfg = b.get("config") or {}
    if "selected_fraction" in cfg:
        out["selected_fraction"] = cfg["selected_fraction"]
    if "workload" in cfg:
        out["workload"] = cfg["workload"].split(";")[0][:90]
    if b.get("parity"):
        out["parity"] = _parity_short(b["parity"])
    return out


def compact_line(full):
    """The driver's line from the full record (which goes to bench_detail.json)."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                        "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                        "ranks_seen", "statistic", "repeats", "value_min", "value_max", "repeat_s",
                        "per_rank_s", "timed_region_s", "star_points_per_s", "plugin_ms_per_step",
                        "library_ms_per_step", "revisited_table_evaluations_per_s"))
    if "parity" in full:
        line["parity"] = _parity_short(full["parity"])
    rl = full.get("roofline")
    if rl is not None:
        line["roofline"] = _roofline_short(rl) if rl.get("unit") == "GB/s" else rl
    g8 = full.get("survey8d_grid")
    if g8 is not None:
        # the SURVEY 8(d)-literal grid (random model order) beside the headline: same workload,
        # same definition of `frac`
        line["survey8d_value"] = g8["value"]
        line["survey8d_frac"] = g8["roofline"]["frac"]
        line["survey8d_grid"] = _block_short(g8)
    if full.get("other_config") is not None:
        line["other_config"] = _block_short(full["other_config"])
    sp = full.get("sharp_posterior")
    if sp is not None:
        b = _block_short(sp)
        # 0.1 % of the grid selected: the call IS the float32 pass, one grid read per 64 stars ->
        # stars/s x 108 MB exceeds what the memory delivers by that reuse; priced against issue
        b.pop("frac", None)
        b["bound"] = "vector issue (float32 pass; one grid read serves 64 stars)"
        b["reuse"] = sp["roofline"].get("reuse", 1.)
        b["hbm_gbs_after_reuse"] = sp["roofline"]["achieved"]
        b["selected_fraction"] = sp.get("selected_fraction")
        fe = sp.get("fit_end_to_end")
        if fe:
            b["fit_end_to_end"] = _pick(fe, ("value", "unit", "stars", "error"))
        line["sharp_posterior"] = b
    fe = full.get("fit_end_to_end")
    if fe is not None:
        out = _pick(fe, ("value", "unit", "stars", "statistic"))
        for k in ("numpy_per_object", "device_lnpost", "host_lnpost", "python_hooks"):
            if k in fe:
                out[k] = fe[k].get("value")
        if "parity" in fe:
            out["parity"] = _parity_short(fe["parity"])
        line["fit_end_to_end"] = out
    cb = full.get("cpu_baseline")
    if cb is not None:
        out = _pick(cb, ("value", "unit", "cores", "kind", "host_cores", "sample"))
        sc = cb.get("single_core")
        if sc:
            out["single_core_value"] = sc.get("value")
        line["cpu_baseline"] = out
    cm = full.get("cluster_mode")
    if cm is not None:
        out = _pick(cm, ("value", "unit", "ms_per_step", "plugin_ms_per_step", "library_ms_per_step",
                         "revisited_table_evaluations_per_s", "error"))
        if cm.get("roofline"):
            out["roofline"] = _pick(cm["roofline"], ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms"))
            out["roofline"]["bound"] = "vector f64"
        if cm.get("cpu_baseline"):
            out["cpu_baseline"] = _pick(cm["cpu_baseline"], ("value", "unit", "cores", "kind"))
        line["cluster_mode"] = out
    line["detail"] = DETAIL_FILE
    return _r(line)


FULL_LINE = False      # --full-line


def emit_line(full):
    """Write the full record to bench_detail.json, print the compact line (one line, < 8 KB)."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(full, f, indent=1)
            except OSError:
                pass
    if FULL_LINE:
        print(json.dumps(full))
        sys.stdout.flush()
        return
    text = json.dumps(compact_line(full), separators=(",", ":"))
    if len(text) >= LINE_LIMIT:
        raise SystemExit("bench line is %d bytes (limit %d): trim compact_line" % (len(text), LINE_LIMIT))
    print(text)
    sys.stdout.flush()


if __name__ == "__main__":
    main()
