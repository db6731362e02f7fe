"""The bench line's contract on a small grid (GPU): one JSON line with the driver's keys, the
SURVEY 8(d) protocol (timed repeats, median / min / max), the `roofline` and `cpu_baseline`
objects and the parity blocks -- all of them green."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract_small_grid():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nmodel", "60000", "--steps", "3",
           "--warmup", "1", "--batch", "128", "--sub-batch", "64", "--repeats", "3",
           "--e2e-stars", "64", "--cpu-seconds", "1", "--cluster-stars", "300"]
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "repeats", "value_min", "value_max", "repeat_s", "per_rank_s", "parity"):
        assert k in d, k
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["repeats"] == 3
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    assert d["value_min"] <= d["value"] <= d["value_max"] and len(d["repeat_s"]) == 3
    assert abs(d["ms_per_step"] * 3 - 1e3 * sorted(d["repeat_s"])[1]) < 1e-6
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / 8000.) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    for blk in (d, d["other_config"], d["survey8d_grid"]):
        p = blk["parity"]
        assert p["sel_equal"] and p["k1_k2_equal"] and p["max_rel"] < 1e-8, p
    pe = d["fit_end_to_end"]["parity"]
    assert pe["model_idx_equal"] and pe["max_rel_log_evid"] < 1e-6, pe
    assert d["cluster_mode"]["value"] > 0
    # the cluster block carries its own CPU leg and prices the kernel against the vector
    # float64 peak (HBM sees next to nothing of it)
    cm = d["cluster_mode"]
    assert cm["cpu_baseline"]["value"] > 0 and cm["cpu_baseline"]["kind"] == "port"
    assert cm["roofline"]["unit"] == "TFLOP/s" and 0. < cm["roofline"]["frac"] < 1.
    # sharp posteriors: same path, a few per cent of the grid selected, parity green
    sp = d["sharp_posterior"]
    assert sp["value"] > 0 and sp["selected_fraction"] < 0.2, sp["selected_fraction"]
    assert sp["parity"]["sel_equal"] and sp["parity"]["k1_k2_equal"] and sp["parity"]["max_rel"] < 1e-8
    assert sp["fit_end_to_end"]["value"] > 0
    # no kernel claims more bytes per second than the memory has without saying how
    for name, k in rl["kernels"].items():
        if k.get("achieved_gbs", 0.) > 8000. and k.get("traffic"):
            assert k["reuse"] > 1., name


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device_over_gloo():
    """The N > 1 path of the line -- process group, grid broadcast, shard_range, per-rank times,
    max over ranks -- executed for real: two ranks share cuda:0 (BRUTUS_BENCH_ONE_DEVICE=1) and
    talk over gloo (RCCL refuses two ranks on one device).  configs[3]'s shape scaled down:
    ONE catalogue split over the ranks (--scaling strong)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BRUTUS_BENCH_ONE_DEVICE="1", BRUTUS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "3", "--scaling", "strong",
           "--nmodel", "60000", "--steps", "4", "--warmup", "1", "--batch", "128", "--sub-batch", "64",
           "--repeats", "2", "--streams", "2", "--single-config", "--no-survey-grid", "--no-sharp",
           "--no-cluster", "--e2e-stars", "0", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong"
    assert len(d["per_rank_s"]) == 2 and all(t > 0 for t in d["per_rank_s"])
    assert d["config"]["distinct_stars_timed_per_rank"] == 4 * 128 // 2
    assert "2 rank(s)" in d["config"]["parallelism"]
    p = d["parity"]
    assert p["sel_equal"] and p["k1_k2_equal"] and p["max_rel"] < 1e-8, p
