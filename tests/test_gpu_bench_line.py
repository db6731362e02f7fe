"""The bench line's contract on a small grid (GPU): one JSON line with the driver's keys, the
SURVEY 8(d) protocol (timed repeats, median / min / max), the `roofline` and `cpu_baseline`
objects and the parity blocks -- all of them green."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_KEYS = {
    "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
    "scaling", "vs_baseline", "dtype", "data", "config", "ranks_seen", "statistic", "repeats",
    "value_min", "value_max", "repeat_s", "per_rank_s", "timed_region_s", "parity", "roofline",
    "survey8d_value", "survey8d_frac", "survey8d_grid", "other_config", "sharp_posterior",
    "fit_end_to_end", "cpu_baseline", "cluster_mode", "detail"}


def _no_rate_above_peak_without_reuse(node, path=""):
    """Every bytes-per-second figure anywhere in the record: at most the HBM peak, unless the
    object it sits in says how (a `reuse` factor) and the key says it is not an HBM rate."""
    if isinstance(node, dict):
        for k, v in node.items():
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                if k in ("achieved_gbs", "hbm_gbs", "hbm_gbs_after_reuse", "measured_stream_gbs") or (
                        k == "achieved" and node.get("unit") == "GB/s"):
                    assert v <= 8000., (path, k, v)
                if k == "algorithmic_gbs_before_reuse":
                    assert node.get("reuse", 0.) > 1., (path, k)
                if k == "frac" and node.get("unit") in ("GB/s", "TFLOP/s"):
                    assert 0. <= v <= 1., (path, v)
            else:
                _no_rate_above_peak_without_reuse(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            _no_rate_above_peak_without_reuse(v, "%s[%d]" % (path, i))


def test_compact_line_of_a_full_size_record_fits_the_driver():
    """CPU: the line built from last round's full-size record (profiles/) is < 8 KB and has
    exactly the key set the small-grid GPU run is held to."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "r05_v7_bench_default.json")) as f:
        full = json.load(f)
    text = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(text) < 8192 and "\n" not in text
    d = json.loads(text)
    assert set(d) == LINE_KEYS, sorted(set(d) ^ LINE_KEYS)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"]


@pytest.mark.gpu
def test_bench_line_contract_small_grid():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nmodel", "60000", "--steps", "3",
           "--warmup", "1", "--batch", "128", "--sub-batch", "64", "--repeats", "3",
           "--e2e-stars", "64", "--cpu-seconds", "1", "--cluster-stars", "300"]
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode("utf-8", "replace")[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    # the driver keeps an 8 KB tail of stdout and parses the last line: the line must fit
    # (round 5's 24.8 KB line left BENCH_r05.json with parsed = null)
    assert len(lines[0]) < 8192, len(lines[0])
    assert out.stdout.decode().strip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    # exactly the key set of the full-size run (bench.compact_line builds both)
    assert set(d) == LINE_KEYS, sorted(set(d) ^ LINE_KEYS)
    # ... and the per-kernel tables are in the detail file beside bench.py
    with open(os.path.join(ROOT, d["detail"])) as f:
        full = json.load(f)
    assert full["value"] == pytest.approx(d["value"], rel=1e-5)
    assert "kernels" in full["roofline"] and "kernels" in full["other_config"]["roofline"]
    _no_rate_above_peak_without_reuse(d)
    _no_rate_above_peak_without_reuse(full)
    assert d["survey8d_frac"] == d["survey8d_grid"]["frac"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "repeats", "value_min", "value_max", "repeat_s", "per_rank_s", "parity"):
        assert k in d, k
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["repeats"] == 3
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    assert d["value_min"] <= d["value"] <= d["value_max"] and len(d["repeat_s"]) == 3
    # (the line's floats carry six significant digits)
    assert d["ms_per_step"] * 3 == pytest.approx(1e3 * sorted(d["repeat_s"])[1], rel=1e-4)
    rl = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and rl["frac"] == pytest.approx(rl["achieved"] / 8000., rel=1e-5)
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
    assert sp["bound"].startswith("vector issue") and "frac" not in sp


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
