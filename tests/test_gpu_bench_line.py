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
