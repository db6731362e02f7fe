#!/bin/bash
# flux continuation rounds of the device-driven call: throughput and batches handed back
for r in 1 2 3 4; do
BRUTUS_FLUX_ROUNDS=$r python - <<PY
import os, sys, json, subprocess, ctypes as C
sys.path.insert(0, os.getcwd())
import bench
sys.argv = ["bench.py", "--single-config", "--steps", "20", "--warmup", "4", "--repeats", "3", "--cpu-seconds", "0",
            "--e2e-stars", "0", "--no-survey-grid", "--no-cluster", "--no-parity", "--no-kernel-timing", "--config", os.environ.get("CFG", "2")]
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
from brutus_amd import _lib
a, b = C.c_int64(0), C.c_int64(0)
_lib.lib().brutus_debug_fit_stats(C.byref(a), C.byref(b))
print("rounds $r cfg", os.environ.get("CFG", "2"), round(d["value"]), "calls", a.value, "handed back", b.value)
PY
done
