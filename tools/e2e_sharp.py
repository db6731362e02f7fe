"""End-to-end fit() on the sharp-posterior workload of bench.py (GPU box): stars/s.
    BRUTUS_AMD_LIB=... python tools/e2e_sharp.py [stars]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
import bench
from brutus_amd import fitting, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
models = synth.make_sharp_grid(750000, 12)[0]
grid = fitting.DeviceGrid(models)
r = bench.end_to_end_sharp(models, grid, n)
print("%s sharp fit(): %.0f stars/s" % (os.environ.get("BRUTUS_AMD_LIB", "in-tree").split("/")[-1], r["value"]))
