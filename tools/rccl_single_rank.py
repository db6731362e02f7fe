"""The RCCL calls of bench.py / parallel.py with ONE rank (all a one-GPU box allows: RCCL refuses two ranks on
one device): process group on the nccl backend with device_id, all_reduce, barrier, broadcast of the grid blob,
broadcast_array, destroy.  Not a scaling test -- a check that the calls are accepted by RCCL as written."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from brutus_amd import fitting, parallel, synth  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
one = torch.ones(1, dtype=torch.int32, device=dev)
dist.all_reduce(one)
assert int(one.item()) == 1
dist.barrier()
models, _, _ = synth.make_mist_like_grid(750000, 12)
grid = fitting.DeviceGrid(models, device=dev)
t1 = time.time()
g2 = parallel.broadcast_grid(grid, 750000, 12, dev, src=0)
torch.cuda.synchronize()
t2 = time.time()
assert g2.nmodel == 750000 and torch.equal(g2.soa, grid.soa)
arr = parallel.broadcast_array(np.arange(12.).reshape(3, 4), device=dev)
assert np.array_equal(arr, np.arange(12.).reshape(3, 4))
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
dist.barrier()
dist.destroy_process_group()
print("rccl single-rank ok: init %.2f s, grid blob broadcast (%.0f MB) %.3f s, backend %s"
      % (t1 - t0, grid.soa.numel() * 4 / 1e6, t2 - t1, "nccl"))
