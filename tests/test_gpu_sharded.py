"""`parallel.fit_sharded` with the real kernels: two, three and four ranks (all on
cuda:0 over gloo -- RCCL refuses two ranks on one device -- so that it runs on
a one-GPU box) must produce the file an unsharded run produces."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
@pytest.mark.parametrize("world,writer", [(2, "rank0"), (3, "rank0"), (4, "rank0"), (3, "per_rank")])
def test_fit_sharded_equals_unsharded(world, writer):
    """(`writer="per_rank"`: every rank its own part file, the index of virtual datasets read back
    on rank 0 -- same comparison.)"""
    env = dict(os.environ, BRUTUS_BENCH_ONE_DEVICE="1", BRUTUS_BENCH_BACKEND="gloo",
               MASTER_ADDR="127.0.0.1", SHARDED_WRITER=writer)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "sharded_smoke.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode("utf-8", "replace")
    assert out.returncode == 0, text[-3000:]
    assert "sharded_smoke ok: %d ranks" % world in text, text[-3000:]
