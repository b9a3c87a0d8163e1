"""-m gpu: the exchange step of the sharded search over a real RCCL process group (backend "nccl"), launched the way the driver
launches bench.py (python -m torch.distributed.run, 127.0.0.1 rendezvous).  The box has one GPU, so the group has one rank: this
checks that RCCL initialises on this stack, that the packed int32 all-gather and the HIP merge kernel run on a side stream, and
that a one-shard exchange is the identity.  World sizes 2 / 3 are covered over gloo in tests/test_shard_gloo.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exchange_over_rccl_world_of_one(gpu, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py"), out],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    assert (z["gk"][0] == z["keys"]).all() and (z["gd"][0] == z["dists"]).all() and (z["gc"][0] == z["counts"]).all()
    assert (z["mk"] == z["keys"]).all() and (z["md"] == z["dists"]).all() and (z["mc"] == z["counts"]).all()
    assert (z["counts"] == 10).all() and (np.diff(z["dists"], axis=1) >= 0).all()
