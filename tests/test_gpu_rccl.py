"""-m gpu: the sharded search over a real RCCL communicator, one rank per visible GPU, launched the way the driver launches bench.py
(python -m torch.distributed.run, 127.0.0.1 rendezvous).  The data path is the library's (csrc/shard.hip): local traversal +
ncclAllGather of the packed per-shard block + merge kernel; torch only distributes the RCCL id.  On this pool's one-GPU boxes the
group has one rank (RCCL still initialises, gathers and merges — the identity); on a multi-GPU node the same test is a real N > 1
RCCL run: every rank must hold the same merged answer and its recall against exact search must hold.  World sizes 2 / 3 of the
partition / exchange logic are also covered over gloo in tests/test_shard_gloo.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_search_over_rccl(la, gpu, tmp_path):
    nproc = min(max(la.device_count(), 1), 6)  # at most 6 processes may use the card(s) of a box at once
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py"), out],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    assert int(z["world"]) == nproc and bool(z["same"])
    assert (z["counts"] == 10).all() and (np.diff(z["dists"], axis=1) >= 0).all()
    assert float(z["recall"]) >= 0.9
    if nproc == 1:
        assert (z["keys"] == z["local_keys"]).all()
