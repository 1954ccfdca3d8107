"""GPU tier: the RCCL start-up path of the N > 1 job on the one GPU a test box has (see tests/rccl_single_rank.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_one_broadcast_and_fingerprint():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # what the image exports and sharding.launch_ranks passes on
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank.py")], capture_output=True,
                       text=True, env=env, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["backend"] == "nccl" and d["gathered"][0]["samples"] > 0
    # the process group is bound to its device at init (sharding.init_process_group passes device_id): c10d must not
    # have had to guess it -- with 8 ranks that guess is "can cause a hang if rank to GPU mapping is heterogeneous"
    assert "Guessing device ID" not in r.stderr and "using the device under current context" not in r.stderr, r.stderr[-3000:]
    assert "[wetts rank 0/1] nccl on cuda:0" in r.stderr
