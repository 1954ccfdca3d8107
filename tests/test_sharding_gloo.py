"""CPU tier: the N>1 layout (utterance sharding + one weight broadcast) with world_size-2 gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wetts_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_utterances_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    lens = rng.integers(32, 129, size=512).tolist()
    for world in (1, 2, 4, 8):
        shards = sharding.shard_utterances(lens, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(512))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= 128  # LPT: within one utterance of each other
        for s in shards:  # sorted by length inside a rank => little padding per batch
            assert [lens[i] for i in s] == sorted((lens[i] for i in s), reverse=True)
    res = [[f"r{r}_{i}" for i in s] for r, s in enumerate(sharding.shard_utterances(lens, 4))]
    back = sharding.unshard(sharding.shard_utterances(lens, 4), res)
    assert all(b is not None for b in back) and len(back) == 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from wetts_amd import checkpoint, config, sharding as sh, synth
    r, lr, w = sh.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = config.make_config(config.MODEL_CONFIGS["tiny"], 20, 2)
    n = checkpoint.blob_numel(cfg)
    if rank == 0:
        blob = checkpoint.pack_blob(cfg, synth.make_state_dict(cfg, 5))
    else:
        blob = torch.zeros(n, dtype=torch.float32)
    sh.broadcast_blob(blob, src=0)
    lens = list(range(40, 40 + 10))
    mine = sh.shard_utterances(lens, world)[rank]
    got = sh.gather_objects({"rank": rank, "idx": mine, "sum": float(blob.double().sum())}, dst=0)
    if rank == 0:
        torch.save(got, os.path.join(out_dir, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_shard(tmp_path):
    from wetts_amd import checkpoint, config, synth
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(tmp_path / "gathered.pt", weights_only=False)
    cfg = config.make_config(config.MODEL_CONFIGS["tiny"], 20, 2)
    ref = float(checkpoint.pack_blob(cfg, synth.make_state_dict(cfg, 5)).double().sum())
    assert [g["rank"] for g in got] == [0, 1]
    assert all(abs(g["sum"] - ref) < 1e-9 for g in got)  # every rank holds rank 0's weights
    assert sorted(got[0]["idx"] + got[1]["idx"]) == list(range(10))


_RANK_SCRIPT = """
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from wetts_amd import sharding as sh
rank, local_rank, world = sh.init_process_group(backend="gloo")
blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
sh.broadcast_blob(blob, src=0)
mine = sh.shard_utterances(list(range(32, 48)), world)[rank]
got = sh.gather_objects(dict(rank=rank, world=dist.get_world_size(), n=len(mine),
                             s=float(blob.sum())), dst=0)
if rank == 0:
    torch.save(got, sys.argv[1])
dist.barrier()
dist.destroy_process_group()
"""


def test_launch_ranks_spawns_the_ranks_itself(tmp_path):
    """The code path behind `python bench.py --gpus N` (no torchrun around it): launch_ranks
    re-executes a script under torch.distributed.run; here with gloo and no GPU requirement."""
    script = tmp_path / "rank_script.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    out = tmp_path / "got.pt"
    rc = sharding.launch_ranks(2, str(script), [str(out)], require_gpus=False)
    assert rc == 0
    got = torch.load(out, weights_only=False)
    assert [g["rank"] for g in got] == [0, 1] and all(g["world"] == 2 for g in got)
    assert all(g["n"] == 8 and g["s"] == 499500.0 for g in got)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a node with fewer than 2 GPUs must fail loudly instead of
    printing a 1-GPU line (round-1 VERDICT item 1).  Here: no GPU at all => exit code 3."""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("node has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WETTS_BENCH_SINGLE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3 and p.stdout.strip() == ""
    assert "refusing" in p.stderr
