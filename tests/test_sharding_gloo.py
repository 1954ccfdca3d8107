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
