"""Multi-GPU layout of the hot path: utterances are independent, so they shard across ranks with
NO collective in the decode loop.  The only communication is one broadcast of the
folded weight blob at start-up (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests) and,
optionally, a gather of per-rank results on the host side.

The reference has no inference-side distributed code to mirror (inference.py:51-53 pins one
device; Triton uses replicas, runtime/gpu_triton/model_repo/generator/config.pbtxt:48-53).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialises torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE if needed."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("WETTS_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def launch_ranks(n, script, argv, require_gpus=True):
    """Starts `n` ranks of `script argv...` on THIS node, one process per GPU, by re-executing it
    under torch.distributed.run with a 127.0.0.1 rendezvous on a free port; returns the exit code.
    This is what `python bench.py --gpus N` does when it is not already running under torchrun.
    With `require_gpus` it refuses (code 3, message on stderr, nothing launched) when the node
    exposes fewer than `n` HIP devices -- a smaller job must never run under a larger label."""
    import socket
    import subprocess
    import sys
    if require_gpus:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < n:
            sys.stderr.write(f"{os.path.basename(script)}: {n} ranks requested but this node exposes "
                             f"{n_dev} HIP device(s); refusing to run a smaller job under a larger "
                             "label\n")
            return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)]
    env = dict(os.environ)
    # the deployment image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only supports dmabuf IPC; without
    # it RCCL fails in hipIpcGetMemHandle); a child environment built here must carry the same setting, and an
    # operator's own value wins
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd + list(argv), env=env)


def shard_utterances(lengths, world_size):
    """Longest-processing-time-first deal of utterances to ranks.

    lengths: per-utterance cost proxy (phoneme counts; frames are ~proportional).
    Returns a list (len world_size) of index lists; every index appears exactly once; ranks'
    total cost differs by at most one utterance's cost from the greedy optimum.  Within a rank
    the indices are sorted by length (descending) so consecutive batches pad little."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    counts = [0] * world_size
    shards = [[] for _ in range(world_size)]
    cap = -(-len(lengths) // world_size)  # keep batch counts equal (weak scaling: B/rank fixed)
    for i in order:
        cands = [r for r in range(world_size) if counts[r] < cap]
        r = min(cands, key=lambda q: (loads[q], q))
        shards[r].append(i)
        loads[r] += int(lengths[i])
        counts[r] += 1
    return shards


def broadcast_blob(blob, src=0):
    """One broadcast of the folded weight blob (float32, contiguous) from rank `src`.
    ~123 MB for v1; ring/tree over xGMI is per-link bound (~153 GB/s/link) => ~1-2 ms."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def gather_objects(obj, dst=0):
    """Host-side gather of small per-rank results (lengths, timings)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def unshard(shards, per_rank_results):
    """Inverse of shard_utterances for host-side results: puts results back in input order."""
    n = sum(len(s) for s in shards)
    out = [None] * n
    for idxs, res in zip(shards, per_rank_results):
        for i, r in zip(idxs, res):
            out[i] = r
    return out
