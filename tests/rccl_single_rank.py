"""Helper of tests/test_gpu_rccl.py: ONE rank on the `nccl` backend (= RCCL on ROCm) doing what bench.py's N > 1
start-up does -- process group over a 127.0.0.1 rendezvous, barrier, the weight-blob broadcast, the min / max
fingerprint all-reduces -- then synthesising with the received blob.  World size 1 cannot move bytes between GPUs, but
it loads librccl, creates the communicator on the device (HSA_ENABLE_IPC_MODE_LEGACY as the image exports it) and
runs the collectives' kernels, which is everything a 1-GPU box can exercise before the driver's 8-GPU run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from wetts_amd import SynthesizerTrn, checkpoint, config, sharding, synth  # noqa: E402
from bench import blob_fingerprint  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rank, local_rank, world = sharding.env_world()
    assert world == 1 and rank == 0
    # the product's own start-up: process group bound to the device (device_id), the default timeout, NCCL_DEBUG=WARN
    sharding.init_process_group(backend="nccl", force=True)
    assert dist.get_backend() == "nccl"
    mname, n_vocab, n_spk = "tiny", 40, 3
    net = SynthesizerTrn(n_vocab, 513, 32, n_speakers=n_spk, **config.MODEL_CONFIGS[mname])
    cfg = net.cfg
    blob = checkpoint.pack_blob(cfg, synth.make_state_dict(cfg, 0)).to(dev)
    before = blob_fingerprint(blob)
    dist.barrier()
    dist.broadcast(blob, src=0)  # sharding.broadcast_blob skips the call at world 1; here it must run
    torch.cuda.synchronize()
    fp = blob_fingerprint(blob)
    lo, hi = fp.clone(), fp.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    ok = bool(torch.equal(lo, hi)) and bool(torch.equal(fp, before)) and int(one.item()) == 1
    net.load_blob(blob)
    x = torch.randint(0, n_vocab, (2, 9), device=dev)
    o, *_ = net.infer(x, torch.tensor([9, 5], device=dev), sid=torch.tensor([0, 2], device=dev))
    torch.cuda.synchronize()
    got = sharding.gather_objects({"rank": rank, "samples": int(o.shape[-1])}, dst=0)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"ok": ok and bool(torch.isfinite(o).all()), "backend": "nccl", "gathered": got,
                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))


if __name__ == "__main__":
    main()
