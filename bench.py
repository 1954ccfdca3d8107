#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X VITS hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one SynthesizerTrn.infer() over one synthetic batch per rank: Baker VITS (v1 config,
22.05 kHz, SDP, ResBlock1, C0=512), batch = 16 utterances x 128 phonemes, fp32
(BASELINE.json configs[1]; `--config multilingual | aishell3 | stress48k` select configs[2..4]).
Inputs are resident in HBM when the timed region starts; outputs stay in HBM.  One process per
GPU: `--gpus N` without a torchrun environment starts the N ranks itself (and refuses if the node
has fewer GPUs); weights are broadcast once from rank 0 (RCCL), then every rank decodes its own
utterance shard with no collective in the loop (weak scaling).

Prints ONE JSON line on rank 0 (see the driver contract).  Extra keys:
  roofline     dominant kernel = the MRF ResBlock conv stack (conv_mfma_kernel), timed live with
               HIP events recorded on the launch stream inside the timed steps
  cpu_baseline the oracle (oracle/vits_oracle.py, a port of the reference running the same ATen
               CPU kernels) timed on this box's host cores on a bounded sample of the workload,
               swept over 1 / 8 / 16 / 32 threads; the best is reported with its thread count
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= f32 vector)
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="baker", choices=["baker", "multilingual", "aishell3",
                                                          "stress48k"],
                    help="BASELINE.json workload preset: baker = configs[1] (the headline), "
                         "multilingual = configs[2], aishell3 = configs[3], stress48k = configs[4]")
    # overrides of the preset (None = take the preset's value)
    ap.add_argument("--model", default=None)
    ap.add_argument("--batch", type=int, default=None, help="utterances per rank per step")
    ap.add_argument("--phonemes", type=int, default=None)
    ap.add_argument("--speakers", type=int, default=None, help="rows of the speaker table")
    ap.add_argument("--ragged", action="store_true", help="Tx ~ U{32..phonemes} (configs[3] style)")
    ap.add_argument("--buckets", type=int, default=0,
                    help="padded sub-batches per step (0 = auto: 1 for equal lengths, 4 for ragged batches; "
                         "SURVEY 8e: each rank buckets its length-sorted shard)")
    ap.add_argument("--decoder-dtype", default=None, choices=["f32", "bf16", "f16", "uint8"],
                    help="HiFi-GAN arithmetic; the headline metric is quoted at f32 (uint8 = the "
                         "export_onnx.py --quant dynamic-quantisation variant)")
    ap.add_argument("--flow-dtype", default=None, choices=["f32", "bf16", "f16"],
                    help="arithmetic of the flow's WaveNet layers (wetts_set_flow_precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2, help="utterances in the CPU sample")
    # secondary mode: streaming (chunked decoder) latency at B = 1 instead of the throughput step
    ap.add_argument("--stream", action="store_true", help="print the streaming-latency JSON instead")
    ap.add_argument("--stream-phonemes", type=int, default=64)
    ap.add_argument("--stream-chunk", type=int, default=40)
    ap.add_argument("--stream-pad", type=int, default=10)
    ap.add_argument("--stream-reps", type=int, default=30)
    ap.add_argument("--stream-cpu", action="store_true", help="add the CPU-port timing of one window")
    ap.add_argument("--stream-unfused", action="store_true", help="one launch per conv (diagnostic)")
    return ap.parse_args()


def make_inputs(model_name, n_vocab, n_speakers, total_utts, tx, ragged, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, n_vocab, (total_utts, tx), generator=g)
    if ragged:
        lens = torch.randint(32, tx + 1, (total_utts,), generator=g)
    else:
        lens = torch.full((total_utts,), tx, dtype=torch.long)
    sid = torch.randint(0, max(1, n_speakers), (total_utts,), generator=g)
    return x, lens, sid


def stream_bench(args):
    """`bench.py --stream`: chunked-decoder latency at B = 1 (the reference's streaming
    clients inference_onnx.py:37-76, vits_model.cc:96-153).  Prints one JSON line; with --stream-cpu the
    cpu_baseline leg times the oracle (CPU port) on the first decoder window."""
    import statistics
    import numpy as np
    from wetts_amd import SynthesizerTrn, checkpoint, config, synth
    from wetts_amd.session import (DecoderSession, EncoderSession, InferenceSession, depad_bounds,
                                   get_chunks)

    def med(f, n):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)

    dev = torch.device("cuda:0")
    args.model = args.model or "v1"
    net = SynthesizerTrn(256, 513, 32, n_speakers=1, **config.MODEL_CONFIGS[args.model]).to(dev)
    sr = config.SAMPLING_RATES[args.model]
    cfg = net.cfg
    sd = synth.make_state_dict(cfg, seed=0)
    net.load_blob(checkpoint.pack_blob(cfg, sd).to(dev))
    hop = net.hop_length
    if args.stream_unfused:
        net.set_decoder_dtype(torch.float32, fused=False)
    torch.manual_seed(0)
    ids = torch.randint(0, 256, (1, args.stream_phonemes)).numpy()
    feeds = {"input": ids, "input_lengths": np.array([args.stream_phonemes], dtype=np.int64),
             "scales": np.array([[0.667, 1.0, 0.8]], dtype=np.float32),
             "sid": np.array([0], dtype=np.int64)}
    enc, full = EncoderSession(net), InferenceSession(net)
    dec, decg = DecoderSession(net), DecoderSession(net, use_graph=True)
    torch.manual_seed(1)
    z = enc.run(None, feeds)[0]
    L = z.shape[1]
    wins = get_chunks(L, args.stream_chunk, args.stream_pad)
    sid = feeds["sid"]

    def stream(d):
        out = []
        for i, (a, b) in enumerate(wins):
            o = d.run(None, {"z": z[:, a:b], "sid": sid})[0].reshape(1, -1)
            lo, hi = depad_bounds(len(wins), i, args.stream_chunk, args.stream_pad, hop, o.shape[1])
            out.append(o[0, lo:hi])
        return np.concatenate(out)

    a0, a1 = stream(dec), stream(decg)  # warm-up (captures the graphs) + equality
    res = {"model": args.model, "sampling_rate": sr, "phonemes": args.stream_phonemes, "frames": int(L),
           "audio_s": L * hop / float(sr),
           "windows": len(wins), "chunk": args.stream_chunk, "pad": args.stream_pad,
           "graph_equals_plain": bool(np.array_equal(a0, a1)), "samples": int(a0.size)}
    for _ in range(3):
        enc.run(None, feeds)
        full.run(None, feeds)
    w0, wm = wins[0], wins[min(1, len(wins) - 1)]
    res["encoder_ms"] = med(lambda: enc.run(None, feeds), args.stream_reps)
    for name, d in (("plain", dec), ("graph", decg)):
        res[f"first_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, w0[0]:w0[1]], "sid": sid}), args.stream_reps)
        res[f"middle_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, wm[0]:wm[1]], "sid": sid}), args.stream_reps)
        res[f"stream_total_ms_{name}"] = med(lambda: stream(d), max(5, args.stream_reps // 3))
        res[f"first_chunk_latency_ms_{name}"] = res["encoder_ms"] + res[f"first_window_ms_{name}"]
    res["non_stream_ms"] = med(lambda: full.run(None, feeds), args.stream_reps)
    res["rtf_stream_graph"] = (res["encoder_ms"] + res["stream_total_ms_graph"]) / 1e3 / res["audio_s"]
    if args.stream_cpu:  # the oracle (CPU port of the reference) on the same two stages
        from oracle import vits_oracle as vo  # cpu_baseline leg only -- never on the product path
        from tests import util
        W = {k: v.float() for k, v in checkpoint.fold_weight_norm(sd).items()}
        cd = util.cfg_dict(cfg)
        zt = torch.from_numpy(z).transpose(1, 2).contiguous()
        g = W["emb_g.weight"][0:1].unsqueeze(-1)
        for thr in (1, min(16, os.cpu_count())):  # more threads only oversubscribe this tiny conv
            torch.set_num_threads(thr)
            with torch.no_grad():
                vo.decoder(W, cd, zt[:, :, w0[0]:w0[1]], g)
                t0 = time.perf_counter()
                for _ in range(3):
                    vo.decoder(W, cd, zt[:, :, w0[0]:w0[1]], g)
                res.setdefault("cpu_baseline", {"kind": "port", "sample": "first decoder window, mean of 3"})[
                    f"first_window_ms_{thr}thr"] = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps(res), flush=True)


# BASELINE.json `configs`, by index: what each names, as flags of this script.  configs[0] is the
# reference's own CPU-runnable plumbing case (no GPU line); configs[1] is the headline.
PRESETS = {
    # configs[1]: Baker v1, B = 16 x 128 phonemes, fp32, 22.05 kHz, single speaker
    "baker": dict(model="v1", batch=16, phonemes=128, ragged=False, n_speakers=1, sr=22050,
                  decoder_dtype="f32", flow_dtype="f32", tag="BASELINE.json configs[1]"),
    # configs[2]: multilingual v3, B = 64, bf16, two speakers (baker + ljspeech, multilingual/run.sh:23-27)
    "multilingual": dict(model="v3", batch=64, phonemes=128, ragged=False, n_speakers=2, sr=16000,
                         decoder_dtype="bf16", flow_dtype="bf16", tag="BASELINE.json configs[2]"),
    # configs[3]: AISHELL-3 v1 (examples/aishell-3/configs/v1.json: baker v1 at sampling_rate 44100),
    # 218-row speaker table, 64 ragged utterances per GPU (512 over 8 GPUs)
    "aishell3": dict(model="v1", batch=64, phonemes=128, ragged=True, n_speakers=218, sr=44100,
                     decoder_dtype="f32", flow_dtype="f32", tag="BASELINE.json configs[3]"),
    # configs[4]: builder-defined 48 kHz stress shape (no such reference recipe), fp16
    "stress48k": dict(model="stress48k", batch=16, phonemes=128, ragged=False, n_speakers=1,
                      sr=48000, decoder_dtype="f16", flow_dtype="f16", tag="BASELINE.json configs[4]"),
}


def cpu_baseline(cfg, sd, x, lens, sid, n_utts, sr, hop):
    """Times the oracle (CPU port of the reference path; /root/reference does not exist on the GPU
    box, so `kind` is "port": same ATen CPU kernels, same module order) on `n_utts` utterances of
    the same workload, at 1 / 8 / 16 / 32 threads; `value` is the best, with its thread count.
    The single-thread figure is the reference's own setting (inference.py:49-50)."""
    from oracle import vits_oracle as vo  # checker / baseline only -- never on the product path
    from tests import util
    from wetts_amd import checkpoint
    W = checkpoint.fold_weight_norm(sd)
    cd = util.cfg_dict(cfg)
    xs, ls, ss = x[:n_utts], lens[:n_utts], sid[:n_utts]
    saved = torch.get_num_threads()
    host = os.cpu_count() or 1
    sweep = []
    try:
        torch.set_num_threads(min(8, host))
        vo.infer(W, cd, xs[:1, :16], torch.tensor([16]), ss[:1], 0.667, 1.0, 0.8)  # warm-up (tiny)
        for thr in [t for t in (1, 8, 16, 32) if t <= host]:
            torch.set_num_threads(thr)
            torch.manual_seed(1)
            # the single-thread point is the slowest: bound it to one utterance
            n = 1 if thr == 1 else n_utts
            tm = {}
            t0 = time.perf_counter()
            o, _, y_mask, _ = vo.infer(W, cd, xs[:n], ls[:n], ss[:n], noise_scale=0.667,
                                       length_scale=1.0, noise_scale_w=0.8, timers=tm)
            dt = time.perf_counter() - t0
            samples = float(y_mask.sum().item()) * hop
            sweep.append({"threads": thr, "samples_per_s": samples / dt, "seconds": dt,
                          "utterances": n, "rtf": dt / (samples / sr),
                          "stage_s": {k: round(v, 4) for k, v in tm.items()}})
    finally:
        torch.set_num_threads(saved)
    best = max(sweep, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"],
            "kind": "port",
            "kind_reason": "the GPU box has no /root/reference; oracle/vits_oracle.py restates it on "
                           "the same ATen CPU kernels and is pinned to it by tests/golden",
            "host_cores": host, "rtf": best["rtf"], "stage_s": best["stage_s"],
            "thread_sweep": sweep,
            "sample": f"{best['utterances']} of the batch's utterances x {int(ls[0])} phonemes, oracle "
                      f"infer() once per thread count ({sum(r['seconds'] for r in sweep):.0f} s total)"}


def spawn_ranks(args):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks here (one process
    per GPU, RCCL rendezvous on 127.0.0.1).  Refuses -- exit code 3, nothing on stdout -- when the
    node has fewer than N GPUs (WETTS_BENCH_SINGLE_DEVICE=1: dry run, all ranks share GPU 0)."""
    from wetts_amd import sharding
    return sharding.launch_ranks(args.gpus, __file__, sys.argv[1:],
                                 require_gpus=not os.environ.get("WETTS_BENCH_SINGLE_DEVICE"))


def main():
    args = parse_args()
    if args.stream:
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        return stream_bench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    from wetts_amd import SynthesizerTrn, _lib, checkpoint, config, sharding, synth

    rank, local_rank, world = sharding.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    single_dev = bool(os.environ.get("WETTS_BENCH_SINGLE_DEVICE"))  # dry run: all ranks on GPU 0
    if single_dev:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no HIP device "
                         f"({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)  # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local_rank)
    sharding.init_process_group()
    lib = _lib.load()

    pre = PRESETS[args.config]
    mname = args.model or pre["model"]
    batch = args.batch or pre["batch"]
    phonemes = args.phonemes or pre["phonemes"]
    ragged = args.ragged or pre["ragged"]
    ddtype = args.decoder_dtype or pre["decoder_dtype"]
    fdtype = args.flow_dtype or (pre["flow_dtype"] if not args.decoder_dtype else "f32")
    n_speakers = args.speakers or pre["n_speakers"]
    sr = pre["sr"] if not args.model else config.SAMPLING_RATES[mname]
    n_vocab = 256  # synthetic phone table
    net = SynthesizerTrn(n_vocab, 513, 32, n_speakers=n_speakers, **config.MODEL_CONFIGS[mname])
    cfg = net.cfg
    hop = net.hop_length

    # ---- weights: rank 0 builds the blob, one broadcast over RCCL, every rank repacks locally
    numel = checkpoint.blob_numel(cfg)
    sd = None
    if rank == 0:
        sd = synth.make_state_dict(cfg, seed=0)
        blob = checkpoint.pack_blob(cfg, sd).to(dev)
    else:
        blob = torch.empty(numel, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_b0 = time.perf_counter()
    sharding.broadcast_blob(blob, src=0)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t_b0) * 1e3
    net.load_blob(blob)
    if ddtype != "f32":
        net.set_decoder_dtype(ddtype)
    if fdtype != "f32":
        net.set_flow_dtype(fdtype)

    # ---- inputs: global utterance list, LPT-dealt to ranks, resident on the device
    total = batch * world
    x, lens, sid = make_inputs(mname, n_vocab, n_speakers, total, phonemes, ragged)
    shards = sharding.shard_utterances(lens.tolist(), world)
    mine = torch.tensor(shards[rank], dtype=torch.long)
    xh, lh, sh = x[mine].contiguous(), lens[mine].contiguous(), sid[mine].contiguous()
    # the shard is length-sorted (sharding.shard_utterances), so consecutive slices pad little: a ragged
    # shard is decoded as `nb` padded sub-batches, each cut to its own longest utterance
    nb = args.buckets or (4 if ragged else 1)
    nb = max(1, min(nb, len(mine)))
    per = -(-len(mine) // nb)
    host_buckets = []
    for i in range(0, len(mine), per):
        tx = int(lh[i:i + per].max())
        host_buckets.append((xh[i:i + per, :tx].contiguous(), lh[i:i + per].contiguous(),
                             sh[i:i + per].contiguous()))
    dev_buckets = [tuple(t.to(dev) for t in hb) for hb in host_buckets]
    torch.manual_seed(1 + rank)  # the library's Philox stream follows torch.initial_seed()

    def step(buckets=dev_buckets):
        # eps_w / eps_z = None: both standard-normal draws come from the library's Philox kernel
        outs, masks_ = [], []
        for (xb_, lb_, sb_) in buckets:
            o, attn, y_mask, _ = net.infer(xb_, lb_, sid=sb_, noise_scale=0.667, length_scale=1.0,
                                           noise_scale_w=0.8)
            outs.append(o)
            masks_.append(y_mask)
        return outs, masks_

    # untimed set-up, before the W warm-up steps of the contract: a fresh box starts with the GPU
    # in a low power state and with lazy one-time initialisation pending (code objects, the MRF
    # event pool); run the step for ~2.5 s so that neither lands inside the timed region
    _lib.check(lib.wetts_set_mrf_timing(net._handle, 1), "set_mrf_timing")
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 2.5:
        step()
        torch.cuda.synchronize()
    ms0, nl0, nc0 = C.c_double(), C.c_int64(), C.c_int32()
    _lib.check(lib.wetts_read_mrf_timing(net._handle, C.byref(ms0), C.byref(nl0), C.byref(nc0)),
               "read_mrf_timing")  # drains the set-up events
    _lib.check(lib.wetts_set_mrf_timing(net._handle, 0), "set_mrf_timing")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    _lib.check(lib.wetts_set_mrf_timing(net._handle, 1), "set_mrf_timing")
    frames = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    masks = []
    for _ in range(args.steps):
        o, y_masks = step()
        masks.extend(y_masks)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for ym in masks:
        frames += float(ym.sum().item())
    ms, nl, nc = C.c_double(), C.c_int64(), C.c_int32()
    _lib.check(lib.wetts_read_mrf_timing(net._handle, C.byref(ms), C.byref(nl), C.byref(nc)),
               "read_mrf_timing")
    _lib.check(lib.wetts_set_mrf_timing(net._handle, 0), "set_mrf_timing")
    padded_frames = float(sum(ym.numel() for ym in masks))  # B*Ty: what the decoder computes

    # PCIe-inclusive variant (SURVEY 8d's wall: H2D of the ids, D2H of the audio; the driver contract
    # says inputs are resident when the timed region starts, so this is reported beside `value`,
    # never as it): a few extra steps with pinned host buffers on both sides
    pinned = [tuple(t.pin_memory() for t in hb) for hb in host_buckets]
    n_pcie = max(1, min(3, args.steps))

    def pcie_pass(n, pipelined):
        """ids H2D -> infer() -> audio D2H for n steps.  Serial: every step waits for its own copy-out
        (a blocking .cpu()).  Pipelined: the audio goes to a pinned double buffer on a copy stream while the
        next step computes -- what a server that streams results does."""
        copy_stream = torch.cuda.Stream(device=dev)
        bufs, done = [None, None], [None, None]
        frames_, k = 0.0, 0
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            for hb in pinned:
                xd2, ld2, sd2 = (t.to(dev, non_blocking=True) for t in hb)
                o, _, y_mask, _ = net.infer(xd2, ld2, sid=sd2, noise_scale=0.667, length_scale=1.0,
                                            noise_scale_w=0.8)
                if not pipelined:
                    _ = o.cpu()
                else:
                    slot = k & 1
                    if done[slot] is not None:
                        done[slot].synchronize()  # the host has consumed this buffer's previous contents
                    if bufs[slot] is None or bufs[slot].numel() < o.numel():
                        bufs[slot] = torch.empty(int(o.numel() * 1.25), dtype=o.dtype).pin_memory()
                    ready = torch.cuda.Event()
                    ready.record()
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(ready)
                        bufs[slot][:o.numel()].copy_(o.reshape(-1), non_blocking=True)
                        o.record_stream(copy_stream)
                        done[slot] = torch.cuda.Event()
                        done[slot].record()
                    k += 1
                frames_ += float(net._last["y_lengths_host"].sum().item())
        torch.cuda.synchronize()
        return frames_ * hop / (time.perf_counter() - t_)

    pcie_rate = pcie_pass(n_pcie, False)
    n_pipe = max(n_pcie, min(8, args.steps))
    pcie_pass(1, True)  # allocates the pinned buffers outside the timed pass
    pcie_pipe_rate = pcie_pass(n_pipe, True)

    # ---- reduce over ranks: time = max, work = sum
    stat = torch.tensor([elapsed, frames, padded_frames, ms.value, float(nl.value), pcie_rate, pcie_pipe_rate],
                        dtype=torch.float64, device=dev)
    if world > 1:
        mx = stat.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stat.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, frames, pcie_rate, pcie_pipe_rate = float(mx[0]), float(sm[1]), float(sm[5]), float(sm[6])
    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    samples = frames * hop
    value = samples / elapsed
    fl, by, mfl, mby = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    lib.wetts_hifigan_cost(C.byref(cfg), C.byref(fl), C.byref(by), C.byref(mfl), C.byref(mby))
    # dominant kernel: algorithmic FLOPs of the MRF convs over the frames rank 0 decoded
    # (padded frames: the decoder has no masks, decoders.py:63-82), / live device time
    mrf_flops = mfl.value * padded_frames
    mrf_tflops = mrf_flops / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    mrf_gbs = mby.value * padded_frames / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0
    traffic, traffic_src = None, None
    try:  # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this same command
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
        if files and args.config == "baker" and not (args.model or args.batch or args.phonemes):
            dom = json.load(open(files[-1]))["dominant_conv_mfma"]
            traffic, traffic_src = dom["hbm_bytes_per_launch"], os.path.basename(files[-1])
    except Exception:
        pass
    if ddtype != "f32":
        # 16-bit activations: per-conv algorithmic bytes (SURVEY 8d accounting: every conv reads its
        # input and writes its output once, each residual add reads x once more) are half the f32
        # figure.  The fused ResBlock pair kernel moves fewer bytes than that through HBM (the
        # intermediate stays in LDS), which is how `achieved` can approach the roofline; k=3 pairs
        # are HBM/latency-bound, k=11 pairs MFMA-bound -- both views are reported.
        gbs = 0.5 * mrf_gbs
        roofline = {
            "kernel": "resblock_pair16_kernel + conv_bf16_kernel (MRF ResBlock convs, "
                      f"{ddtype} channel-last; C<=128 pairs fused)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "launches": int(nl.value), "avg_launch_ms": ms.value / max(1, nl.value),
            "bytes_per_launch": 0.5 * mby.value * padded_frames / max(1, nl.value),
            "mfma_view": {"achieved": mrf_tflops, "peak": 2500.0, "unit": "TFLOP/s",
                          "frac": mrf_tflops / 2500.0},
            "mrf_share_of_step": ms.value / (elapsed * 1e3),
        }
    else:
        roofline = {
            "kernel": ("conv_mfma_kernel (Vocos ConvNeXt pointwise GEMMs 512<->1536)"
                       if "vocos" in mname else
                       "the MRF ResBlock conv class: conv_mfma_kernel<.., MRF=true, FAST=true> (single "
                       "convs) + resblock_chain32_kernel (whole ResBlock1 at C=32 k<=7 / C=64 k=3, single "
                       "pairs at C=32 k=11 / C=128 k=3)"),
            "bound": "mfma", "achieved": mrf_tflops, "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": mrf_tflops / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per MRF launch (2*FETCH_SIZE+WRITE_SIZE)*1024, PMC",
            "traffic_source": traffic_src,
            "launches": int(nl.value), "avg_launch_ms": ms.value / max(1, nl.value),
            "flops_per_launch": mrf_flops / max(1, nl.value),
            "hbm_view": {"achieved": mrf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mrf_gbs / HBM_PEAK_GBS,
                         "note": "per-conv algorithmic bytes (SURVEY 8d); fp32 convs are "
                                 "compute-bound (AI 113 flop/B > ridge ~20)"},
            "mrf_share_of_step": ms.value / (elapsed * 1e3),
        }
    backend = dist.get_backend() if world > 1 else "none"
    observed_world = dist.get_world_size() if world > 1 else 1
    prec = ("fp32" if ddtype == "f32" else
            "uint8 dynamic-quantised decoder convs (int32 accumulate)" if ddtype == "uint8" else
            ddtype + " decoder") + \
        ("" if fdtype == "f32" else f" + {fdtype} flow WaveNet layers")
    out = {
        "metric": "audio samples/sec + RTF @22.05 kHz, VITS-Baker, 1/2/4/8 MI355X",
        "value": value, "unit": "samples/s", "n_gpus": observed_world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if ddtype == "f32" and fdtype == "f32" else
                 (prec if ddtype == "uint8" else f"{prec} (f32 accumulate)"),
        "data": "synthetic (seeded phoneme ids, seeded random-init weights, ~6.5 frames/phoneme, "
                "Philox noise drawn on the device)",
        "rtf": elapsed / (samples / sr), "x_realtime": (samples / sr) / elapsed,
        "config": {"workload": f"{args.config}_{mname} infer(): B={batch}/GPU x "
                               f"{phonemes} phonemes{' ragged U{32..' + str(phonemes) + '}' if ragged else ''}, "
                               f"{prec}, {n_speakers} speaker(s), {sr} Hz ({pre['tag']})",
                   "global_batch": total, "phonemes": phonemes, "hop": hop,
                   "padded_sub_batches_per_step": nb,
                   "n_speakers": n_speakers, "sampling_rate": sr,
                   "valid_frames_per_step": frames / args.steps,
                   "parallelism": f"utterance-shard x{observed_world} (backend {backend}"
                                  f"{', all ranks on one device: dry run' if single_dev else ''}), "
                                  f"weights broadcast once ({numel * 4 / 1e6:.0f} MB, {bcast_ms:.1f} ms),"
                                  " no collectives in the decode loop"},
        "pcie_inclusive_samples_per_s": pcie_rate,
        "pcie_inclusive_pipelined_samples_per_s": pcie_pipe_rate,
        "pcie_inclusive_note": "ids H2D + infer() + audio D2H per step (SURVEY 8(d)'s wall), reported beside `value`, "
                               f"never as it.  Serial: blocking copy-out each step ({n_pcie} steps); pipelined: pinned "
                               f"double buffer, copy-out on its own stream under the next step ({n_pipe} steps)",
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, sd, x, lens, sid, min(args.cpu_sample, total), sr,
                                           hop)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
