#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X VITS hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one SynthesizerTrn.infer() over one synthetic batch per rank: Baker VITS (v1 config,
22.05 kHz, SDP, ResBlock1, C0=512), batch = 16 utterances x 128 phonemes, fp32
(BASELINE.json configs[1]).  Inputs are resident in HBM when the timed region starts; outputs
stay in HBM.  One process per GPU; weights are broadcast once from rank 0 (RCCL), then every rank
decodes its own utterance shard with no collective in the loop (weak scaling).

Prints ONE JSON line on rank 0 (see the driver contract).  Extra keys:
  roofline     dominant kernel = the MRF ResBlock conv stack (conv_mfma_kernel), timed live with
               HIP events recorded on the launch stream inside the timed steps
  cpu_baseline the oracle (oracle/vits_oracle.py, a port of the reference running the same ATen
               CPU kernels) timed on this box's host cores on a bounded sample of the workload
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
    ap.add_argument("--model", default="v1")
    ap.add_argument("--batch", type=int, default=16, help="utterances per rank per step")
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--ragged", action="store_true", help="Tx ~ U{32..phonemes} (configs[3] style)")
    ap.add_argument("--decoder-dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="HiFi-GAN arithmetic; the headline metric is quoted at f32")
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


def cpu_baseline(cfg, sd, x, lens, sid, n_utts, sr, hop):
    """Times the oracle (CPU port of the reference path, same ATen kernels) on `n_utts` utterances
    of the same workload.  Returns the cpu_baseline object."""
    from oracle import vits_oracle as vo  # checker / baseline only -- never on the product path
    from tests import util
    from wetts_amd import checkpoint
    W = checkpoint.fold_weight_norm(sd)
    cd = util.cfg_dict(cfg)
    xs, ls, ss = x[:n_utts], lens[:n_utts], sid[:n_utts]
    torch.manual_seed(1)
    cores = torch.get_num_threads()
    vo.infer(W, cd, xs[:1, :16], torch.tensor([16]), ss[:1], 0.667, 1.0, 0.8)  # warm-up (tiny)
    t0 = time.perf_counter()
    o, _, y_mask, _ = vo.infer(W, cd, xs, ls, ss, noise_scale=0.667, length_scale=1.0,
                               noise_scale_w=0.8)
    dt = time.perf_counter() - t0
    samples = float(y_mask.sum().item()) * hop
    one = None
    try:  # the reference's own setting is ONE thread (inference.py:49-50); time 1 utterance
        torch.set_num_threads(1)
        t1 = time.perf_counter()
        o1, _, ym1, _ = vo.infer(W, cd, xs[:1, :64], torch.tensor([64]), ss[:1], noise_scale=0.667,
                                 length_scale=1.0, noise_scale_w=0.8)
        d1 = time.perf_counter() - t1
        one = {"value": float(ym1.sum().item()) * hop / d1, "cores": 1,
               "sample": f"1 utterance x 64 phonemes, {d1:.1f} s"}
    finally:
        torch.set_num_threads(cores)
    return {"value": samples / dt, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "single_thread": one,
            "sample": f"{n_utts} of the batch's utterances ({int(ls.sum())} phonemes, "
                      f"{int(y_mask.sum().item())} frames), oracle infer() once, {dt:.1f} s",
            "rtf": dt / (samples / sr)}


def main():
    args = parse_args()
    if args.stream:
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        return stream_bench(args)
    from wetts_amd import SynthesizerTrn, _lib, checkpoint, config, sharding, synth

    rank, local_rank, world = sharding.init_process_group()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    if os.environ.get("WETTS_BENCH_SINGLE_DEVICE"):  # dry-run of the N>1 control flow on one GPU
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _lib.load()

    n_vocab, n_speakers = 256, 1  # SURVEY §8(d) cfg 2: synthetic phone table, Baker single speaker
    model = config.MODEL_CONFIGS[args.model]
    sr = config.SAMPLING_RATES[args.model]
    net = SynthesizerTrn(n_vocab, 513, 32, n_speakers=n_speakers, **model)
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
    t_b0 = time.perf_counter()
    sharding.broadcast_blob(blob, src=0)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t_b0) * 1e3
    net.load_blob(blob)
    if args.decoder_dtype != "f32":
        net.set_decoder_dtype(args.decoder_dtype)

    # ---- inputs: global utterance list, LPT-dealt to ranks, resident on the device
    total = args.batch * world
    x, lens, sid = make_inputs(args.model, n_vocab, n_speakers, total, args.phonemes, args.ragged)
    shards = sharding.shard_utterances(lens.tolist(), world)
    mine = torch.tensor(shards[rank], dtype=torch.long)
    xd, ld, sd_ids = x[mine].to(dev), lens[mine].to(dev), sid[mine].to(dev)

    def step():
        o, attn, y_mask, _ = net.infer(xd, ld, sid=sd_ids, noise_scale=0.667, length_scale=1.0,
                                       noise_scale_w=0.8)
        return o, y_mask

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
        o, y_mask = step()
        masks.append(y_mask)
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

    # PCIe-inclusive variant (never `value`): one extra step with D2H of the audio
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    o, y_mask = step()
    _ = o.cpu()
    pcie_s = time.perf_counter() - t1
    pcie_rate = float(y_mask.sum().item()) * hop / pcie_s

    # ---- reduce over ranks: time = max, work = sum
    stat = torch.tensor([elapsed, frames, padded_frames, ms.value, float(nl.value)],
                        dtype=torch.float64, device=dev)
    if world > 1:
        mx = stat.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stat.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, frames = float(mx[0]), float(sm[1])
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
        if files and args.model == "v1" and args.batch == 16 and args.phonemes == 128:
            dom = json.load(open(files[-1]))["dominant_conv_mfma"]
            traffic, traffic_src = dom["hbm_bytes_per_launch"], os.path.basename(files[-1])
    except Exception:
        pass
    if args.decoder_dtype != "f32":
        # 16-bit activations: per-conv algorithmic bytes (SURVEY 8d accounting: every conv reads its
        # input and writes its output once, each residual add reads x once more) are half the f32
        # figure.  The fused ResBlock pair kernel moves fewer bytes than that through HBM (the
        # intermediate stays in LDS), which is how `achieved` can approach the roofline; k=3 pairs
        # are HBM/latency-bound, k=11 pairs MFMA-bound -- both views are reported.
        gbs = 0.5 * mrf_gbs
        roofline = {
            "kernel": "resblock_pair16_kernel + conv_bf16_kernel (MRF ResBlock convs, "
                      f"{args.decoder_dtype} channel-last; C<=128 pairs fused)",
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
                       if "vocos" in args.model else
                       "conv_mfma_kernel + resblock_pair32_kernel (MRF ResBlock convs; the C=32 "
                       "stage and the k=3 pairs of C=64/128 run as fused pairs)"),
            "bound": "mfma", "achieved": mrf_tflops, "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": mrf_tflops / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per conv_mfma launch (2*FETCH_SIZE+WRITE_SIZE)*1024, PMC",
            "traffic_source": traffic_src,
            "launches": int(nl.value), "avg_launch_ms": ms.value / max(1, nl.value),
            "flops_per_launch": mrf_flops / max(1, nl.value),
            "hbm_view": {"achieved": mrf_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mrf_gbs / HBM_PEAK_GBS,
                         "note": "per-conv algorithmic bytes (SURVEY 8d); fp32 convs are "
                                 "compute-bound (AI 113 flop/B > ridge ~20)"},
            "mrf_share_of_step": ms.value / (elapsed * 1e3),
        }
    out = {
        "metric": "audio samples/sec + RTF @22.05 kHz, VITS-Baker, 1/2/4/8 MI355X",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.decoder_dtype == "f32" else
                 f"{args.decoder_dtype} decoder (f32 accumulate), f32 encoder/flow",
        "data": "synthetic (seeded phoneme ids, seeded random-init weights, ~6.5 frames/phoneme)",
        "rtf": elapsed / (samples / sr), "x_realtime": (samples / sr) / elapsed,
        "config": {"workload": f"baker_{args.model} infer(): B={args.batch}/GPU x "
                               f"{args.phonemes} phonemes{' ragged' if args.ragged else ''}, "
                               f"{'fp32' if args.decoder_dtype == 'f32' else args.decoder_dtype + ' decoder'}, "
                               f"{sr} Hz (BASELINE.json configs[1])",
                   "global_batch": total, "phonemes": args.phonemes, "hop": hop,
                   "valid_frames_per_step": frames / args.steps,
                   "parallelism": f"utterance-shard x{world}, weights broadcast once "
                                  f"({numel * 4 / 1e6:.0f} MB, {bcast_ms:.1f} ms), no collectives "
                                  "in the decode loop"},
        "pcie_inclusive_samples_per_s": pcie_rate,
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
