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
                    help="0 = sub-batches chosen by wetts_amd.batching.plan (SURVEY 8e: each rank buckets its "
                         "length-sorted shard); N > 0 = N equal-count buckets (round-2 behaviour, for A/B)")
    ap.add_argument("--max-pad-frac", type=float, default=0.08, help="padding share batching.plan may spend")
    ap.add_argument("--decode", default=None, choices=["padded", "ragged"],
                    help="padded = the reference's batched call (every row decoded to the longest of its sub-batch); "
                         "ragged = SynthesizerTrn.infer(ragged=True): every utterance decoded over its own frames, "
                         "the audio the reference's one-utterance-per-call CLI produces (default for ragged presets)")
    ap.add_argument("--max-batch", type=int, default=0, help="largest padded sub-batch (0 = no cap)")
    ap.add_argument("--length-scale", type=float, default=0.92,
                    help="calibrated so the synthetic duration heads give 6.0 +- 0.5 frames/phoneme (SURVEY 8d)")
    ap.add_argument("--presteps-s", type=float, default=2.5, help="untimed set-up before the warm-up steps")
    ap.add_argument("--cpu-baseline-multi", action="store_true",
                    help="also time the CPU baseline on rank 0 when N > 1 (the contract asks for it at N = 1 only)")
    ap.add_argument("--mas", action="store_true", help="print the monotonic-alignment-search JSON instead")
    ap.add_argument("--decoder-dtype", default=None, choices=["f32", "bf16", "f16", "uint8"],
                    help="HiFi-GAN arithmetic; the headline metric is quoted at f32 (uint8 = the "
                         "export_onnx.py --quant dynamic-quantisation variant)")
    ap.add_argument("--flow-dtype", default=None, choices=["f32", "bf16", "f16"],
                    help="arithmetic of the flow's WaveNet layers (wetts_set_flow_precision)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1: SynthesizerTrn.set_overlap(True) -- the encoder stages of step k + 1 run on a side stream beside "
                         "the decoder of step k (back-to-back calls as a two-stage pipeline); 0: strictly one stage at a time")
    ap.add_argument("--decoder-serial", action="store_true",
                    help="f32 decoder: the three ResBlock chains of a stage on ONE stream (WETTS_DECODER_SERIAL; the "
                         "form whose kernel trace has non-overlapping per-kernel durations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "0", "1"],
                    help="refresh roofline.traffic by running the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of "
                         "this same command as child processes after the timed region (+ ~1 min).  auto: on for a "
                         "single-GPU run on a real device that is not itself being profiled; the committed "
                         "profiles/rNN_hbm_traffic.json figure is the fallback and stays in the line beside it")
    ap.add_argument("--cpu-fixture", type=int, default=16,
                    help="also time the oracle on the first N utterances of the batch as ONE padded call (SURVEY 8d's "
                         "fixture) at the sweep's best thread count; 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=1,
                    help="utterances in the CPU sample (1 = the reference CLI's own call shape, inference.py:83-110)")
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
    encg = EncoderSession(net, use_graph=True)
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
    # graphed encoder call: the frame count is sampled, so warm every frame bucket the repetitions will visit
    for _ in range(12):
        encg.run(None, feeds)
    torch.manual_seed(2)
    zp = enc.run(None, feeds)[0]
    torch.manual_seed(2)
    zg = encg.run(None, feeds)[0]
    res["encoder_graph_vs_plain_rel_rms"] = float(np.sqrt(((zg - zp) ** 2).mean()) / max(1e-30, np.sqrt((zp ** 2).mean()))) \
        if zg.shape == zp.shape else None
    w0, wm = wins[0], wins[min(1, len(wins) - 1)]
    res["encoder_ms"] = med(lambda: enc.run(None, feeds), args.stream_reps)
    res["encoder_ms_graph"] = med(lambda: encg.run(None, feeds), args.stream_reps)
    res["encoder_graph_entries"] = [len(encg._graphed._pre), sum(len(e["post"]) for e in encg._graphed._pre.values())]
    for name, d in (("plain", dec), ("graph", decg)):
        res[f"first_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, w0[0]:w0[1]], "sid": sid}), args.stream_reps)
        res[f"middle_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, wm[0]:wm[1]], "sid": sid}), args.stream_reps)
        res[f"stream_total_ms_{name}"] = med(lambda: stream(d), max(5, args.stream_reps // 3))
        res[f"first_chunk_latency_ms_{name}"] = res["encoder_ms" if name == "plain" else "encoder_ms_graph"] + \
            res[f"first_window_ms_{name}"]
    res["non_stream_ms"] = med(lambda: full.run(None, feeds), args.stream_reps)
    res["rtf_stream_graph"] = (res["encoder_ms_graph"] + res["stream_total_ms_graph"]) / 1e3 / res["audio_s"]
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
    print(json.dumps(res), file=json_out(), flush=True)


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
                     decoder_dtype="f32", flow_dtype="f32", decode="ragged", tag="BASELINE.json configs[3]"),
    # configs[4]: builder-defined 48 kHz stress shape (no such reference recipe), fp16
    "stress48k": dict(model="stress48k", batch=16, phonemes=128, ragged=False, n_speakers=1,
                      sr=48000, decoder_dtype="f16", flow_dtype="f16", tag="BASELINE.json configs[4]"),
}


def cpu_baseline(cfg, sd, x, lens, sid, n_utts, sr, hop, length_scale=1.0, n_fixture=0):
    """Times the oracle (CPU port of the reference path; /root/reference does not exist on the GPU
    box, so `kind` is "port": same ATen CPU kernels, same module order) on ONE bounded sample of the
    workload -- the first `n_utts` utterances, the same sample at every thread count -- at 1 / 8 / 16 / 32
    threads: one warm-up run, then the median of three.  `value` is the best thread count;
    `threads_1` is the reference's own setting (inference.py:49-50: torch.set_num_threads(1))."""
    import statistics
    from oracle import vits_oracle as vo  # checker / baseline only -- never on the product path
    from tests import util
    from wetts_amd import checkpoint
    W = checkpoint.fold_weight_norm(sd)
    cd = util.cfg_dict(cfg)
    xs, ls, ss = x[:n_utts], lens[:n_utts], sid[:n_utts]
    saved = torch.get_num_threads()
    host = os.cpu_count() or 1
    sweep = []
    t_all = time.perf_counter()
    try:
        for thr in [t for t in (1, 8, 16, 32) if t <= host]:
            torch.set_num_threads(thr)
            runs, tm = [], {}
            for rep in range(4):  # rep 0 = warm-up
                torch.manual_seed(1)
                tm = {}
                t0 = time.perf_counter()
                o, _, y_mask, _ = vo.infer(W, cd, xs, ls, ss, noise_scale=0.667, length_scale=length_scale,
                                           noise_scale_w=0.8, timers=tm)
                dt = time.perf_counter() - t0
                if rep:
                    runs.append(dt)
            dt = statistics.median(runs)
            samples = float(y_mask.sum().item()) * hop
            sweep.append({"threads": thr, "samples_per_s": samples / dt, "seconds_median_of_3": dt,
                          "seconds_runs": [round(r, 4) for r in runs], "utterances": n_utts,
                          "rtf": dt / (samples / sr), "stage_s_last_run": {k: round(v, 4) for k, v in tm.items()}})
    finally:
        torch.set_num_threads(saved)
    best = max(sweep, key=lambda r: r["samples_per_s"])
    # SURVEY 8(d)'s fixture itself (the benched batch, up to 16 utterances) at the best thread count of the sweep
    fixture = None
    if n_fixture > n_utts:
        try:
            torch.set_num_threads(best["threads"])
            xf, lf, sf = x[:n_fixture], lens[:n_fixture], sid[:n_fixture]
            for rep in range(1):  # (the sweep above has warmed ATen; a B = 16 run is 10-25 s of host time)
                torch.manual_seed(1)
                t0 = time.perf_counter()
                o, _, y_mask, _ = vo.infer(W, cd, xf, lf, sf, noise_scale=0.667, length_scale=length_scale,
                                           noise_scale_w=0.8)
                dtf = time.perf_counter() - t0
            sf_ = float(y_mask.sum().item()) * hop
            fixture = {"utterances": n_fixture, "threads": best["threads"], "samples_per_s": sf_ / dtf,
                       "seconds": dtf, "rtf": dtf / (sf_ / sr), "method": "one timed run of the padded batch, after the sweep"}
        finally:
            torch.set_num_threads(saved)
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "batch_fixture": fixture,
            "kind": "port",
            "kind_reason": "the GPU box has no /root/reference; oracle/vits_oracle.py restates it on "
                           "the same ATen CPU kernels and is pinned to it by tests/golden",
            "host_cores": host, "rtf": best["rtf"], "stage_s": best["stage_s_last_run"],
            "threads_1": {"samples_per_s": sweep[0]["samples_per_s"], "rtf": sweep[0]["rtf"],
                          "note": "the reference's own setting (inference.py:49-50)"},
            "best": {"threads": best["threads"], "samples_per_s": best["samples_per_s"]},
            "method": "same sample at every thread count; 1 warm-up + median of 3 (SURVEY 8d)",
            "thread_sweep": sweep,
            "sample": f"{n_utts} utterance(s) of the batch x {int(ls[0])} phonemes, oracle infer() 4x per "
                      f"thread count ({time.perf_counter() - t_all:.0f} s total)"}


def mas_bench(args):
    """`bench.py --mas`: the monotonic-alignment search kernel (monotonic_align.py:22-57) on
    training-shaped score tensors, device time from HIP events on the launch stream (median), against
    oracle/mas_oracle.c (the reference's numba loop restated in C, single thread like the reference's
    serial loop over the batch) on the host.  One JSON line."""
    import ctypes as C
    import statistics
    import subprocess
    import numpy as np
    from wetts_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", odir])
    olib = C.CDLL(os.path.join(odir, "_build", "libmas_oracle.so"))  # cpu_baseline leg only
    olib.mas_oracle.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
    olib.mas_oracle.restype = None
    rows = []
    for (b, ty, tx) in ((16, 800, 128), (64, 1000, 200)):
        g = torch.Generator().manual_seed(7)
        neg = torch.randn(b, ty, tx, generator=g)
        t_y = torch.randint(ty // 2, ty + 1, (b,), generator=g).int()
        t_x = torch.minimum(torch.randint(tx // 3, tx + 1, (b,), generator=g).int(), t_y)
        nd, tyd, txd = neg.to(dev), t_y.to(dev), t_x.to(dev)
        path = torch.empty(b, ty, tx, dtype=torch.int32, device=dev)
        ws = torch.empty(b * ty * tx, dtype=torch.float32, device=dev)

        def run():
            _lib.check(lib.wetts_mas(_lib.ptr(nd), _lib.ptr(tyd), _lib.ptr(txd), b, ty, tx, _lib.ptr(path),
                                     _lib.ptr(ws), ws.numel() * 4, _lib.current_stream_ptr()), "wetts_mas")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.stream_reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = statistics.median(ts)
        # host: the C restatement of the numba kernel, 1 warm-up + median of 3
        cpu = []
        negn = neg.numpy()
        for rep in range(4):
            vals = np.array(negn, dtype=np.float32, copy=True)
            pth = np.zeros(vals.shape, dtype=np.int32)
            t0 = time.perf_counter()
            olib.mas_oracle(pth.ctypes.data, vals.ctypes.data, t_y.numpy().ctypes.data, t_x.numpy().ctypes.data,
                            b, ty, tx)
            if rep:
                cpu.append((time.perf_counter() - t0) * 1e3)
        same = bool(np.array_equal(path.cpu().numpy(), pth))
        cells = float((t_y.double() * t_x.double()).sum())
        byt = 2.0 * b * ty * tx * 4  # scores read once, path written once
        rows.append({"shape": [b, ty, tx], "device_ms": ms, "device_ms_min": min(ts),
                     "cpu_ms_1thread": statistics.median(cpu), "speedup": statistics.median(cpu) / ms,
                     "bit_exact_vs_c_oracle": same, "valid_cells_per_s": cells / (ms * 1e-3),
                     "hbm_gbs": byt / (ms * 1e-3) / 1e9, "hbm_frac": byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "rows_per_us_per_utterance": float(t_y.max()) / (ms * 1e3)})
    print(json.dumps({"metric": "monotonic alignment search (maximum_path), device ms per batch",
                      "kernel": "mas_wave_kernel (one wave per utterance, DPP row step, decision bits in LDS)",
                      "bound": "latency of the serial row recurrence (t_y dependent steps per utterance)",
                      "cases": rows,
                      "cpu_baseline": {"kind": "port", "cores": 1,
                                       "what": "oracle/mas_oracle.c = maximum_path_jit restated in C, serial over "
                                               "the batch like the reference (monotonic_align.py:11-19)"}}),
          file=json_out(), flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks here (one process
    per GPU, RCCL rendezvous on 127.0.0.1).  Refuses -- exit code 3, nothing on stdout -- when the
    node has fewer than N GPUs (WETTS_BENCH_SINGLE_DEVICE=1: dry run, all ranks share GPU 0)."""
    from wetts_amd import sharding
    stub = bool(os.environ.get("WETTS_BENCH_TEST_BACKEND"))
    return sharding.launch_ranks(args.gpus, __file__, sys.argv[1:],
                                 require_gpus=not (os.environ.get("WETTS_BENCH_SINGLE_DEVICE") or stub))


class HipBackend:
    """Everything of the bench that touches the device.  The control flow in main() -- plan, warm-up, timed
    loop, reductions over ranks, the JSON line -- talks to the device only through this object, so that
    tests/test_bench_control_flow.py can run that whole flow at world size 2 on the CPU (gloo) against a stub
    with the same methods (selected by WETTS_BENCH_TEST_BACKEND; its JSON line is labelled, see `label`)."""
    label = None  # a stub puts a marker here; it is copied into the JSON line

    def __init__(self, rank, local_rank):
        from wetts_amd import _lib
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        self.single_dev = bool(os.environ.get("WETTS_BENCH_SINGLE_DEVICE"))  # dry run: all ranks on GPU 0
        if self.single_dev:
            local_rank = 0
        elif torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no HIP device "
                             f"({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)  # before the process group: RCCL binds to the current device
        self.device = torch.device("cuda", local_rank)
        self._lib_mod = _lib
        self.lib = _lib.load()

    def sync(self):
        torch.cuda.synchronize()

    def make_model(self, mname, n_vocab, n_speakers):
        from wetts_amd import SynthesizerTrn, config
        return SynthesizerTrn(n_vocab, 513, 32, n_speakers=n_speakers, **config.MODEL_CONFIGS[mname])

    def set_mrf_timing(self, net, on):
        self._lib_mod.check(self.lib.wetts_set_mrf_timing(net._handle, 1 if on else 0), "set_mrf_timing")

    def read_mrf_timing(self, net):
        ms, nl, nc = C.c_double(), C.c_int64(), C.c_int32()
        self._lib_mod.check(self.lib.wetts_read_mrf_timing(net._handle, C.byref(ms), C.byref(nl), C.byref(nc)),
                            "read_mrf_timing")
        return ms.value, nl.value

    def read_mrf_bytes(self, net):
        by = C.c_double()
        self._lib_mod.check(self.lib.wetts_read_mrf_bytes(net._handle, C.byref(by)), "read_mrf_bytes")
        return by.value

    def hifigan_cost(self, cfg):
        fl, by, mfl, mby = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self.lib.wetts_hifigan_cost(C.byref(cfg), C.byref(fl), C.byref(by), C.byref(mfl), C.byref(mby))
        return mfl.value, mby.value

    def pcie_pass(self, net, pinned, n, pipelined, infer_kw):
        """ids H2D -> infer() -> audio D2H for n steps.  Serial: every step waits for its own copy-out
        (a blocking .cpu()).  Pipelined: the audio goes to a pinned double buffer on a copy stream while the
        next step computes -- what a server that streams results does."""
        dev = self.device
        hop = net.hop_length
        copy_stream = torch.cuda.Stream(device=dev)
        bufs, done = [None, None], [None, None]
        frames_, k = 0.0, 0
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n):
            for hb in pinned:
                # (net.upload: on the stream that reads the ids -- the encoder's side stream when calls are pipelined)
                xd2, ld2, sd2 = (net.upload(t, non_blocking=True) if hasattr(net, "upload") else t.to(dev, non_blocking=True) for t in hb)
                o, _, y_mask, _ = net.infer(xd2, ld2, sid=sd2, **infer_kw)
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

    def pin(self, t):
        return t.pin_memory()


def load_backend(rank, local_rank):
    spec = os.environ.get("WETTS_BENCH_TEST_BACKEND")  # "module:Class" -- CPU control-flow tests only
    if spec:
        import importlib
        mod, cls = spec.split(":")
        return getattr(importlib.import_module(mod), cls)(rank, local_rank)
    return HipBackend(rank, local_rank)


def blob_fingerprint(blob):
    """Order-sensitive checksum of the weight blob, computed where the blob lives: two float64 sums (plain and
    index-weighted) -- any rank whose broadcast copy differs from rank 0's shows up in the min / max over
    ranks."""
    b = blob.detach().to(torch.float64)
    w = torch.arange(1, b.numel() + 1, dtype=torch.float64, device=b.device) % 8191
    return torch.stack([b.sum(), (b * w).sum()])


_JSON_OUT = None


def json_out():
    """The stream the ONE JSON line goes to: the process's original stdout.  File descriptor 1 itself is pointed at
    stderr for the rest of the run, because communication libraries print connection banners from C++ straight to
    fd 1 (Gloo does: "[Gloo] Rank 0 is connected to 1 peer ranks"), which would break the one-line contract."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def main():
    args = parse_args()
    if args.stream or args.mas:
        json_out()
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
        return stream_bench(args) if args.stream else mas_bench(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # the ranks inherit this process's stdout untouched
    json_out()
    from wetts_amd import sharding

    rank, local_rank, world = sharding.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    def partial_line(phase, seen, why):
        """A multi-rank job that cannot finish (a rank that never came up, a hung collective, a rank that crashed) still
        prints ONE line from rank 0 -- no value, `partial`, the phase, and how many ranks were seen -- and exits with
        code 4, instead of hanging until the launcher gives up."""
        if rank != 0:
            return
        sharding.diag("partial_line", phase=phase, ranks_seen=seen, why=why)
        print(json.dumps({
            "metric": "audio samples/sec + RTF @22.05 kHz, VITS-Baker, 1/2/4/8 MI355X", "value": None,
            "unit": "samples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "partial": True, "ranks_seen": seen if seen is not None else 1, "failed_phase": phase, "error": why,
            "config": {"workload": f"{args.config}: did not complete"}}), file=json_out(), flush=True)

    mon = sharding.PhaseMonitor(rank, world, on_expire=partial_line)
    try:
        _bench(args, rank, local_rank, world, mon)
    except SystemExit:
        raise
    except BaseException as e:  # an exception out of a collective (timeout, a peer that went away): same exit path
        if world > 1:
            mon.expire_now(f"{type(e).__name__}: {str(e)[:400]}")
        raise
    finally:
        mon.done()


def _bench(args, rank, local_rank, world, mon):
    from wetts_amd import batching, checkpoint, config, sharding, synth
    # phase deadlines (s): the collective phases stay below the process-group timeout (sharding.DEFAULT_TIMEOUT_S) so
    # that the monitor -- which can still print the partial line -- fires before c10d's watchdog aborts the process
    mon.enter("device", 100)
    be = load_backend(rank, local_rank)
    dev = be.device
    single_dev = getattr(be, "single_dev", False)
    mon.enter("rendezvous", 100)
    sharding.init_process_group()
    mon.attach_store()
    mon.enter("weights", 100)

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
    net = be.make_model(mname, n_vocab, n_speakers)
    cfg = net.cfg
    hop = net.hop_length
    # SURVEY 8(d) duration pinning: 6.0 +- 0.5 frames per phoneme.  The synthetic duration heads give
    # ceil(6 e^{0.1 z}) = 6.57 on average at length_scale 1; 0.92 calibrates the mean to 6.0-6.1
    infer_kw = dict(noise_scale=0.667, length_scale=args.length_scale, noise_scale_w=0.8)
    decode = args.decode or pre.get("decode", "padded")
    if decode == "ragged":
        infer_kw["ragged"] = True

    # ---- weights: rank 0 builds the blob, one broadcast over RCCL, every rank repacks locally
    numel = checkpoint.blob_numel(cfg)
    sd = None
    if rank == 0:
        sd = synth.make_state_dict(cfg, seed=0)
        blob = checkpoint.pack_blob(cfg, sd).to(dev)
    else:
        blob = torch.empty(numel, dtype=torch.float32, device=dev)
    be.sync()
    if world > 1:
        dist.barrier()
    t_b0 = time.perf_counter()
    sharding.broadcast_blob(blob, src=0)
    be.sync()
    bcast_ms = (time.perf_counter() - t_b0) * 1e3
    # every rank verifies what it received: min == max over ranks of the blob fingerprint
    fp = blob_fingerprint(blob)
    blob_ok, ranks_seen = True, 1
    if world > 1:
        lo, hi = fp.clone(), fp.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        blob_ok = bool(torch.equal(lo, hi)) and bool(torch.isfinite(fp).all())
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        if not blob_ok:  # never time a rank that decodes with other weights
            raise SystemExit(f"rank {rank}: weight blob differs across ranks after the broadcast")
    mon.enter("setup", 300)
    net.load_blob(blob)
    if args.overlap and hasattr(net, "set_overlap"):
        net.set_overlap(True)
    if ddtype != "f32":
        net.set_decoder_dtype(ddtype)
    elif args.decoder_serial:
        net.set_decoder_dtype("f32", serial=True)
    if fdtype != "f32":
        net.set_flow_dtype(fdtype)

    # ---- inputs: global utterance list -> batching.plan: LPT deal to ranks, then each rank's length-sorted shard
    # cut into padded sub-batches (the plan is deterministic, every rank computes the same one)
    total = batch * world
    x, lens, sid = make_inputs(mname, n_vocab, n_speakers, total, phonemes, ragged)
    pl = batching.plan(lens.tolist(), world, max_pad_frac=args.max_pad_frac, max_batch=args.max_batch,
                       ragged=decode == "ragged")
    my_buckets = pl.buckets[rank]
    if args.buckets > 0:  # A/B: round 2's fixed number of equal-count buckets
        my_buckets = batching.equal_count_buckets(pl.shards[rank], lens.tolist(), args.buckets)
    nb = len(my_buckets)
    host_buckets = []
    for bk in my_buckets:
        ii = torch.tensor(bk.indices, dtype=torch.long)
        host_buckets.append((x[ii, :bk.tx].contiguous(), lens[ii].contiguous(), sid[ii].contiguous()))
    dev_buckets = [tuple(t.to(dev) for t in hb) for hb in host_buckets]
    torch.manual_seed(1 + rank)  # the library's Philox stream follows the device generator

    def step(buckets=dev_buckets):
        # eps_w / eps_z = None: both standard-normal draws come from the library's Philox kernel
        outs, masks_ = [], []
        for (xb_, lb_, sb_) in buckets:
            o, attn, y_mask, _ = net.infer(xb_, lb_, sid=sb_, **infer_kw)
            outs.append(o)
            masks_.append(y_mask)
        return outs, masks_

    # untimed set-up, before the W warm-up steps of the contract: a fresh box starts with the GPU
    # in a low power state and with lazy one-time initialisation pending (code objects, the MRF
    # event pool); run the step for ~2.5 s so that neither lands inside the timed region
    be.set_mrf_timing(net, True)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.presteps_s:
        step()
        be.sync()
    be.read_mrf_timing(net)  # drains the set-up events
    be.set_mrf_timing(net, False)
    for _ in range(args.warmup):
        step()
    be.sync()
    mon.enter("timed", 100 + 2.0 * args.steps)
    if world > 1:
        dist.barrier()
    be.set_mrf_timing(net, True)
    frames = 0.0
    be.sync()
    t0 = time.perf_counter()
    masks = []
    for _ in range(args.steps):
        o, y_masks = step()
        masks.extend(y_masks)
    be.sync()
    my_elapsed = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for ym in masks:
        frames += float(ym.sum().item())
    mrf_launched_bytes = be.read_mrf_bytes(net)  # (before the set(.., False) below resets the counters)
    mrf_ms, mrf_launches = be.read_mrf_timing(net)
    be.set_mrf_timing(net, False)
    padded_frames = float(sum(ym.numel() for ym in masks))  # B*Ty: what a padded decode computes
    decoded_frames = frames if decode == "ragged" else padded_frames  # ragged: every row over its own frames
    # In the timed region the dominant class shares the chip with the NEXT step's encoder stages (--overlap 1), so
    # its launches last longer than they do alone.  A few extra steps with the pipelining off give the class's
    # isolated duration -- kernel quality -- beside the timed region's figure (reported as roofline.isolated).
    iso = None
    if args.overlap and hasattr(net, "set_overlap"):
        net.set_overlap(False)
        be.set_mrf_timing(net, True)
        iso_frames = 0.0
        for _ in range(max(1, min(3, args.steps))):
            _, yms = step()
            iso_frames += float(sum((ym.sum().item() if decode == "ragged" else ym.numel()) for ym in yms))
        be.sync()
        iso_bytes = be.read_mrf_bytes(net)
        iso_ms, iso_launches = be.read_mrf_timing(net)
        be.set_mrf_timing(net, False)
        iso = (iso_ms, iso_launches, iso_frames, iso_bytes)
        # ... and, for the f32 decoder, the same with the three-stream fork of a stage's chains off as well
        # (WETTS_DECODER_SERIAL): one launch after the other on one stream -- the form whose per-kernel durations a
        # kernel trace can be compared with (in the forked form the kernels of the three chains overlap in time)
        if ddtype == "f32" and hasattr(net, "set_decoder_dtype") and not getattr(be, "label", None) and \
                not args.decoder_serial:
            net.set_decoder_dtype("f32", serial=True)
            be.set_mrf_timing(net, True)
            ser_frames = 0.0
            for _ in range(max(1, min(3, args.steps))):
                _, yms = step()
                ser_frames += float(sum((ym.sum().item() if decode == "ragged" else ym.numel()) for ym in yms))
            be.sync()
            ser_ms, ser_launches = be.read_mrf_timing(net)
            be.set_mrf_timing(net, False)
            net.set_decoder_dtype("f32", serial=False)
            iso = iso + ((ser_ms, ser_launches, ser_frames),)
        net.set_overlap(True)

    # PCIe-inclusive variant (SURVEY 8d's wall: H2D of the ids, D2H of the audio; the driver contract
    # says inputs are resident when the timed region starts, so this is reported beside `value`,
    # never as it): a few extra steps with pinned host buffers on both sides
    pinned = [tuple(be.pin(t) for t in hb) for hb in host_buckets]
    n_pcie = max(1, min(3, args.steps))
    pcie_rate = be.pcie_pass(net, pinned, n_pcie, False, infer_kw)
    n_pipe = max(n_pcie, min(8, args.steps))
    be.pcie_pass(net, pinned, 1, True, infer_kw)  # allocates the pinned buffers outside the timed pass
    pcie_pipe_rate = be.pcie_pass(net, pinned, n_pipe, True, infer_kw)

    # ---- reduce over ranks: time = max, work = sum; every rank's own loop time is gathered for `rank_ms`
    mon.enter("reduce", 100)
    stat = torch.tensor([elapsed, frames, decoded_frames, mrf_ms, float(mrf_launches), pcie_rate, pcie_pipe_rate],
                        dtype=torch.float64, device=dev)
    rank_ms = [my_elapsed / args.steps * 1e3]
    if world > 1:
        mx = stat.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stat.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, frames, pcie_rate, pcie_pipe_rate = float(mx[0]), float(sm[1]), float(sm[5]), float(sm[6])
        mine_t = torch.tensor([my_elapsed / args.steps * 1e3], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        rank_ms = [float(t.item()) for t in allt]
    backend = dist.get_backend() if world > 1 else "none"
    observed_world = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()  # the last collective: rank 0 formats the line (and, if asked, times the CPU baseline) alone
        be.sync()
        mon.done()  # (before the store goes away under the monitor thread)
        try:  # every rank leaves the group in order (no "process group has NOT been destroyed" noise, no rank that
            dist.destroy_process_group()  # tears its communicator down while a peer still uses it)
        except Exception as e:  # never fail a finished measurement on the way out
            sys.stderr.write(f"[wetts rank {rank}/{world}] destroy_process_group: {e}\n")
    mon.done()
    if rank != 0:
        return

    samples = frames * hop
    value = samples / elapsed
    mfl, mby = be.hifigan_cost(cfg)
    # dominant kernel: algorithmic FLOPs of the MRF convs over the frames rank 0 decoded
    # (padded frames: the decoder has no masks, decoders.py:63-82), / live device time
    mrf_flops = mfl * decoded_frames
    mrf_tflops = mrf_flops / (mrf_ms * 1e-3) / 1e12 if mrf_ms > 0 else 0.0
    mrf_gbs = mby * decoded_frames / (mrf_ms * 1e-3) / 1e9 if mrf_ms > 0 else 0.0
    nl_ = max(1, mrf_launches)
    default_shape = not (args.model or args.batch or args.phonemes or args.buckets or args.speakers)

    def lib_digest():
        try:
            return open(os.path.join(ROOT, "wetts_amd", "lib", "build.sha256")).read().strip()[:16]
        except OSError:
            return None

    def committed_traffic(key):
        """Committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this same command (profiles/): (bytes per launch,
        file, digest of the library sources the counter pass ran on)."""
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
            for f in reversed(files):
                d = json.load(open(f))
                if key in d and (default_shape or key == "pw_" + str(args.model)):
                    return d[key]["hbm_bytes_per_launch"], os.path.basename(f), d[key].get("lib_digest", d.get("lib_digest"))
        except Exception:
            pass
        return None, None, None

    traffic_memo = {}
    live_note = {}

    def live_wanted():
        if args.live_traffic == "0" or world != 1 or be.label or not torch.cuda.is_available():
            return False
        if args.live_traffic == "1":
            return True
        # auto: never inside a profiler run (tools/gpu_round.sh profiles this script; rocprofv3 preloads its tool library)
        return not any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) and \
            "rocprofiler" not in os.environ.get("LD_PRELOAD", "")

    def pmc_traffic(key):
        """HBM bytes per launch of the dominant class from the PMC counters: (bytes, source, digest of the library the
        counter pass ran on).  LIVE when possible -- the two `rocprofv3 --pmc` passes of this same command (one step each)
        run as child processes right here, so the figure belongs to the library and the box of THIS line; else the
        committed summary under profiles/ (`traffic_current` then says whether it was taken on the library that just ran)."""
        if key in traffic_memo:
            return traffic_memo[key]
        res = committed_traffic(key)
        if live_wanted() and key != "none":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import pmc_traffic as pt
                cls = "mrf16" if key.startswith("mrf16_") else "pw" if key.startswith("pw_") else key
                keep, skip = [], False
                for a in sys.argv[1:]:  # this command, one timed step, no CPU baseline, no nested refresh
                    if skip:
                        skip = False
                        continue
                    if a in ("--steps", "--warmup", "--presteps-s", "--live-traffic", "--cpu-sample", "--cpu-fixture"):
                        skip = True
                        continue
                    if a.startswith(("--steps=", "--warmup=", "--presteps-s=", "--live-traffic=")) or a == "--no-cpu-baseline":
                        continue
                    keep.append(a)
                child = [os.path.abspath(__file__)] + keep + ["--steps", "1", "--warmup", "1", "--presteps-s", "0.3",
                                                              "--no-cpu-baseline", "--live-traffic", "0"]
                t_l = time.perf_counter()
                b_l, n_l = pt.live(child, cls, budget_s=240.0, log=lambda m: sys.stderr.write(m + "\n"))
                if b_l is not None:
                    live_note.update(committed={"traffic": res[0], "source": res[1], "head": res[2]},
                                     launches_in_counter_pass=n_l, seconds=round(time.perf_counter() - t_l, 1))
                    res = (b_l, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, run by bench.py "
                                "after the timed region", lib_digest())
                else:
                    live_note.update(error=str(n_l))
            except Exception as e:  # a measurement aid must never cost the line
                live_note.update(error=f"{type(e).__name__}: {e}")
        traffic_memo[key] = res
        return res

    def stamp_traffic(roof, src, dig):
        roof["traffic_source"] = src
        roof["traffic_head"] = dig  # sources digest (wetts_amd/lib/build.sha256) of the library the PMC pass ran on
        roof["traffic_current"] = bool(dig) and dig == lib_digest()
        if live_note:
            roof["traffic_live"] = dict(live_note)

    # algorithmic bytes of the class at the granularity it was launched with (wetts_read_mrf_bytes), over the timed
    # steps; a ragged decode computes lens[b] of the dense rows the library counted
    dense_frac = decoded_frames / padded_frames if padded_frames > 0 else 1.0
    launched_bytes = mrf_launched_bytes * dense_frac
    launched_gbs = launched_bytes / (mrf_ms * 1e-3) / 1e9 if mrf_ms > 0 else 0.0

    if ddtype == "uint8":
        # the dynamically quantised graph keeps f32 tensors between its nodes (DynamicQuantizeLinear in, Cast * scale
        # + bias out), so the per-conv algorithmic bytes are the f32 figure; int8 MFMA makes every Conv node HBM-bound
        roofline = {
            "kernel": "the quantised Conv nodes of the MRF ResBlocks: qminmax + qquantize + qconv_i8_kernel "
                      "(v_mfma_i32_32x32x32_i8) per node; ConvTranspose1d stays f32 (conv_mfma_kernel)",
            "bound": "hbm", "achieved": launched_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": launched_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("mrf_uint8")[0],
            "traffic_unit": "HBM bytes per Conv node (its quantise + integer-conv kernels), PMC",
            "launches": int(mrf_launches), "avg_launch_ms": mrf_ms / nl_,
            "launches_note": "one 'launch' = one Conv node = three kernels (range, quantise, integer conv)",
            "bytes_per_launch": launched_bytes / nl_,
            "bytes_note": "f32 planes each Conv node reads / writes once (nothing is fused across nodes: = SURVEY 8d's "
                          "per-conv figure)",
            "mrf_share_of_step": mrf_ms / (elapsed * 1e3),
        }
        stamp_traffic(roofline, *pmc_traffic("mrf_uint8")[1:])
    elif ddtype != "f32":
        # 16-bit activations.  `achieved` = the algorithmic bytes of the launches AS LAUNCHED (a fused ResBlock1 pair:
        # x in, x' out; a whole ResBlock2 stage: x in, mean out -- SURVEY 8d's "resblock-fused" accounting, counted by
        # the library per launch) / live device time: a fraction of something the kernels do, <= 1 by construction.
        # SURVEY 8d's per-conv figure (every conv reads its input and writes its output, every residual add reads x once
        # more) counts planes the fused kernels never move; it is kept as the labelled side view `perconv_view`, an
        # accounting rate that can exceed the peak.  k = 3 blocks are HBM / latency-bound, k = 11 blocks MFMA-bound --
        # both views are reported.
        gbs = 0.5 * mrf_gbs
        traffic, traffic_src, traffic_dig = pmc_traffic(f"mrf16_{args.config}")
        roofline = {
            "kernel": "the 16-bit MRF ResBlock class: rb2_stage16_kernel / resblock_pair16_kernel / conv_bf16_kernel "
                      f"({ddtype} channel-last)",
            "bound": "hbm", "achieved": launched_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": launched_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_unit": "HBM bytes per MRF launch (2*FETCH_SIZE+WRITE_SIZE)*1024, PMC",
            "launches": int(mrf_launches), "avg_launch_ms": mrf_ms / nl_,
            "bytes_per_launch": launched_bytes / nl_,
            "bytes_note": "algorithmic bytes at launch granularity (wetts_read_mrf_bytes): per launch, every 16-bit "
                          "[B][C][T] plane it must read or write once",
            "perconv_view": {"accounting_rate": gbs, "unit": "GB/s", "ratio_to_hbm_peak": gbs / HBM_PEAK_GBS,
                             "bytes_per_launch": 0.5 * mby * decoded_frames / nl_,
                             "note": "SURVEY 8(d) per-conv bytes / the same time: NOT a bandwidth (the fused kernels do "
                                     "not move these bytes); shown for continuity with rounds 1-4"},
            "mfma_view": {"achieved": mrf_tflops, "peak": 2500.0, "unit": "TFLOP/s",
                          "frac": mrf_tflops / 2500.0},
            "mrf_share_of_step": mrf_ms / (elapsed * 1e3),
        }
        stamp_traffic(roofline, traffic_src, traffic_dig)
        # the bandwidth the class really drew (PMC bytes per launch of the committed counter pass / live duration):
        if traffic and mrf_ms > 0:
            real = traffic / (mrf_ms / nl_ * 1e-3) / 1e9
            roofline["hbm_real"] = {"achieved": real, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": real / HBM_PEAK_GBS,
                                    "note": "PMC bytes per launch / live launch duration"}
    else:
        traffic, traffic_src, traffic_dig = pmc_traffic("pw_" + mname if "vocos" in mname else
                                                        "dominant_conv_mfma" if args.config == "baker" else "none")
        roofline = {
            "kernel": ("pw_gemm_kernel (Vocos ConvNeXt pointwise GEMMs 512<->1536, LDS-DMA GEMM of gemm_pw.hip)"
                       if "vocos" in mname else
                       "the MRF ResBlock conv class: conv_mfma_kernel<.., MRF=true, FAST=true> (single "
                       "convs) + resblock_chain32_kernel (whole ResBlock1 / single pairs)"),
            "bound": "mfma", "achieved": mrf_tflops, "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": mrf_tflops / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per MRF launch (2*FETCH_SIZE+WRITE_SIZE)*1024, PMC",
            "launches": int(mrf_launches), "avg_launch_ms": mrf_ms / nl_,
            "flops_per_launch": mrf_flops / nl_,
            "hbm_view": {"achieved": launched_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": launched_gbs / HBM_PEAK_GBS, "bytes_per_launch": launched_bytes / nl_,
                         "note": "algorithmic bytes at launch granularity (wetts_read_mrf_bytes); fp32 convs are "
                                 f"compute-bound (per-conv AI {(mfl / mby) if mby > 0 else 0.0:.0f} flop/B from "
                                 "wetts_hifigan_cost > ridge ~20)"},
            "mrf_share_of_step": mrf_ms / (elapsed * 1e3),
        }
        stamp_traffic(roofline, traffic_src, traffic_dig)
    if iso and iso[0] > 0 and roofline.get("unit") in ("TFLOP/s", "GB/s"):
        iso_ms, iso_launches, iso_frames, iso_bytes = iso[:4]
        per = mfl * iso_frames if roofline["unit"] == "TFLOP/s" else iso_bytes * dense_frac
        ach = per / (iso_ms * 1e-3) / (1e12 if roofline["unit"] == "TFLOP/s" else 1e9)
        roofline["isolated"] = {
            "achieved": ach, "frac": ach / roofline["peak"], "avg_launch_ms": iso_ms / max(1, iso_launches),
            "note": "the same class over extra steps with the pipelining off (nothing else on the chip): kernel quality; "
                    "`achieved` / `frac` above are the timed region's, where the class runs beside the next step's "
                    "encoder stages"}
    if iso and len(iso) > 4 and iso[4][0] > 0 and roofline.get("unit") == "TFLOP/s":
        ser_ms, ser_launches, ser_frames = iso[4]
        ach = mfl * ser_frames / (ser_ms * 1e-3) / 1e12
        roofline["isolated_serial"] = {
            "achieved": ach, "frac": ach / roofline["peak"], "launches_per_step": ser_launches / max(1, min(3, args.steps)),
            "avg_launch_ms": ser_ms / max(1, ser_launches),
            "note": "pipelining off AND the stage's three ResBlock chains on ONE stream (WETTS_DECODER_SERIAL, grouped "
                    "launches): one kernel at a time, so `avg_launch_ms` is directly a rocprofv3 --stats average "
                    "(profiles/r05_kernel_stats_serial.csv).  In the default schedule the chains of a stage run on three "
                    "streams: `launches` / `avg_launch_ms` above are then the class's WALL time (HIP events around each "
                    "stage) per launch, and a trace's per-kernel durations overlap and sum to more than it"}
    prec = ("fp32" if ddtype == "f32" else
            "uint8 dynamic-quantised decoder convs (int32 accumulate)" if ddtype == "uint8" else
            ddtype + " decoder") + \
        ("" if fdtype == "f32" else f" + {fdtype} flow WaveNet layers")
    valid_phonemes = float(lens.sum())
    # the BASELINE.json label only when the preset's own model and shape run; an override is named as one
    overridden = [k for k, v in (("model", args.model), ("batch", args.batch), ("phonemes", args.phonemes),
                                 ("speakers", args.speakers), ("decoder-dtype", args.decoder_dtype),
                                 ("flow-dtype", args.flow_dtype)) if v]
    wtag = pre["tag"] if not overridden else \
        "not a BASELINE.json config: preset '" + args.config + "' with --" + ", --".join(overridden) + " overridden"
    out = {
        "metric": "audio samples/sec + RTF @22.05 kHz, VITS-Baker, 1/2/4/8 MI355X",
        "value": value,
        # contract (4): inputs are resident in HBM when the timed region starts.  SURVEY 8(d)'s wall -- which includes the
        # H2D of the ids and the D2H of the audio -- is measured in the same run and printed beside it
        "value_excludes": "H2D of phoneme ids / D2H of audio (inputs and outputs resident in HBM); the PCIe-inclusive rate "
                          "of SURVEY 8(d) is `pcie_inclusive_pipelined_samples_per_s`",
        "unit": "samples/s", "n_gpus": observed_world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if ddtype == "f32" and fdtype == "f32" else
                 (prec if ddtype == "uint8" else f"{prec} (f32 accumulate)"),
        "data": "synthetic (seeded phoneme ids, seeded random-init weights, "
                f"{frames / args.steps / max(1.0, valid_phonemes):.2f} frames/phoneme at length_scale "
                f"{args.length_scale}, Philox noise drawn on the device)",
        "rtf": elapsed / (samples / sr), "x_realtime": (samples / sr) / elapsed,
        "config": {"workload": f"{args.config}_{mname} infer(): B={batch}/GPU x "
                               f"{phonemes} phonemes{' ragged U{32..' + str(phonemes) + '}' if ragged else ''}, "
                               f"{prec}, {n_speakers} speaker(s), {sr} Hz ({wtag})",
                   "global_batch": total, "phonemes": phonemes, "hop": hop,
                   "padded_sub_batches_per_step": nb, "decode": decode,
                   "decoder_schedule": ("one stream (--decoder-serial)" if args.decoder_serial else
                                        "a stage's k = 3 / 7 / 11 ResBlock chains on three streams") if ddtype == "f32" else "one stream",
                   "pipelining": ("encoder stages of call k + 1 on a side stream beside the decoder of call k "
                                  "(SynthesizerTrn.set_overlap)") if args.overlap else "none",
                   "sub_batch_plan": {"chosen_by": "equal-count (--buckets)" if args.buckets > 0 else
                                      "wetts_amd.batching.plan (DP over the length-sorted shard)",
                                      "max_pad_frac": args.max_pad_frac, "phoneme_pad_frac": pl.stats["pad_frac"],
                                      "sizes_rank0": [len(b) for b in my_buckets],
                                      "tx_rank0": [b.tx for b in my_buckets]},
                   "frame_pad_frac_rank0": 1.0 - (stat[1].item() / padded_frames) if padded_frames else 0.0,
                   "n_speakers": n_speakers, "sampling_rate": sr,
                   "length_scale": args.length_scale,
                   "frames_per_phoneme": frames / args.steps / max(1.0, valid_phonemes),
                   "valid_frames_per_step": frames / args.steps,
                   "parallelism": f"utterance-shard x{observed_world} (backend {backend}"
                                  f"{', all ranks on one device: dry run' if single_dev else ''}), "
                                  f"weights broadcast once ({numel * 4 / 1e6:.0f} MB, {bcast_ms:.1f} ms),"
                                  " no collectives in the decode loop"},
        "ranks_seen": ranks_seen, "blob_checksum_ok": blob_ok,
        "rank_ms": rank_ms, "imbalance": max(rank_ms) / (sum(rank_ms) / len(rank_ms)) - 1.0,
        "plan_slot_imbalance": pl.stats["imbalance"],
        "pcie_inclusive_samples_per_s": pcie_rate,
        "pcie_inclusive_pipelined_samples_per_s": pcie_pipe_rate,
        "pcie_inclusive_note": "ids H2D + infer() + audio D2H per step (SURVEY 8(d)'s wall), reported beside `value`, "
                               f"never as it.  Serial: blocking copy-out each step ({n_pcie} steps); pipelined: pinned "
                               f"double buffer, copy-out on its own stream under the next step ({n_pipe} steps)",
        "roofline": roofline,
    }
    if be.label:
        out["backend_label"] = be.label
    sharding.diag("result", n_gpus=observed_world, value=value, ms_per_step=out["ms_per_step"], ranks_seen=ranks_seen,
                  rank_ms=rank_ms, imbalance=out["imbalance"], workload=out["config"]["workload"])
    # contract: the CPU baseline is a rank-0, N = 1 measurement; --cpu-baseline-multi adds it at N > 1 too
    # (rank 0, after the timed region and its barrier; the other ranks wait at the final barrier)
    if not args.no_cpu_baseline and (world == 1 or args.cpu_baseline_multi):
        out["cpu_baseline"] = cpu_baseline(cfg, sd, x, lens, sid, min(args.cpu_sample, total), sr, hop,
                                           args.length_scale, n_fixture=min(args.cpu_fixture, total))
    print(json.dumps(out), file=json_out(), flush=True)


if __name__ == "__main__":
    main()
