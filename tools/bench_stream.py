#!/usr/bin/env python3
"""Streaming (chunked decoder) latency on one MI355X -- SURVEY 8(f).1: the reference's streaming
clients (inference_onnx.py:37-76, runtime/core/model/vits_model.cc:96-153) run the encoder once and
the decoder window by window (chunk 40 frames, pad 10); first-chunk latency is what they publish.

Measures, at B=1 on synthetic Baker-v1 weights: encoder time, first-window decoder time, first-chunk
latency (host ids in -> first audio piece on the host), middle-window time, whole-utterance stream
time and the non-streaming call, with and without HIP-graph replay of the decoder windows.
Prints one JSON line; `--cpu` adds the oracle (CPU port) timing of the same two stages."""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wetts_amd import SynthesizerTrn, checkpoint, config, synth  # noqa: E402
from wetts_amd.session import (DecoderSession, EncoderSession, InferenceSession,  # noqa: E402
                               depad_bounds, get_chunks)


def med(f, n):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="v1", help="wetts_amd.config.MODEL_CONFIGS key")
    ap.add_argument("--phonemes", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=40)
    ap.add_argument("--pad", type=int, default=10)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="one launch per conv (diagnostic)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = SynthesizerTrn(256, 513, 32, n_speakers=1, **config.MODEL_CONFIGS[args.model]).to(dev)
    sr = config.SAMPLING_RATES[args.model]
    cfg = net.cfg
    sd = synth.make_state_dict(cfg, seed=0)
    net.load_blob(checkpoint.pack_blob(cfg, sd).to(dev))
    hop = net.hop_length
    if args.unfused:
        net.set_decoder_dtype(torch.float32, fused=False)
    torch.manual_seed(0)
    ids = torch.randint(0, 256, (1, args.phonemes)).numpy()
    feeds = {"input": ids, "input_lengths": np.array([args.phonemes], dtype=np.int64),
             "scales": np.array([[0.667, 1.0, 0.8]], dtype=np.float32),
             "sid": np.array([0], dtype=np.int64)}
    enc, full = EncoderSession(net), InferenceSession(net)
    dec, decg = DecoderSession(net), DecoderSession(net, use_graph=True)
    torch.manual_seed(1)
    z = enc.run(None, feeds)[0]
    L = z.shape[1]
    wins = get_chunks(L, args.chunk, args.pad)
    sid = feeds["sid"]

    def stream(d):
        out = []
        for i, (a, b) in enumerate(wins):
            o = d.run(None, {"z": z[:, a:b], "sid": sid})[0].reshape(1, -1)
            lo, hi = depad_bounds(len(wins), i, args.chunk, args.pad, hop, o.shape[1])
            out.append(o[0, lo:hi])
        return np.concatenate(out)

    a0, a1 = stream(dec), stream(decg)  # warm-up (captures the graphs) + equality
    res = {"model": args.model, "sampling_rate": sr, "phonemes": args.phonemes, "frames": int(L),
           "audio_s": L * hop / float(sr),
           "windows": len(wins), "chunk": args.chunk, "pad": args.pad,
           "graph_equals_plain": bool(np.array_equal(a0, a1)), "samples": int(a0.size)}
    for _ in range(3):
        enc.run(None, feeds)
        full.run(None, feeds)
    w0, wm = wins[0], wins[min(1, len(wins) - 1)]
    res["encoder_ms"] = med(lambda: enc.run(None, feeds), args.reps)
    for name, d in (("plain", dec), ("graph", decg)):
        res[f"first_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, w0[0]:w0[1]], "sid": sid}), args.reps)
        res[f"middle_window_ms_{name}"] = med(
            lambda: d.run(None, {"z": z[:, wm[0]:wm[1]], "sid": sid}), args.reps)
        res[f"stream_total_ms_{name}"] = med(lambda: stream(d), max(5, args.reps // 3))
        res[f"first_chunk_latency_ms_{name}"] = res["encoder_ms"] + res[f"first_window_ms_{name}"]
    res["non_stream_ms"] = med(lambda: full.run(None, feeds), args.reps)
    res["rtf_stream_graph"] = (res["encoder_ms"] + res["stream_total_ms_graph"]) / 1e3 / res["audio_s"]
    if args.cpu:  # the oracle (CPU port of the reference) on the same two stages
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import vits_oracle as vo
        import util
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
                res[f"cpu_first_window_ms_{thr}thr"] = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
